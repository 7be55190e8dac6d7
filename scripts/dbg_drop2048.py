"""Errors of the fused-dropout training step at 1 024 / 2 048 molecules against the restated forward given the hash masks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from chemprop_amd import synth
from chemprop_amd.nn import BondMessagePassing
from oracle import dropout_hash as dh
from test_dropout_gpu import ReplayDropout, _restated_forward
from conftest import parity_err
dev = torch.device("cuda:0")
for n_mols, p, plan_kind in ((2048, 0.3, "full"), (2048, 0.3, "tiles"), (1024, 0.3, "full"), (2048, 0.0, "full")):
    os.environ["DMPNN_TRAIN_PLAN"] = plan_kind
    os.environ["DMPNN_VALIDATE"] = "never"
    cpu_bmg = synth.random_batch(n_mols, "qm9", seed=33)
    torch.manual_seed(4)
    mp = BondMessagePassing(dropout=p)
    state = {k: v.clone() for k, v in mp.state_dict().items()}
    G = torch.randn(cpu_bmg.V.shape[0], mp.output_dim, generator=torch.Generator().manual_seed(6))
    mp = mp.to(dev).train()
    bmg = synth.random_batch(n_mols, "qm9", seed=33); bmg.to(dev)
    errs = []
    for rep in range(2):
        mp.zero_grad(); torch.manual_seed(1234)
        out = mp(bmg); st = out.grad_fn.st
        seed = int(st.args.dropout_seed)
        (out * G.to(dev)).sum().backward()
        grads = {k: q.grad.cpu().numpy().copy() for k, q in mp.named_parameters()}
        errs.append(grads)
    d_h, nE, nV = 300, cpu_bmg.E.shape[0], cpu_bmg.V.shape[0]
    if p > 0:
        scale = 1.0 / (1.0 - p)
        masks = [torch.from_numpy(dh.keep_mask(seed, t, nE, d_h, p).astype(np.float32) * np.float32(scale)) for t in range(2)]
        masks.append(torch.from_numpy(dh.keep_mask(seed, 2, nV, d_h, p).astype(np.float32) * np.float32(scale)))
        drop = ReplayDropout(p, masks)
    else:
        drop = torch.nn.Identity()
    ref = BondMessagePassing(dropout=p); ref.load_state_dict(state); ref.train()
    ref_out = _restated_forward(cpu_bmg, ref, drop)
    (ref_out * G).sum().backward()
    print(n_mols, p, plan_kind, "tiles_only", st.plan.tiles_only, "out err %.2e" % parity_err(out.detach().cpu().numpy(), ref_out.detach().numpy()),
          " ".join(f"{k}:{parity_err(errs[1][k], q.grad.numpy()):.1e}" for k, q in ref.named_parameters()),
          "| run-to-run", " ".join(f"{parity_err(errs[0][k], errs[1][k]):.0e}" for k in errs[0]), flush=True)
