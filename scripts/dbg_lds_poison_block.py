#!/usr/bin/env python
"""Uninitialised-LDS hunt, block only: LDS poisoned before the inference forward, before the training forward and before the backward of
BondMessagePassing alone; outputs / gradients must stay bit-identical.   python scripts/dbg_lds_poison_block.py [act] [mols]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from chemprop_amd import synth
from chemprop_amd.nn import BondMessagePassing

lp = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "liblds_poison.so"))
lp.lds_poison.argtypes = [ctypes.c_uint, ctypes.c_void_p]
dev = torch.device("cuda:0")
act = sys.argv[1] if len(sys.argv) > 1 else "relu"
n_mols = int(sys.argv[2]) if len(sys.argv) > 2 else 512
pat = 0x7FC07FC0


def poison():
    assert lp.lds_poison(pat, torch.cuda.current_stream().cuda_stream) == 0


torch.manual_seed(0)
mp = BondMessagePassing(activation=act).to(dev)
bmg = synth.random_batch(n_mols, "qm9", seed=31)
bmg.to(dev)
G = torch.randn(int(bmg.V.shape[0]), mp.output_dim, device=dev)
# inference
mp.eval()
with torch.no_grad():
    outs = []
    for i in range(8):
        if i >= 4:
            poison()
        outs.append(mp(bmg).clone())
    torch.cuda.synchronize()
print("inference: poisoned outputs differing from the clean one:", sum(1 for i in range(4, 8) if not torch.equal(outs[i], outs[3])),
      "non-finite:", [int((~torch.isfinite(o)).sum()) for o in outs[4:]])
# training
mp.train()
res = []
for i in range(12):
    mp.zero_grad(set_to_none=True)
    mode = "clean" if i < 4 else ("fwd" if i < 8 else "bwd")
    if mode == "fwd":
        poison()
    out = mp(bmg)
    if mode == "bwd":
        poison()
    out.backward(G)
    torch.cuda.synchronize()
    res.append((mode, out.detach().clone(), {k: p.grad.clone() for k, p in mp.named_parameters()}))
ref = res[3]
for mode in ("fwd", "bwd"):
    for r in [x for x in res if x[0] == mode][:2]:
        bad = [k for k in r[2] if not torch.equal(r[2][k], ref[2][k])]
        print(f"training, LDS poisoned before {mode}: output equal {torch.equal(r[1], ref[1])}; differing gradients {bad}")
        for k in bad[:2]:
            e = (~torch.isfinite(r[2][k])).nonzero()
            print(f"    {k} {tuple(r[2][k].shape)} non-finite {len(e)}: rows {sorted(set(e[:, 0].tolist()))[:10]} cols {sorted(set(e[:, -1].tolist()))[:10]}")
