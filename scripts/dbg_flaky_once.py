#!/usr/bin/env python
"""One fresh process = the flaky test's sequence: CPU restatement (loss + backward), model to the GPU, module-path loss.backward();
prints the pattern of non-finite gradient entries (if any) and of the intermediate tensors it can reach."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from chemprop_amd import agg as cagg
from chemprop_amd import synth
from chemprop_amd.model import MPNN, RegressionFFN
from chemprop_amd.nn import BondMessagePassing
from oracle import model_torch as om

act, n_mols = "elu", 512
cfg = dict(mp=dict(activation=act), agg="norm", bn=True, ffn=dict(n_tasks=1, activation=act))
torch.manual_seed(17)
model = MPNN(BondMessagePassing(activation=act), cagg.NormAggregation(), RegressionFFN(n_tasks=1, activation=act), batch_norm=True)
state = {k: v.clone() for k, v in model.state_dict().items()}
gen = torch.Generator().manual_seed(23)
targets = torch.randn(n_mols, 1, generator=gen)
weights = 0.5 + torch.rand(n_mols, 1, generator=gen)
cpu_bmg = synth.random_batch(n_mols, "qm9", seed=31)
if "nocpu" not in sys.argv:
    ref_model = om.Model(state, cfg)
    ref_loss = ref_model.loss(cpu_bmg, targets, weights, None, None)
    ref_loss.backward()
dev = torch.device("cuda:0")
model = model.to(dev).train()
bmg = synth.random_batch(n_mols, "qm9", seed=31)
bmg.to(dev)
tg, wg = targets.to(dev), weights.to(dev)
loss = model.loss(bmg, tg, wg)
loss.backward()
torch.cuda.synchronize()
bad = {k: int((~torch.isfinite(p.grad)).sum()) for k, p in model.named_parameters() if not torch.isfinite(p.grad).all()}
print(f"loss {float(loss.detach()):.6f} non-finite: {bad}")
for k in bad:
    g = dict(model.named_parameters())[k].grad
    idx = (~torch.isfinite(g)).nonzero()
    rows = sorted(set(idx[:, 0].tolist()))
    cols = sorted(set(idx[:, -1].tolist())) if g.dim() == 2 else []
    print("   ", k, tuple(g.shape), "rows", rows[:10], "..", rows[-3:], f"({len(rows)})", "cols", cols[:10], "..", cols[-3:], f"({len(cols)})")
