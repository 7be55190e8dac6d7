#!/usr/bin/env python
"""Hunt for the flaky NaN of the split-row products: the module-path training step, repeated in ONE process with the caches emptied
between repetitions (fresh allocations), reporting the pattern of non-finite gradient entries."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chemprop_amd import agg as cagg
from chemprop_amd import synth
from chemprop_amd.model import MPNN, RegressionFFN
from chemprop_amd.nn import BondMessagePassing

dev = torch.device("cuda:0")
act, n_mols = sys.argv[1] if len(sys.argv) > 1 else "elu", 512
torch.manual_seed(17)
model = MPNN(BondMessagePassing(activation=act), cagg.NormAggregation(), RegressionFFN(n_tasks=1, activation=act), batch_norm=True).to(dev).train()
gen = torch.Generator().manual_seed(23)
targets = torch.randn(n_mols, 1, generator=gen).to(dev)
weights = (0.5 + torch.rand(n_mols, 1, generator=gen)).to(dev)
bmg = synth.random_batch(n_mols, "qm9", seed=31)
bmg.to(dev)
n_bad = 0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    model.zero_grad(set_to_none=True)
    loss = model.loss(bmg, targets, weights)
    loss.backward()
    torch.cuda.synchronize()
    bad = {k: int((~torch.isfinite(p.grad)).sum()) for k, p in model.named_parameters() if not torch.isfinite(p.grad).all()}
    if bad or rep == 0:
        print(f"rep {rep} loss {float(loss.detach()):.6f} non-finite gradient entries: {bad}")
    for k, n in bad.items():
        g = dict(model.named_parameters())[k].grad
        idx = (~torch.isfinite(g)).nonzero()
        rows = sorted(set(idx[:, 0].tolist()))
        cols = sorted(set(idx[:, -1].tolist())) if g.dim() == 2 else []
        print("   ", k, tuple(g.shape), "rows", rows[:8], "..", rows[-3:], f"({len(rows)})", "cols", cols[:8], "..", cols[-3:], f"({len(cols)})")
    n_bad += bool(bad)
    if rep % 3 == 2:
        torch.cuda.empty_cache()
print("repetitions with a non-finite gradient:", n_bad)
