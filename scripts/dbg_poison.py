#!/usr/bin/env python
"""Uninitialised-read hunt: fill the caching allocator's pools with NaN, free them, then run the module-path training step
(model.loss(...).backward()) of the flaky test's configuration and report which gradients are not finite."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chemprop_amd import synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

dev = torch.device("cuda:0")
pat = float(sys.argv[1]) if len(sys.argv) > 1 else float("nan")
# poison: large blocks (split later for big buffers) and many small ones (the small pool)
big = [torch.full((64 << 20,), pat, device=dev) for _ in range(16)]      # 16 x 256 MB
small = [torch.full((n,), pat, device=dev) for n in (64, 300, 1200, 4096, 30000, 90000, 200000) for _ in range(64)]
torch.cuda.synchronize()
del big, small

from chemprop_amd import agg as cagg
from chemprop_amd.model import MPNN, RegressionFFN
from chemprop_amd.nn import BondMessagePassing

act, n_mols = "elu", 512
torch.manual_seed(17)
model = MPNN(BondMessagePassing(activation=act), cagg.NormAggregation(), RegressionFFN(n_tasks=1, activation=act), batch_norm=True).to(dev).train()
gen = torch.Generator().manual_seed(23)
targets = torch.randn(n_mols, 1, generator=gen).to(dev)
weights = (0.5 + torch.rand(n_mols, 1, generator=gen)).to(dev)
bmg = synth.random_batch(n_mols, "qm9", seed=31)
bmg.to(dev)
for rep in range(2):
    model.zero_grad(set_to_none=True)
    loss = model.loss(bmg, targets, weights)
    loss.backward()
    torch.cuda.synchronize()
    bad = {k: int((~torch.isfinite(p.grad)).sum()) for k, p in model.named_parameters() if not torch.isfinite(p.grad).all()}
    print(f"rep {rep} loss {float(loss):.6f} non-finite gradient entries: {bad}")
    for k, n in bad.items():
        g = dict(model.named_parameters())[k].grad
        idx = (~torch.isfinite(g)).nonzero()
        print("   ", k, tuple(g.shape), "first bad", idx[:6].tolist(), "rows", sorted(set(idx[:, 0].tolist()))[:12] if g.dim() == 2 else "", "cols", sorted(set(idx[:, -1].tolist()))[:12])
