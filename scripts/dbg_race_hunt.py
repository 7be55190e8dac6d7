#!/usr/bin/env python
"""Race hunt: the module-path training step (block through autograd + the head node) repeated under timing noise from a second stream;
every gradient must be BIT-IDENTICAL to the first repetition's (all reductions of the step are deterministic).
   python scripts/dbg_race_hunt.py [act] [reps] [noise 0|1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chemprop_amd import agg as cagg
from chemprop_amd import synth
from chemprop_amd.model import MPNN, RegressionFFN
from chemprop_amd.nn import BondMessagePassing

dev = torch.device("cuda:0")
act = sys.argv[1] if len(sys.argv) > 1 else "elu"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
noise = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
n_mols = int(sys.argv[4]) if len(sys.argv) > 4 else 512
torch.manual_seed(17)
model = MPNN(BondMessagePassing(activation=act), cagg.NormAggregation(), RegressionFFN(n_tasks=1, activation=act), batch_norm=True).to(dev).train()
state = {k: v.clone() for k, v in model.state_dict().items()}
gen = torch.Generator().manual_seed(23)
targets = torch.randn(n_mols, 1, generator=gen).to(dev)
weights = (0.5 + torch.rand(n_mols, 1, generator=gen)).to(dev)
bmg = synth.random_batch(n_mols, "qm9", seed=31)
bmg.to(dev)
side = torch.cuda.Stream()
A = torch.randn(2048, 2048, device=dev)
ref = None
n_diff = n_nan = 0
for rep in range(reps):
    model.load_state_dict(state)          # (batch-norm running statistics back: the same arithmetic every repetition)
    model.zero_grad(set_to_none=True)
    if noise:
        with torch.cuda.stream(side):
            for _ in range(rep % 4):
                A @ A
    loss = model.loss(bmg, targets, weights)
    loss.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    if rep < 4:           # (a module's first batches are validated on another plan: the reference repetition is a later one)
        continue
    if ref is None:
        ref = g
        continue
    bad = [k for k in g if not torch.equal(g[k], ref[k])]
    nan = [k for k in g if not torch.isfinite(g[k]).all()]
    if bad or nan:
        n_diff += bool(bad); n_nan += bool(nan)
        if n_diff + n_nan <= 6:
            for k in bad:
                d = (g[k] != ref[k]).nonzero()
                print(f"rep {rep}: {k} differs in {len(d)} entries, max |d| {float((g[k] - ref[k]).abs().max()):.3e}, first {d[:4].tolist()}, non-finite {int((~torch.isfinite(g[k])).sum())}")
print(f"act {act} mols {n_mols} noise {noise} TILE_WAVES={os.environ.get('DMPNN_TILE_WAVES')} KEEP_ROWS={os.environ.get('DMPNN_KEEP_ROWS')}: {reps} repetitions, {n_diff} with a differing gradient, {n_nan} with a non-finite one")
