#!/usr/bin/env python
"""The flaky NaN of round 6 showed on a module's FIRST training batch (validated on the full plan, k_prepare_small): every repetition here
is a first batch — a fresh module with the same weights — and must reproduce the first repetition's gradients bit for bit.
   python scripts/dbg_first_batch.py [act] [reps] [mols]"""
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
n_mols = int(sys.argv[3]) if len(sys.argv) > 3 else 512
torch.manual_seed(17)
proto = MPNN(BondMessagePassing(activation=act), cagg.NormAggregation(), RegressionFFN(n_tasks=1, activation=act), batch_norm=True)
state = {k: v.clone() for k, v in proto.state_dict().items()}
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
    model = MPNN(BondMessagePassing(activation=act), cagg.NormAggregation(), RegressionFFN(n_tasks=1, activation=act), batch_norm=True)
    model.load_state_dict(state)
    model = model.to(dev).train()
    with torch.cuda.stream(side):
        for _ in range(rep % 4):
            A @ A
    if rep % 5 == 4:
        torch.cuda.empty_cache()
    loss = model.loss(bmg, targets, weights)
    loss.backward()
    torch.cuda.synchronize()
    route = model.message_passing.__dict__.get("_dmpnn_route")
    g = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    if ref is None:
        ref = g
        print("route of the first batch:", route)
        continue
    bad = [k for k in g if not torch.equal(g[k], ref[k])]
    nan = [k for k in g if not torch.isfinite(g[k]).all()]
    if bad or nan:
        n_diff += bool(bad); n_nan += bool(nan)
        if n_diff + n_nan <= 4:
            for k in bad:
                d = (g[k] != ref[k]).nonzero()
                bad_e = (~torch.isfinite(g[k])).nonzero()
                print(f"rep {rep}: {k} {tuple(g[k].shape)} differs in {len(d)} entries, max |d| {float((g[k] - ref[k]).abs().nan_to_num(1e30).max()):.3e}, non-finite {len(bad_e)}: first {bad_e[:6].tolist()} last {bad_e[-3:].tolist()}")
print(f"act {act} mols {n_mols} TILE_WAVES={os.environ.get('DMPNN_TILE_WAVES')} KEEP_ROWS={os.environ.get('DMPNN_KEEP_ROWS')}: {reps} first batches, {n_diff} with a differing gradient, {n_nan} with a non-finite one")
