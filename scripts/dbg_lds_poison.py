#!/usr/bin/env python
"""Uninitialised-LDS hunt: every CU's LDS is filled with a NaN pattern right before the forward and right before the backward of the
module-path training step (a fresh module per repetition = its first, validated batch; then steady-state batches); any kernel that reads
LDS it never wrote turns a gradient non-finite or changes it.   python scripts/dbg_lds_poison.py [act] [reps] [pattern hex]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from chemprop_amd import agg as cagg
from chemprop_amd import synth
from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
from chemprop_amd.nn import BondMessagePassing

lp = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "liblds_poison.so"))
lp.lds_poison.argtypes = [ctypes.c_uint, ctypes.c_void_p]
dev = torch.device("cuda:0")
act = sys.argv[1] if len(sys.argv) > 1 else "elu"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pat = int(sys.argv[3], 16) if len(sys.argv) > 3 else 0x7FC07FC0     # NaN as fp32 and as two f16 NaNs
where = sys.argv[4] if len(sys.argv) > 4 else "both"                 # fwd | bwd | both
n_mols = 512


def poison():
    rc = lp.lds_poison(pat, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


torch.manual_seed(17)
proto = MPNN(BondMessagePassing(activation=act), cagg.NormAggregation(), RegressionFFN(n_tasks=1, activation=act), batch_norm=True)
state = {k: v.clone() for k, v in proto.state_dict().items()}
gen = torch.Generator().manual_seed(23)
targets = torch.randn(n_mols, 1, generator=gen).to(dev)
weights = (0.5 + torch.rand(n_mols, 1, generator=gen)).to(dev)
bmg = synth.random_batch(n_mols, "qm9", seed=31)
bmg.to(dev)
ref = {}
bad_total = 0
for rep in range(reps):
    model = MPNN(BondMessagePassing(activation=act), cagg.NormAggregation(), RegressionFFN(n_tasks=1, activation=act), batch_norm=True)
    model.load_state_dict(state)
    model = model.to(dev).train()
    for batch_no in range(4):          # batches 0, 1: validated on the full plan; 2, 3: the tile plan
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        do_poison = rep > 0
        if do_poison and where in ("fwd", "both"):
            poison()
        loss = model.loss(bmg, targets, weights)
        if do_poison and where in ("bwd", "both"):
            poison()
        loss.backward()
        torch.cuda.synchronize()
        g = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        g["loss"] = loss.detach().clone()
        key = batch_no
        if rep == 0:
            ref[key] = g
            continue
        bad = [k for k in g if not torch.equal(g[k], ref[key][k])]
        if bad:
            bad_total += 1
            if bad_total <= 6:
                print(f"rep {rep} batch {batch_no}: differing: {[k.replace('message_passing.', 'mp.').replace('predictor.', 'p.') for k in bad]}")
                for k in bad[:1]:
                    e = (~torch.isfinite(g[k])).nonzero()
                    print(f"rep {rep} batch {batch_no}: {k} {tuple(g[k].shape)} differs; non-finite entries {len(e)}: first {e[:5].tolist()} last {e[-2:].tolist()}; max |d| {float((g[k] - ref[key][k]).abs().nan_to_num(1e30).max()):.3e}")
# the fused whole-model step too
model = proto.to(dev).train()
tr = FusedTrainer(model, lr=0.0)
outs = []
for i in range(8):
    model.load_state_dict(state)
    if i >= 4:
        poison()
    out = tr.step(bmg, targets, weights)
    torch.cuda.synchronize()
    outs.append([v.detach().clone() for v in tr.sync.views] + [out[0].detach().clone()])
fused_bad = sum(1 for i in range(5, 8) if any(not torch.equal(a, b) for a, b in zip(outs[i], outs[3])))
print(f"act {act} KEEP_ROWS={os.environ.get('DMPNN_KEEP_ROWS')} TILE_WAVES={os.environ.get('DMPNN_TILE_WAVES')} pattern {pat:#x}: module path {reps - 1} poisoned modules x 4 batches, {bad_total} differ; fused step: {fused_bad} of 3 poisoned steps differ")
