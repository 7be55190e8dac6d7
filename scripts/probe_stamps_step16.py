#!/usr/bin/env python
"""Phase timing of one workgroup (tile 37) of the per-step fused route's kernels from in-kernel cycle stamps (dmpnn_debug_timestamps):
k_step16 of an inference forward on 40-atom molecules — which launch is stamped: the LAST one that runs with the buffer armed
(the finalize on the step kernel passes no stamp buffer, so: the last depth update)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chemprop_amd import _lib, engine, synth
from chemprop_amd.nn import BondMessagePassing

dev = torch.device("cuda:0")
lib = _lib.load()
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "synth40"
bmg = synth.random_batch(nm, kind, seed=1)
bmg.to(dev)
mp = BondMessagePassing().to(dev).eval()
plan = engine.GraphPlan.from_bmg(bmg, light=True)
W = (mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias)
buf = torch.zeros(128, dtype=torch.int64, device=dev)
names = ["entry", "everything requested (+ x contraction)", "operand tile landed", "MFMA loop issued", "unscaled", "all waves through the contraction",
         "tile written", "segment pass 1", "end (message rows written)"]
with torch.no_grad():
    for _ in range(3):
        engine.forward(plan, bmg.V, bmg.E, *W, depth=3, route="fused16")
    for depth, label in ((2, "depth 2: the update step that leaves Mv (no message out)"), (3, "depth 3: stamps of the LAST update as well; first update = message out")):
        buf.zero_()
        lib.dmpnn_debug_timestamps(buf.data_ptr())
        engine.forward(plan, bmg.V, bmg.E, *W, depth=depth, route="fused16")
        torch.cuda.synchronize()
        lib.dmpnn_debug_timestamps(None)
        st = buf.cpu().tolist()
        print(f"--- {label}  ({nm} {kind} molecules, {int(bmg.E.shape[0])} directed edges)")
        prev = st[0]
        for i, n in enumerate(names):
            if i < len(st) and st[i]:
                print(f"{n:42s} +{st[i] - prev:8d} cycles   (t = {st[i] - st[0]})")
                prev = st[i]
