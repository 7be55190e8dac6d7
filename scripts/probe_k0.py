#!/usr/bin/env python
"""Cycle stamps of the multi-workgroup K0's packing block (k_prepare_tiles_batch_multi, last block), relative to its entry."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import engine, synth, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bmg = synth.random_batch(nm, "qm9", seed=1000); bmg.to(dev)
buf = torch.zeros(128, dtype=torch.int64, device=dev)
for _ in range(5):
    engine.GraphPlan.from_bmg(bmg, light="tiles")
lib.dmpnn_debug_timestamps(buf.data_ptr())
engine.GraphPlan.from_bmg(bmg, light="tiles"); torch.cuda.synchronize()
lib.dmpnn_debug_timestamps(None)
st = buf.cpu().tolist()
t0 = st[32]
for i, n in enumerate(["entry", "published", "ranges in LDS", "ranges checked", "packed (end)"]):
    print(f"{n:18s} t={st[32 + i] - t0}")
for i, n in ((3, "p4 next pointers"), (4, "p5 chain walk + tables")):
    print(f"{n:24s} t={st[48 + i] - t0}")
