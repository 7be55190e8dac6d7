#!/usr/bin/env python
"""Phase timing of the split-MFMA whole-forward tile kernel (workgroup 0), from in-kernel cycle stamps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import engine, synth, _lib
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
lib = _lib.load()
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bmg = synth.random_batch(nm, "qm9", seed=1000); bmg.to(dev)
print("molecules", nm, "edges", bmg.E.shape[0])
mp = BondMessagePassing().to(dev).eval()
plan = engine.GraphPlan.from_bmg(bmg)
buf = torch.zeros(128, dtype=torch.int64, device=dev)
fw = lambda: engine.forward(plan, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, depth=3, route="mega", mfma="split16", keep=("keep" in sys.argv))
with torch.no_grad():
    for _ in range(5): fw()
    lib.dmpnn_debug_timestamps(buf.data_ptr())
    plan = engine.GraphPlan.from_bmg(bmg, light=("tiles" if "tiles" in sys.argv else ("light" in sys.argv)))
    fw(); torch.cuda.synchronize()
    lib.dmpnn_debug_timestamps(None)
st = buf.cpu().tolist()
names = ["entry"] + (["m1 loads issued", "m2 LDS metadata", "m3 atom table"] if os.environ.get("DMPNN_LIB", "").endswith("metastamps.so") else []) + ["meta", "init A staged", "K1 contract", "K1 seg mfma", "K1 tile scale", "K1 split written"]
for u in (1, 2):
    names += [f"upd{u} contract", f"upd{u} unscale+tau", f"upd{u} seg mfma", f"upd{u} tile scale", f"upd{u} split written"]
names += ["fin Mv part", "fin V staged", "fin V part", "out stored"]
prev = st[0]
for i, n in enumerate(names):
    if i < len(st) and st[i]:
        print(f"{n:18s} +{(st[i]-prev):8d} cycles   (t={(st[i]-st[0])})")
        prev = st[i]

print("--- plan kernel (k_prepare_small)")
pn = ["entry", "int64 loaded", "narrow+hist", "validate", "scan", "fill", "sort", "inverse", "outputs+atom tiles", "maxnbr", "piece tiles"]
ps = st[32:]
prev = ps[0]
for i, n in enumerate(pn):
    if ps[i]:
        print(f"{n:20s} +{(ps[i]-prev):8d} cycles   (t={(ps[i]-ps[0])})")
        prev = ps[i]

if "tiles" in sys.argv:
    print("--- (tile plan from the batch vector: entry, loaded, order checked, molecule ranges, packed;  piece stamps: next ptrs at [3], walk at [4])")
print("--- piece tiles")
qs = st[48:]
prev = qs[0]
for i, n in enumerate(["entry", "start bits", "start ranks", "piece starts", "next ptrs", "chain walk"]):
    if qs[i]:
        print(f"{n:20s} +{(qs[i]-prev):8d} cycles"); prev = qs[i]
