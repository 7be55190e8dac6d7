"""Phase stamps of k_step16 (workgroup 37) for one config: python scripts/probe_step16.py <kind> <n_mols> [d_h]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import _lib, engine, synth
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
lib = _lib.load()
kind, n = sys.argv[1], int(sys.argv[2]); d_h = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dims = dict(d_v=106, d_e=28) if kind == "cgr" else {}
bmg = synth.random_batch(n, kind, seed=1); bmg.to(dev)
mp = BondMessagePassing(d_h=d_h, **dims).to(dev).eval()
plan = engine.GraphPlan.from_bmg(bmg)
fw = lambda: engine.forward(plan, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, depth=3, route="fused16")
buf = torch.zeros(64, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(5): fw()
    torch.cuda.synchronize()
    lib.dmpnn_debug_timestamps(buf.data_ptr())
    fw(); torch.cuda.synchronize()
    lib.dmpnn_debug_timestamps(None)
st = buf.cpu().tolist()[:16]
names = ["entry", "requested", "landed", "mfma loop", "unscaled", "contraction barrier", "tile written", "pass1", "end(msg)"]
print(f"{kind}-{n} d_h={d_h} E={bmg.E.shape[0]}: stamps of the LAST k_step16 launch that wrote them (the Mv step: no message)")
prev = st[0]
for i, nm in enumerate(names):
    if i < len(st) and st[i]:
        print(f"  {nm:22s} +{st[i]-prev:8d} cycles (t={st[i]-st[0]})"); prev = st[i]
