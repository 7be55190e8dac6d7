import sys
sys.path.insert(0, "/root/repo")
import torch
from chemprop_amd import synth, engine
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
bmg = synth.random_batch(512, "qm9", seed=1000); bmg.to(dev)
mp = BondMessagePassing().to(dev).eval()
plan = engine.GraphPlan.from_bmg(bmg)
wc = {}
def f():
    with torch.no_grad():
        return engine.forward(plan, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, depth=3, wcache=wc)
out, st = f()
print(st.route, bool(torch.isnan(out).any()))
for _ in range(5): f()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): f()
b.record(); b.synchronize()
print("us", a.elapsed_time(b) / 50 * 1e3)
