import sys
sys.path.insert(0, "/root/repo")
import torch
from chemprop_amd import synth, engine
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
bmg = synth.random_batch(256, "synth40", seed=11); bmg.to(dev)
torch.manual_seed(3)
mp = BondMessagePassing(bias=True, undirected=True).to(dev)
plan = engine.GraphPlan.from_bmg(bmg)
G = torch.randn(bmg.V.shape[0], 300, device=dev)
res = {}
for mf in ("f32", "split16"):
    out, st = engine.forward(plan, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, mp.W_i.bias, mp.W_h.bias,
                             depth=3, undirected=True, keep=True, route="general", mfma=mf)
    need = {k: True for k in ("W_i", "b_i", "W_h", "b_h", "W_o", "b_o")}; need.update(W_d=False, b_d=False)
    grads = engine.backward(st, G, need)
    res[mf] = (out, st, grads)
    print(mf, st.route)
def perr(a, b): return float((a - b).abs().max() / max(1.0, float(b.abs().max())))
a, b = res["split16"], res["f32"]
print("out", perr(a[0], b[0]))
print("H0", perr(a[1].H0, b[1].H0), "Hs", [perr(x, y) for x, y in zip(a[1].Hs, b[1].Hs)], "Ms", [perr(x, y) for x, y in zip(a[1].Ms, b[1].Ms)], "Mv", perr(a[1].Mv, b[1].Mv))
for k in a[2]:
    if a[2][k] is not None: print(k, perr(a[2][k], b[2][k]))
ma, mb = a[0] > 0, b[0] > 0
d = (ma != mb)
print("out mask flips", int(d.sum()), "per col max", int(d.sum(0).max()), "rows", int(d.any(1).sum()))
z = torch.where(d, b[0].abs() + a[0].abs(), torch.zeros_like(a[0]))
print("magnitudes at flips: max", float(z.max()), "out absmax", float(b[0].abs().max()))
for t in range(2):
    dd = (a[1].Hs[t] > 0) != (b[1].Hs[t] > 0)
    zz = torch.where(dd, a[1].Hs[t].abs() + b[1].Hs[t].abs(), torch.zeros_like(a[1].Hs[t]))
    print("Hs", t, "flips", int(dd.sum()), "max magnitude at flip", float(zz.max()), "absmax", float(b[1].Hs[t].abs().max()))
dd = (a[1].H0 > 0) != (b[1].H0 > 0)
print("H0 sign flips", int(dd.sum()))
