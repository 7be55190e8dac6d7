#!/usr/bin/env python
"""Whole-forward timing of the routes on the BASELINE configs (eager, plan included)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import engine, synth
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
def t_ms(fn, reps=10):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps
for (n, kind, h, depth) in [(512, "qm9", 300, 3), (512, "zinc", 512, 6), (512, "synth40", 300, 3), (4096, "synth40", 300, 3), (4096, "qm9", 300, 3), (32768, "qm9", 300, 3), (64, "cgr", 300, 3)]:
    bmg = synth.random_batch(n, kind, seed=1); bmg.to(dev)
    d_v, d_e = bmg.V.shape[1], bmg.E.shape[1]
    mp = BondMessagePassing(d_v=d_v, d_e=d_e, d_h=h, depth=depth).to(dev).eval()
    nE = bmg.E.shape[0]
    res = []
    with torch.no_grad():
        plan = engine.GraphPlan.from_bmg(bmg)
        for name, kw in (("auto", {}), ("fused", dict(route="fused")), ("general-f32", dict(route="general", mfma="f32")), ("general16", dict(route="general", mfma="split16"))):
            try:
                f = lambda: engine.forward(plan, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, depth=depth, **kw)
                _, st = f()
                res.append(f"{name}[{st.route}] {t_ms(f)*1e3:8.1f} us")
            except Exception as e:
                res.append(f"{name}: n/a")
        full = t_ms(lambda: mp(bmg))
    print(f"{n:6d} {kind:8s} h={h} d={depth} E={nE:7d} | module {full*1e3:8.1f} us ({nE*(depth-1)/full/1e3:7.1f} M upd/s) | " + " | ".join(res))
