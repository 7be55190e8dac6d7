"""Training step of the atom-message blocks beside the bond block's (512 QM9-shaped molecules, d_h 300, depth 3): the tile-kernel
route (round 4) and, for the mol-atom-bond blocks, the per-step chain it replaces (DMPNN_MEGA=0).  One JSON line."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import json, time, torch
from chemprop_amd import synth, distributed as ddp
from chemprop_amd.nn import AtomMessagePassing, BondMessagePassing
from chemprop_amd.mab import MABAtomMessagePassing, MABBondMessagePassing
from chemprop_amd.optim import FlatAdam
dev = torch.device("cuda:0")
bmg = synth.random_batch(512, "qm9", seed=0); bmg.to(dev)
nV = bmg.V.shape[0]
res = {}
def timeit(f, n=50):
    for _ in range(8): f()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
only = sys.argv[1] if len(sys.argv) > 1 else None   # "atom": the atom block's training step alone (under rocprofv3)
for name, cls in (("bond", BondMessagePassing), ("atom", AtomMessagePassing)):
    if only and name != only:
        continue
    torch.manual_seed(0)
    m = cls().to(dev).train()
    s = ddp.GradSync(list(m.parameters()), modules=[m]); o = FlatAdam(s, lr=1e-4)
    G = torch.randn(nV, 300, device=dev)
    def f():
        with ddp.backward_on_calling_thread():
            m(bmg).backward(G)
        s.allreduce(); o.step()
    res[name + "_train_us"] = round(timeit(f), 1); res[name + "_route"] = m.__dict__.get("_dmpnn_route")
for name, cls in (() if only else (("mabbond", MABBondMessagePassing), ("mabatom", MABAtomMessagePassing))):
    torch.manual_seed(0)
    m = cls().to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    Gv, Ge = torch.randn(nV, 300, device=dev), torch.randn(bmg.E.shape[0], 300, device=dev)
    def f():
        opt.zero_grad(set_to_none=True)
        hv, he = m(bmg)
        ((hv * Gv).sum() + (he * Ge).sum()).backward()
        opt.step()
    res[name + "_train_us"] = round(timeit(f, 30), 1); res[name + "_route"] = m.__dict__.get("_dmpnn_route")
    import os
    os.environ["DMPNN_MEGA"] = "0"
    m2 = cls().to(dev).train(); opt2 = torch.optim.Adam(m2.parameters(), lr=1e-4)
    def f2():
        opt2.zero_grad(set_to_none=True)
        hv, he = m2(bmg)
        ((hv * Gv).sum() + (he * Ge).sum()).backward()
        opt2.step()
    res[name + "_rows_train_us"] = round(timeit(f2, 10), 1); res[name + "_rows_route"] = m2.__dict__.get("_dmpnn_route")
    os.environ["DMPNN_MEGA"] = "1"
print(json.dumps(res))
