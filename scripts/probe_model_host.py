"""Host issue time vs wall time of FusedTrainer.step (512 qm9 molecules), with and without the prefetched plan; cProfile of the host side."""
import copy, cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chemprop_amd import agg as cagg, synth
from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
from chemprop_amd.nn import BondMessagePassing

dev = torch.device("cuda:0")
b = synth.random_batch(512, "qm9", seed=1000); b.to(dev)
b2 = copy.copy(b); b2.edge_index, b2.rev_edge_index, b2.batch = b.edge_index.clone(), b.rev_edge_index.clone(), b.batch.clone()
pair = [b, b2]
torch.manual_seed(0)
m = MPNN(BondMessagePassing(d_h=300), cagg.NormAggregation(), RegressionFFN(n_tasks=1, input_dim=300), batch_norm=True).to(dev).train()
tr = FusedTrainer(m, lr=1e-4)
y = torch.randn(512, 1, device=dev)

def plain(i): tr.step(pair[i & 1], y)
def pref(i):
    tr.prefetch_plan(pair[(i + 1) & 1]); tr.step(pair[i & 1], y)

from chemprop_amd import engine
from chemprop_amd.nn import _plan_key
plans = [engine.GraphPlan.from_bmg(x, light="tiles") for x in pair]
torch.cuda.synchronize()
ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
def noplan(i):   # (K0 not run at all: the plan of the batch built once, handed over as if prefetched long ago)
    x = pair[i & 1]
    m.message_passing.__dict__["_dmpnn_prefetched"] = {_plan_key(x): (plans[i & 1], ev, "tiles")}
    tr.step(x, y)

for name, fn in (("plain", plain), ("prefetch", pref), ("no K0", noplan), ("plain", plain)):
    for i in range(20): fn(i)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:9s}: host issue {1e6 * (t1 - t0) / n:7.1f} us/step   wall {1e6 * (t2 - t0) / n:7.1f} us/step")
m.message_passing.__dict__.pop("_dmpnn_prefetched", None)
# the module path (BondMessagePassing through autograd + FlatAdam), K0 inside the forward vs prefetched between forward and backward
from chemprop_amd import distributed as ddp
from chemprop_amd.optim import FlatAdam
torch.manual_seed(0)
mp = BondMessagePassing(d_h=300).to(dev).train()
sync = ddp.GradSync(list(mp.parameters()), modules=[mp]); opt = FlatAdam(sync, lr=1e-4)
G = torch.randn(int(b.V.shape[0]), 300, device=dev)
def mod(i, prefetch):
    with ddp.backward_on_calling_thread():
        out = mp(pair[i & 1])
        if prefetch: mp.prefetch_plan(pair[(i + 1) & 1])
        out.backward(G)
    sync.allreduce(); opt.step()
for name, pf in (("module", False), ("module+pf", True), ("module", False), ("module+pf", True)):
    for i in range(20): mod(i, pf)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n): mod(i, pf)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    mp.__dict__.pop("_dmpnn_prefetched", None)
    print(f"{name:9s}: host issue {1e6 * (t1 - t0) / n:7.1f} us/step   wall {1e6 * (t2 - t0) / n:7.1f} us/step")
sys.exit(0)
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for i in range(200): plain(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
