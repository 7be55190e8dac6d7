"""Host issue time vs wall time of the whole-model training step at 512 QM9-shaped molecules: FusedTrainer.step (ONE C call) and the
module path (MPNN.loss(...).backward() + FlatAdam); then cProfile of the host side of the fused step.  Host ~ wall => host-bound."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chemprop_amd import agg as cagg, distributed as ddp, synth
from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
from chemprop_amd.nn import BondMessagePassing
from chemprop_amd.optim import FlatAdam

dev = torch.device("cuda:0")
b = synth.random_batch(512, "qm9", seed=1000); b.to(dev)
y = torch.randn(512, 1, device=dev)


def model():
    torch.manual_seed(0)
    return MPNN(BondMessagePassing(d_h=300), cagg.NormAggregation(), RegressionFFN(n_tasks=1, input_dim=300), batch_norm=True).to(dev).train()


m = model()
tr = FusedTrainer(m, lr=1e-4)
m2 = model()
sync = ddp.GradSync([p for p in m2.parameters() if p.requires_grad], modules=[m2.message_passing])
opt = FlatAdam(sync, lr=1e-4)


def fused(): tr.step(b, y)


def module():
    with ddp.backward_on_calling_thread():
        m2.loss(b, y).backward()
    sync.allreduce(); opt.step()


for name, fn in (("fused", fused), ("module", module), ("fused", fused), ("module", module)):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:7s}: host issue {1e6 * (t1 - t0) / n:7.1f} us/step   wall {1e6 * (t2 - t0) / n:7.1f} us/step")
for name, fn in (("fused", fused), ("module", module)):
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(200): fn()
    pr.disable()
    torch.cuda.synchronize()
    print("==== cProfile,", name, "(by own time)")
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
