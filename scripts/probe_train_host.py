"""Is the 512-molecule training step host-bound?  Per repetition of 100 steps: the time the Python loop needs to ENQUEUE them
(clock read before the closing synchronize) against the time until the device is done.  Enqueue ~ total => host-bound."""
import sys
import time

import torch

sys.path.insert(0, ".")
from chemprop_amd import distributed as ddp
from chemprop_amd import synth
from chemprop_amd.nn import BondMessagePassing
from chemprop_amd.optim import FlatAdam

dev = torch.device("cuda:0")
n_mols = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bmg = synth.random_batch(n_mols, "qm9", seed=1000)
bmg.to(dev)
torch.manual_seed(0)
mp = BondMessagePassing().to(dev).train()
sync = ddp.GradSync(list(mp.parameters()), modules=[mp])
opt = FlatAdam(sync, lr=1e-4)
G = torch.randn(int(bmg.V.shape[0]), mp.output_dim, device=dev)


def step():
    mp(bmg).backward(G)
    sync.allreduce()
    opt.step()


for _ in range(20):
    step()
for rep in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: enqueue {1e4 * (t1 - t0):7.1f} us/step   total {1e4 * (t2 - t0):7.1f} us/step   device tail {1e6 * (t2 - t1):8.1f} us")
# the parts of the host time
import cProfile
import pstats

pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
