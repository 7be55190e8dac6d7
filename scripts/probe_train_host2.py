"""Where the host time of the 512-molecule block training step goes: cProfile by own time, backward on the calling thread (as bench.py)."""
import cProfile
import pstats
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
    with ddp.backward_on_calling_thread():
        mp(bmg).backward(G)
    sync.allreduce()
    opt.step()


for _ in range(30):
    step()
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: enqueue {5e3 * (t1 - t0):7.1f} us/step   total {5e3 * (t2 - t0):7.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(400):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
