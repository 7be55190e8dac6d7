"""Inference forward of the headline shape at 4 096 molecules (tile kernels on the multi-workgroup tile plan), fresh batch objects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import synth
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
for n in (4096, 16384):
    b = synth.random_batch(n, "qm9", seed=1); b.to(dev)
    torch.manual_seed(0)
    m = BondMessagePassing().eval().to(dev)
    with torch.no_grad():
        for _ in range(6): m(b)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): m(b)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    nE = int(b.E.shape[0])
    print(f"qm9-{n}: forward {best:.1f} us  {2 * nE / best:.1f} M edge-updates/s  route={m.__dict__.get('_dmpnn_route')}")
