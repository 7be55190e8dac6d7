#!/usr/bin/env python
"""Host-side cost of one module forward (eager): cProfile over many calls, and wall time per call with / without sync."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import synth
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bmg = synth.random_batch(nm, "qm9", seed=1000); bmg.to(dev)
mp = BondMessagePassing().to(dev).eval()
with torch.no_grad():
    for _ in range(50): mp(bmg)
    torch.cuda.synchronize()
    N = 3000
    t0 = time.perf_counter()
    for _ in range(N): mp(bmg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue time {1e6*(t1-t0)/N:.1f} us/call; with final sync {1e6*(t2-t0)/N:.1f} us/call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N): mp(bmg)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
