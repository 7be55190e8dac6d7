#!/usr/bin/env python
"""Fixed vs per-row cost of the weight-gradient kernel (k_wgrad + k_wgrad_reduce)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import engine
dev = torch.device("cuda:0")
def t_ms(fn, reps=30):
    for _ in range(5): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps
for (M, N, K) in [(570, 300, 300), (1140, 300, 300), (2280, 300, 300), (4560, 300, 300), (9120, 300, 300), (18240, 300, 300), (36480, 300, 300),
                  (9120, 300, 86), (9120, 300, 64), (9120, 64, 64), (4636, 300, 372)]:
    gZ = torch.randn(M, N, device=dev); A = torch.randn(M, K, device=dev)
    f = lambda: engine.linear_wgrad(gZ, A)
    print(f"M={M:6d} N={N:4d} K={K:4d}: {t_ms(f)*1e3:8.1f} us (two launches + host)")
