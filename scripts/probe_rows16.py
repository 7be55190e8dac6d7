#!/usr/bin/env python
"""Timing of the per-step contraction: fp32-MFMA kernel vs the split-f16 kernel (weights pre-split each call)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import engine
dev = torch.device("cuda:0")
def t_ms(fn, reps=20):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps
print(f"{'M':>8} {'N':>5} {'K':>5} {'f32 us':>9} {'TF':>7} {'split us':>9} {'TF':>7}")
for (M, N, K) in [(9120, 300, 300), (4636, 300, 300), (9120, 300, 86), (25500, 512, 512), (36864, 300, 300), (582580, 300, 300), (291290, 512, 512)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev); Cadd = torch.randn(M, N, device=dev)
    f = lambda: engine.linear(A, W, None, Cadd=Cadd, act="relu", out=C)
    s = lambda: engine.linear(A, W, None, Cadd=Cadd, act="relu", out=C, mfma="split16")
    tf, ts = t_ms(f), t_ms(s)
    fl = 2.0 * M * N * K
    print(f"{M:8d} {N:5d} {K:5d} {tf*1e3:9.1f} {fl/tf/1e9:7.1f} {ts*1e3:9.1f} {fl/ts/1e9:7.1f}")
