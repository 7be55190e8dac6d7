#!/usr/bin/env python
"""Contention probe: time of the K=300 contraction vs number of concurrently active tiles, random vs zero data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import engine
dev = torch.device("cuda:0")
def t_us(fn, reps=40):
    for _ in range(8): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3
N, K = 300, 300
for fill in ("randn", "zeros"):
    for tiles in (1, 8, 32, 64, 128, 190, 256):
        M = 48 * tiles
        A = (torch.randn(M, K, device=dev) if fill == "randn" else torch.zeros(M, K, device=dev))
        W = (torch.randn(N, K, device=dev) if fill == "randn" else torch.zeros(N, K, device=dev))
        C = torch.empty(M, N, device=dev)
        print(f"{fill:6s} tiles={tiles:4d}: {t_us(lambda: engine.linear(A, W, None, act='relu', out=C)):8.2f} us")
# K sweep at 190 tiles, long K to separate per-chunk steady state
for K in (300, 1200, 2400):
    M = 48 * 190
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    t = t_us(lambda: engine.linear(A, W, None, act='relu', out=C), 20)
    print(f"K={K}: {t:8.2f} us  -> {t / (K / 32):.3f} us/chunk, {2.0*M*N*K/t/1e6:.1f} TF")
for K in (1200,):
    for tiles in (1, 64):
        M = 48 * tiles
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
        t = t_us(lambda: engine.linear(A, W, None, act='relu', out=C), 20)
        print(f"tiles={tiles} K={K}: {t:8.2f} us -> {t / (K / 32):.3f} us/chunk")
