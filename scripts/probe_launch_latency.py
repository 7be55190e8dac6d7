#!/usr/bin/env python
"""Latency probe: where does the fixed per-launch cost of the contraction go?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import engine, synth
dev = torch.device("cuda:0")
def t_us(fn, reps=50):
    for _ in range(10): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3
x = torch.zeros(256, device=dev)
print("torch x.add_(1) [256 floats]:", round(t_us(lambda: x.add_(1.0)), 2), "us  (launch floor of a trivial dependent kernel)")
for (M, N, K) in [(48, 300, 32), (48, 300, 300), (48 * 64, 300, 32), (48 * 190, 300, 32), (48 * 190, 300, 64), (48*256, 300, 300), (48*512, 300, 300), (16, 64, 4), (48*190, 64, 32)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    print(f"linear M={M:6d} N={N} K={K:4d}: {t_us(lambda: engine.linear(A, W, None, act='relu', out=C)):8.2f} us")
# python-side cost of one call (no GPU wait): time 200 calls wall vs events
import time
A = torch.randn(48, 32, device=dev); W = torch.randn(300, 32, device=dev); C = torch.empty(48, 300, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): engine.linear(A, W, None, act='relu', out=C)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host issue time per linear call:", round((t1 - t0) / 200 * 1e6, 2), "us; incl. drain:", round((t2 - t0) / 200 * 1e6, 2))
bmg = synth.random_batch(512, "qm9", seed=1000); bmg.to(dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): plan = engine.GraphPlan.from_bmg(bmg)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host issue time per GraphPlan:", round((t1 - t0) / 200 * 1e6, 2), "us")
from chemprop_amd.nn import BondMessagePassing
mp = BondMessagePassing().to(dev).eval()
with torch.no_grad():
    for _ in range(5): mp(bmg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): mp(bmg)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host issue time per forward:", round((t1 - t0) / 200 * 1e6, 2), "us; wall incl. drain:", round((t2 - t0) / 200 * 1e6, 2))
