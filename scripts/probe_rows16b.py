import sys, os
sys.path.insert(0, "/root/repo")
import torch
from chemprop_amd import engine, _lib
import ctypes as C
dev = torch.device("cuda:0")
def t_ms(fn, reps=30):
    for _ in range(5): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps
for (M, N, K, cadd) in [(9120, 300, 32, 0), (9120, 300, 32, 1), (9120, 300, 128, 1), (9120, 300, 256, 1), (9120, 300, 300, 1), (9120, 64, 32, 0), (48*256, 300, 300, 1), (48*512, 300, 300, 1)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); Cc = torch.empty(M, N, device=dev); Cadd = torch.randn(M, N, device=dev) if cadd else None
    s = lambda: engine.linear(A, W, None, Cadd=Cadd, act="relu", out=Cc, mfma="split16")
    f = lambda: engine.linear(A, W, None, Cadd=Cadd, act="relu", out=Cc)
    print(M, N, K, cadd, "split %.1f us   f32 %.1f us" % (t_ms(s) * 1e3, t_ms(f) * 1e3))
