#!/usr/bin/env python
"""Timing probe for the fp32-MFMA contraction: separates fixed cost from per-k-chunk cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import engine, synth

dev = torch.device("cuda:0")
def t_ms(fn, reps=30):
    for _ in range(5): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps

only = sys.argv[1] if len(sys.argv) > 1 else "all"
print(f"{'M':>8} {'N':>5} {'K':>5} {'cadd':>5} {'us':>9} {'TF':>8}")
shapes = [(9120, 300, 32, 0), (9120, 300, 96, 0), (9120, 300, 300, 0), (9120, 300, 300, 1), (9120, 300, 600, 1),
          (9120, 300, 88, 0), (9120, 300, 86, 0), (4636, 300, 372, 0), (12288, 300, 300, 1), (36864, 300, 300, 1),
          (73728, 300, 300, 1), (582580, 300, 300, 1), (9120, 64, 300, 1), (256 * 48, 300, 304, 1)]
if only == "one":
    shapes = [(9120, 300, 300, 1)]
for (M, N, K, cadd) in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev); Cadd = torch.randn(M, N, device=dev) if cadd else None
    fn = lambda: engine.linear(A, W, None, Cadd=Cadd, act="relu", out=C)
    ms = t_ms(fn)
    print(f"{M:8d} {N:5d} {K:5d} {cadd:5d} {ms*1e3:9.2f} {2.0*M*N*K/ms/1e9:8.2f}")
# K1-like: gather + concat, scalar path
bmg = synth.random_batch(512, "qm9", seed=1000); bmg.to(dev)
plan = engine.GraphPlan.from_bmg(bmg)
Wi = torch.randn(300, 86, device=dev)
fn = lambda: engine.linear(bmg.V, Wi, None, A2=bmg.E, gather1=plan.src32, n_rows=plan.n_edges)
print("K1 (gather+concat, K=86):", round(t_ms(fn) * 1e3, 2), "us")
fn = lambda: engine.GraphPlan.from_bmg(bmg)
print("K0 plan (512 mols):", round(t_ms(fn) * 1e3, 2), "us")
fn = lambda: engine.GraphPlan.from_bmg(bmg, light=True)
print("K0 plan, light (512 mols):", round(t_ms(fn) * 1e3, 2), "us")
fn = lambda: engine.GraphPlan.from_bmg(bmg, light="tiles")
print("K0 plan, tiles only (512 mols):", round(t_ms(fn) * 1e3, 2), "us")
H = torch.randn(plan.n_edges, 300, device=dev); Mo = torch.empty_like(H)
print("K2 message:", round(t_ms(lambda: engine.message(plan, H, out=Mo)) * 1e3, 2), "us")
print("K4 aggregate:", round(t_ms(lambda: engine.aggregate(plan, H)) * 1e3, 2), "us")
# fused route pieces
H0 = torch.randn(plan.n_edges, 300, device=dev); Wh = torch.randn(300, 300, device=dev) * 0.05
Mn = torch.empty_like(H); Mv = torch.empty(plan.n_atoms, 300, device=dev)
print("fused update (K3+K2):", round(t_ms(lambda: engine.update_fused(plan, H, H0, Wh, M_next=Mn)) * 1e3, 2), "us")
print("fused update (K3+K4):", round(t_ms(lambda: engine.update_fused(plan, H, H0, Wh, want_M=False, want_Mv=True, Mv=Mv)) * 1e3, 2), "us")
Wo = torch.randn(300, 372, device=dev) * 0.05; bo = torch.randn(300, device=dev)
print("K5 finalize:", round(t_ms(lambda: engine.linear(bmg.V, Wo, bo, A2=Mv, act="relu")) * 1e3, 2), "us")
import torch.nn as nn
from chemprop_amd.nn import BondMessagePassing
mp = BondMessagePassing().to(dev).eval()
with torch.no_grad():
    for name, env in (("mega", {}), ("fused", {"DMPNN_MEGA": "0"}), ("general", {"DMPNN_GENERAL": "1"})):
        os.environ.update({"DMPNN_MEGA": "1", "DMPNN_GENERAL": "0"}); os.environ.update(env)
        print("whole forward, eager,", name, ":", round(t_ms(lambda: mp(bmg)) * 1e3, 2), "us")
    os.environ.update({"DMPNN_MEGA": "1", "DMPNN_GENERAL": "0"})
    from chemprop_amd import _lib
    plan2 = engine.GraphPlan.from_bmg(bmg)
    for mf in ("f32", "split16"):
        fw = lambda: engine.forward(plan2, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, depth=3, route="mega", mfma=mf)
        print(f"mega kernel only (plan reused), mfma={mf}:", round(t_ms(fw) * 1e3, 2), "us")
