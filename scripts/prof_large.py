"""Profiler target at 32 768 molecules (round-1 VERDICT: the HBM-roofline claims for the scatter/gather step had HIP events
only). Runs, on one batch of 32 768 synthetic molecules whose working set (~1.4 GB) exceeds the 256 MiB Infinity Cache:
  * K2 stand-alone, k_segment<message>  (M[e] = S[src e] - H[rev e]), 20 launches;
  * the whole forward on the route `dmpnn_forward` picks at this size (fused16: k_rows16<SEG> + k_step16 per depth step), 5 passes.
Prints one JSON line with HIP-event times; run under `rocprofv3 --kernel-trace --stats` (and separate --pmc passes) to get
the per-kernel durations / FETCH_SIZE / WRITE_SIZE that profiles/r02_large_* hold.

usage: python scripts/prof_large.py [--mols 32768] [--hidden 300] [--depth 3]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=32768)
    ap.add_argument("--hidden", type=int, default=300)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--kind", default="zinc")
    a = ap.parse_args()
    from chemprop_amd import engine, synth
    from chemprop_amd.nn import BondMessagePassing

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    big = synth.random_batch(a.mols, a.kind, seed=5)
    big.to(dev)
    plan = engine.GraphPlan.from_bmg(big)
    nE = int(big.E.shape[0])
    h = a.hidden
    H = torch.randn(nE, h, device=dev)
    M = torch.empty(nE, h, device=dev)

    def ev(fn, n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n

    k2 = lambda: engine.message(plan, H, out=M)
    ev(k2, 3)
    t2 = ev(k2, 20)
    b2 = 2.0 * nE * h * 4 + 3.0 * nE * 4
    mp = BondMessagePassing(d_h=h, depth=a.depth).to(dev).eval()
    with torch.no_grad():
        fwd = lambda: mp(big)
        ev(fwd, 2)
        tf = ev(fwd, 5)
    print(json.dumps({
        "molecules": a.mols, "directed_edges": nE, "hidden": h, "depth": a.depth,
        "k_segment_message": {"launch_us": round(t2 * 1e3, 1), "algorithmic_bytes": b2, "GB/s": round(b2 / (t2 * 1e-3) / 1e9, 1),
                              "frac_of_8TBs": round(b2 / (t2 * 1e-3) / 1e9 / 8000.0, 4)},
        "forward": {"ms": round(tf, 3), "route": getattr(mp, "_dmpnn_route", None),
                    "edge_updates_per_s_M": round(nE * a.depth / (tf * 1e-3) / 1e6, 1)},
    }))


if __name__ == "__main__":
    main()
