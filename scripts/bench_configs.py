"""BASELINE configs 3-5 (+ a large QM9 batch): inference forward of the module, per route, eager, inputs resident.
usage: python scripts/bench_configs.py [json_out]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chemprop_amd import synth  # noqa: E402
from chemprop_amd.data import BatchMolGraph  # noqa: E402
from chemprop_amd.nn import BondMessagePassing  # noqa: E402

dev = torch.device("cuda:0")
CONFIGS = [
    ("zinc-512 h512 d6 (configs[2])", "zinc", 512, dict(d_h=512, depth=6)),
    ("zinc-512 h300 d3", "zinc", 512, dict()),
    ("synth40-512 (configs[3], per-GPU batch 512)", "synth40", 512, dict()),
    ("synth40-4096 (configs[3], per-GPU batch 4096)", "synth40", 4096, dict()),
    ("cgr-64 (configs[4], notebook batch)", "cgr", 64, dict(d_v=106, d_e=28)),
    ("cgr-512 (configs[4])", "cgr", 512, dict(d_v=106, d_e=28)),
    ("qm9-32768", "qm9", 32768, dict()),
]
if len(sys.argv) > 2:
    CONFIGS = [c for c in CONFIGS if any(k in c[0] for k in sys.argv[2:])]


def timed(fn, n, reps=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return min(ts)


out = {}
for name, kind, n_mols, kw in CONFIGS:
    b = synth.random_batch(n_mols, kind, seed=1)
    b.to(dev)
    bmg = BatchMolGraph.from_tensors(b.V, b.E, b.edge_index, b.rev_edge_index, b.batch, len(b))
    bmg.oversize = b.oversize
    nE = int(bmg.E.shape[0])
    res = {"directed_edges": nE, "atoms": int(bmg.V.shape[0]), "mols": n_mols}
    ref = None
    for tag, env in (("before (DMPNN_FUSED16=0)", {"DMPNN_FUSED16": "0"}), ("now", {"DMPNN_FUSED16": "1"})):
        os.environ.update(env)
        torch.manual_seed(0)
        mp = BondMessagePassing(**kw).eval().to(dev)
        with torch.no_grad():
            for _ in range(4):
                o = mp(bmg)
            route = "mega16 (replay)" if mp.__dict__.get("_dmpnn_replay") is not None else mp.__dict__.get("_dmpnn_route")
            us = timed(lambda: mp(bmg), 20 if nE > 100000 else 50)
        if ref is None:
            ref = o
        err = float((o - ref).abs().max() / max(1.0, float(ref.abs().max())))
        upd = nE * (mp.depth - 1)
        res[tag] = {"us": round(us, 1), "M_edge_updates_per_s": round(upd / us, 1), "route": route, "err_vs_before": err}
        print(f"{name:48s} {tag:26s} E={nE:7d} {us:9.1f} us  {upd / us:8.1f} M edge-updates/s  route={route} err={err:.1e}", flush=True)
    res["speedup"] = round(res["before (DMPNN_FUSED16=0)"]["us"] / res["now"]["us"], 3)
    out[name] = res
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
