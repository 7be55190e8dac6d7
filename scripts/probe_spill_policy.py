"""One 40-atom molecule among N QM9-shaped ones: the tile route with one spill tile (bare tensors) against the per-step
routes (the batching code knows: bmg.oversize) — the data behind the module-level switch (nn._route, nn._spill_monitor)."""
import os
import sys

import torch

sys.path.insert(0, ".")
from chemprop_amd import synth
from chemprop_amd.data import BatchMolGraph
from chemprop_amd.nn import BondMessagePassing

dev = torch.device("cuda:0")


def timed(f, n=30):
    for _ in range(8):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for n_small in (tuple(int(a) for a in sys.argv[1:]) or (512, 1024, 2000, 3000, 4096)):
    mgs = synth.random_molgraphs(n_small, "qm9", seed=3)
    mgs[n_small // 2] = synth.random_molgraphs(1, "synth40", seed=9)[0]
    host = BatchMolGraph(mgs)
    host.to(dev)
    bare = BatchMolGraph.from_tensors(host.V, host.E, host.edge_index, host.rev_edge_index, host.batch, len(host))
    clean = synth.random_batch(n_small, "qm9", seed=3)
    clean.to(dev)
    torch.manual_seed(0)
    res = {}
    with torch.no_grad():
        for tag, b, force in (("no oversize molecule", clean, None), ("tile route + 1 spill tile", bare, False), ("per-step routes", host, None)):
            mp = BondMessagePassing().eval().to(dev)
            if force is False:
                os.environ["DMPNN_VALIDATE"] = "never"   # (no look at the first batches: the module stays on the tile route — the measurement)
            for _ in range(4):
                mp(b)
            res[tag] = (timed(lambda: mp(b)), mp.__dict__.get("_dmpnn_route"), mp.__dict__.get("_dmpnn_replay") is not None)
            if force is False:
                os.environ.pop("DMPNN_VALIDATE")
    print(f"{n_small} molecules, {int(host.E.shape[0])} directed edges: " + "; ".join(f"{k}: {v[0]:.1f} us ({'tile kernel' if v[2] else v[1]})" for k, v in res.items()))
