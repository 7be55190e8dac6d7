"""Kernel A/B timing on one box: the tile kernel alone (plan + pre-split reused, direct C calls), K0 alone, and the module's
steady forward, for the library named by DMPNN_LIB (default: the in-tree build).
usage: [DMPNN_LIB=...] python scripts/ab_tile.py [n_mols ...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chemprop_amd import _lib, engine, synth  # noqa: E402
from chemprop_amd.data import BatchMolGraph  # noqa: E402
from chemprop_amd.nn import BondMessagePassing  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
tag = os.path.basename(os.environ.get("DMPNN_LIB", "in-tree"))
sizes = [int(a) for a in sys.argv[1:]] or [512, 4096]


def timed(fn, n=200, reps=5):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / n * 1e3)
    best.sort()
    return best[0], best[len(best) // 2]


for n_mols in sizes:
    b = synth.random_batch(n_mols, "qm9", seed=0)
    b.to(dev)
    bmg = BatchMolGraph.from_tensors(b.V, b.E, b.edge_index, b.rev_edge_index, b.batch, len(b))
    torch.manual_seed(0)
    mp = BondMessagePassing().eval().to(dev)
    with torch.no_grad():
        for _ in range(5):
            out = mp(bmg)
        r = mp.__dict__.get("_dmpnn_replay")
        assert r is not None
        nV, nE = int(bmg.V.shape[0]), int(bmg.E.shape[0])
        nbytes = engine.plan_bytes(nV, nE)
        buf = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
        a = _lib.FwdArgs.from_buffer_copy(r.args)
        a.plan, a.n_atoms, a.n_edges = buf.data_ptr(), nV, nE
        a.V, a.E, a.out = bmg.V.data_ptr(), bmg.E.data_ptr(), out.data_ptr()
        a.Mv = a.Hv = buf.data_ptr()
        a.edge_index, a.rev_edge_index = bmg.edge_index.data_ptr(), bmg.rev_edge_index.data_ptr()
        a.flags |= _lib.F_WSPLIT_READY
        small = engine.small_plan_fits(nV, nE)
        a.flags = (a.flags & ~_lib.F_LOADER_TILES) | (0 if small else _lib.F_LOADER_TILES)
        a.n_tiles_launch = 0
        sp = torch.empty((3 * nE + nV) * a.ldh, dtype=torch.float32, device=dev)
        a.spill_ws, a.spill_bytes = sp.data_ptr(), sp.numel() * 4
        stream = engine._stream_ptr(dev)
        k0 = lambda: lib.dmpnn_prepare_tiles(a.edge_index, a.rev_edge_index, bmg.batch.data_ptr(), nV, nE, buf.data_ptr(), nbytes, stream)
        assert k0() == 0
        if os.environ.get("AB_EXACT_GRID"):   # launch exactly the plan's tiles (header word 6) instead of the layout's bound
            torch.cuda.synchronize()
            a.n_tiles_launch = int(buf[6].item()) + int(os.environ.get("AB_EXACT_GRID_EXTRA", "0"))
            print(f"  grid: {a.n_tiles_launch} workgroups (the plan's tiles)")
        fw = lambda: lib.dmpnn_forward(C.byref(a), stream)
        assert fw() == 0
        torch.cuda.synchronize()
        ref = mp(bmg)
        assert torch.equal(ref, out)
        t_tile = timed(fw)
        t_k0 = timed(k0)
        t_both = timed(lambda: (k0(), fw()))
        t_mod = timed(lambda: mp(bmg))
    upd = 2 * nE
    print(f"[{tag}] {n_mols} mols E={nE}: tile kernel {t_tile[0]:.2f} (med {t_tile[1]:.2f}) us | K0 {t_k0[0]:.2f} us | K0+tile {t_both[0]:.2f} us"
          f" | module forward {t_mod[0]:.2f} (med {t_mod[1]:.2f}) us = {upd / t_mod[0]:.1f} M edge-updates/s")
