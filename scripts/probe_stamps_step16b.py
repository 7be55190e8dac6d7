#!/usr/bin/env python
"""Cycle stamps of one workgroup (tile 37) of k_step16, last depth update of an inference forward on 40-atom molecules — for a library
built with -DDMPNN_STEP16_STAMPS2 (round 5: what the 27-29 k cycles between kernel entry and "operand tile landed" consist of:
tile-table read | DMA issue | the rest of the requests | everything LANDED (forced s_waitcnt vmcnt(0): measurement build only) |
the x contraction | barrier), or a regular build (DMPNN_STAMPS2=0: the stamps of scripts/probe_stamps_step16.py).
usage: DMPNN_LIB=chemprop_amd/variants/libdmpnn_<tag>.so python scripts/probe_stamps_step16b.py [n_mols] [kind]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chemprop_amd import _lib, engine, synth
from chemprop_amd.nn import BondMessagePassing

dev = torch.device("cuda:0")
lib = _lib.load()
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "synth40"
s2 = os.environ.get("DMPNN_STAMPS2", "1") == "1"
bmg = synth.random_batch(nm, kind, seed=1)
bmg.to(dev)
mp = BondMessagePassing().to(dev).eval()
plan = engine.GraphPlan.from_bmg(bmg, light=True)
W = (mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias)
buf = torch.zeros(128, dtype=torch.int64, device=dev)
names = ["entry"] + (["tile table read", "DMA issued", "everything requested", "everything LANDED (forced wait)"] if s2 else []) + [
    "x contraction done (stamp 1)", "barrier passed: operand tile landed", "MFMA loop issued", "unscaled", "all waves through the contraction",
    "tile written", "segment pass 1", "end (message rows written)"]
tag = os.path.basename(os.environ.get("DMPNN_LIB", "in-tree"))
with torch.no_grad():
    for _ in range(3):
        engine.forward(plan, bmg.V, bmg.E, *W, depth=3, route="fused16")
    for rep in range(3):
        buf.zero_()
        lib.dmpnn_debug_timestamps(buf.data_ptr())
        engine.forward(plan, bmg.V, bmg.E, *W, depth=3, route="fused16")
        torch.cuda.synchronize()
        lib.dmpnn_debug_timestamps(None)
        st = buf.cpu().tolist()
        print(f"--- [{tag}] rep {rep}: last depth update, tile 37  ({nm} {kind} molecules, {int(bmg.E.shape[0])} directed edges)")
        prev = st[0]
        for i, n in enumerate(names):
            if i < len(st) and st[i]:
                print(f"{n:46s} +{st[i] - prev:8d} cycles   (t = {st[i] - st[0]})")
                prev = st[i]
