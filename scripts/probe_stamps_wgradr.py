#!/usr/bin/env python
"""Cycle stamps of workgroup 0 of k_wgrad16r (the weight-gradient products on split rows) in a 512-molecule training step
(DMPNN_KEEP_ROWS=1 forces the split-row route).  The stamp buffer is armed for the backward pass only."""
import os
import sys

os.environ.setdefault("DMPNN_KEEP_ROWS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chemprop_amd import _lib, synth
from chemprop_amd import distributed as ddp
from chemprop_amd.nn import BondMessagePassing

dev = torch.device("cuda:0")
lib = _lib.load()
n_mols = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bmg = synth.random_batch(n_mols, "qm9", seed=1000)
bmg.to(dev)
torch.manual_seed(0)
mp = BondMessagePassing().to(dev).train()
sync = ddp.GradSync(list(mp.parameters()), modules=[mp])
G = torch.randn(int(bmg.V.shape[0]), mp.output_dim, device=dev)
for _ in range(5):
    mp(bmg).backward(G)
buf = torch.zeros(128, dtype=torch.int64, device=dev)
for rep in range(3):
    out = mp(bmg)
    torch.cuda.synchronize()
    buf.zero_()
    # (the library's stamp pointer is thread-local and the backward pass runs on autograd's device thread: armed from a tensor hook)
    out.register_hook(lambda g: (lib.dmpnn_debug_timestamps(buf.data_ptr()), g)[1])
    out.backward(G)
    torch.cuda.synchronize()
    st = buf.cpu().tolist()
    names = ["entry", "tails requested", "stage 0 requested", "F known", "stage 0 in its image"]
    n_st = (len([x for x in st[:64] if x]) - 6) // 3
    for i in range(n_st):
        names += [f"s{i} top", f"s{i} products + stores + requests", f"s{i} barrier"]
    names += ["slabs stored"]
    print(f"--- rep {rep}: molecules {n_mols}, edges {bmg.E.shape[0]}")
    prev = st[0]
    for i, n in enumerate(names):
        if i < 64 and st[i]:
            print(f"{n:22s} +{st[i] - prev:8d} cycles   (t = {st[i] - st[0]})")
            prev = st[i]
