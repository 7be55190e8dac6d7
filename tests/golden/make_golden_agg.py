#!/usr/bin/env python
"""Freeze golden vectors of the aggregation step from the EXECUTED reference (build container only).

    python tests/golden/make_golden_agg.py        # rewrites tests/golden/agg/*.npz

Runs the reference's own ``chemprop.nn.agg`` classes (imported from ``/root/reference`` through
``oracle/ref_shim.py``; CPU torch, fp32) and stores inputs ``H, batch``, the outputs of Mean / Sum / Norm /
Attentive aggregation and the gradients of ``sum(out * G)`` w.r.t. ``H`` (and the attentive layer).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def cases():
    rng = np.random.default_rng(7)
    sizes = rng.integers(1, 30, size=40)
    yield "agg_qm9like_h300", np.repeat(np.arange(40), sizes), 300, 100.0
    yield "agg_gaps_h7", np.array([0, 0, 0, 2, 2, 5, 5, 5, 5, 6]), 7, 3.5           # molecules 1, 3, 4 have no atoms
    yield "agg_single_atoms_h64", np.arange(9), 64, 100.0
    yield "agg_one_molecule_h20", np.zeros(33, dtype=np.int64), 20, 50.0


def main():
    ref_shim.install()
    from chemprop.nn.agg import AttentiveAggregation, MeanAggregation, NormAggregation, SumAggregation

    for name, batch, d, c in cases():
        g = torch.Generator().manual_seed(len(batch) * 31 + d)
        H = torch.randn(len(batch), d, generator=g)
        b = torch.as_tensor(batch, dtype=torch.int64)
        n_mols = int(b.max()) + 1
        G = torch.randn(n_mols, d, generator=g)
        rec = {"H": H.numpy(), "batch": b.numpy(), "G": G.numpy(), "norm": np.float32(c)}
        for key, mod in (("mean", MeanAggregation()), ("sum", SumAggregation()), ("norm", NormAggregation(norm=c))):
            Hr = H.clone().requires_grad_(True)
            out = mod(Hr, b)
            (out * G).sum().backward()
            rec[f"out_{key}"] = out.detach().numpy()
            rec[f"gH_{key}"] = Hr.grad.numpy()
        torch.manual_seed(d)
        att = AttentiveAggregation(output_size=d)
        with torch.no_grad():
            att.W.weight.mul_(0.3)
        Hr = H.clone().requires_grad_(True)
        out = att(Hr, b)
        (out * G).sum().backward()
        rec.update(att_W=att.W.weight.detach().numpy(), att_b=att.W.bias.detach().numpy(), out_att=out.detach().numpy(),
                   gH_att=Hr.grad.numpy(), gW_att=att.W.weight.grad.numpy(), gb_att=att.W.bias.grad.numpy())
        np.savez_compressed(os.path.join(OUT, "agg", name + ".npz"), **rec)
        print(name, H.shape, n_mols)


if __name__ == "__main__":
    main()
