#!/usr/bin/env python
"""Freeze golden vectors of ``chemprop.nn.ffn.MLP`` (f4) from the EXECUTED reference.

    python tests/golden/make_golden_ffn.py        # rewrites tests/golden/ffn/*.npz   (build container only)

Reference class through ``oracle/ref_shim.py``, CPU torch, fp32, eval: input, every parameter, ``out = mlp(X)`` and
the gradients of ``sum(out * G)`` w.r.t. every parameter and the input.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ffn")

# name -> (rows, build kwargs, seed)
CASES = {
    "ffn_default_300_1": (37, dict(input_dim=300, output_dim=1), 80),                                  # RegressionFFN defaults
    "ffn_tasks12_2layers": (50, dict(input_dim=300, output_dim=12, hidden_dim=300, n_layers=2), 81),
    "ffn_no_hidden": (9, dict(input_dim=40, output_dim=3, n_layers=0), 82),
    "ffn_dims_list_tanh": (21, dict(input_dim=64, output_dim=5, hidden_dim=[48, 20, 33], activation="tanh"), 83),
    "ffn_leaky_one_row": (1, dict(input_dim=33, output_dim=2, hidden_dim=17, activation="leakyrelu"), 84),
    "ffn_elu_wide": (130, dict(input_dim=303, output_dim=7, hidden_dim=336, activation="elu"), 85),
}


def main():
    ref_shim.install()
    from chemprop.nn.ffn import MLP

    torch.set_num_threads(1)
    os.makedirs(OUT, exist_ok=True)
    for name, (rows, kw, seed) in CASES.items():
        torch.manual_seed(seed)
        mlp = MLP.build(**kw).eval()
        gen = torch.Generator().manual_seed(4000 + seed)
        X = torch.randn(rows, kw["input_dim"], generator=gen, requires_grad=True)
        out = mlp(X)
        G = torch.randn(out.shape, generator=gen)
        (out * G).sum().backward()
        arrs = dict(X=X.detach().numpy(), out=out.detach().numpy(), G=G.numpy(), gX=X.grad.numpy())
        for k, v in mlp.state_dict().items():
            arrs["w." + k] = v.detach().numpy()
        for k, p in mlp.named_parameters():
            arrs["g." + k] = p.grad.numpy()
        meta = dict(name=name, seed=seed, cfg=kw, torch=torch.__version__, out_sum=float(out.detach().sum()))
        arrs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
        print(f"{name:24s} rows={rows:4d} out={tuple(out.shape)} sum={meta['out_sum']:.6f}")


if __name__ == "__main__":
    main()
