"""CPU restatement of ``chemprop/nn/agg.py`` (TEST INFRASTRUCTURE ONLY — only ``tests/`` may import it).

Same ATen op sequence as the reference, so CPU results are bit-identical to it:
``scatter_reduce_(0, index, H, reduce, include_self=False)`` on a zero tensor of ``batch.max() + 1``
rows.  Pinned by ``tests/golden/agg/*.npz`` (frozen from the executed reference by
``tests/golden/make_golden_agg.py``) and re-checked live against the reference when ``/root/reference``
is present (``tests/test_oracle.py``).
"""
from __future__ import annotations

import torch
from torch import Tensor


def _scatter(H: Tensor, batch: Tensor, reduce: str) -> Tensor:
    # agg.py:74-79 / 91-96
    index = batch.unsqueeze(1).repeat(1, H.shape[1])
    dim_size = int(batch.max()) + 1
    return torch.zeros(dim_size, H.shape[1], dtype=H.dtype).scatter_reduce_(0, index, H, reduce=reduce, include_self=False)


def mean(H: Tensor, batch: Tensor) -> Tensor:
    """MeanAggregation.forward, agg.py:73-79."""
    return _scatter(H, batch, "mean")


def sum_(H: Tensor, batch: Tensor) -> Tensor:
    """SumAggregation.forward, agg.py:90-96."""
    return _scatter(H, batch, "sum")


def norm(H: Tensor, batch: Tensor, c: float = 100.0) -> Tensor:
    """NormAggregation.forward, agg.py:112-113."""
    return sum_(H, batch) / c


def attentive(H: Tensor, batch: Tensor, W: Tensor, b: Tensor) -> Tensor:
    """AttentiveAggregation.forward, agg.py:123-133."""
    dim_size = int(batch.max()) + 1
    logits = torch.nn.functional.linear(H, W, b).exp()
    Z = torch.zeros(dim_size, 1, dtype=H.dtype).scatter_reduce_(0, batch.unsqueeze(1), logits, reduce="sum", include_self=False)
    alphas = logits / Z[batch]
    index = batch.unsqueeze(1).repeat(1, H.shape[1])
    return torch.zeros(dim_size, H.shape[1], dtype=H.dtype).scatter_reduce_(0, index, alphas * H, reduce="sum", include_self=False)
