"""TEST INFRASTRUCTURE — CPU restatement of ``chemprop.nn.ffn.MLP`` (chemprop/nn/ffn.py:24-68) as the ATen op
sequence the reference executes (``nn.Sequential`` of ``Linear`` / ``act, dropout, Linear`` blocks, dropout inactive).
Only tests/ may import it.  Pinned: tests/test_ffn.py checks it against goldens frozen from the executed reference
(tests/golden/ffn/*.npz, tests/golden/make_golden_ffn.py).
"""
from __future__ import annotations

from typing import Sequence

import torch.nn.functional as F
from torch import Tensor

from .dmpnn_torch import activation_fn


def mlp_forward(X: Tensor, weights: Sequence[Tensor], biases: Sequence[Tensor], activation="relu") -> Tensor:
    """h_0 = W_0 x + b_0;  h_l = W_l sigma(h_{l-1}) + b_l   (ffn.py:27-35, 49-58)."""
    tau = activation if callable(activation) else activation_fn(activation)
    H = F.linear(X, weights[0], biases[0])
    for W, b in zip(weights[1:], biases[1:]):
        H = F.linear(tau(H), W, b)
    return H
