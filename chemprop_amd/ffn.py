"""f4, first piece: the predictor's feed-forward stack ``chemprop.nn.ffn.MLP`` (``nn/ffn.py:24-68``) on
the contraction kernel of the block.

    h_0 = W_0 x + b_0,      h_l = W_l dropout(sigma(h_{l-1})) + b_l            (ffn.py:27-35)

Same construction as the reference (``MLP.build``: an ``nn.Sequential`` of ``Sequential(Linear)`` then
``Sequential(act, dropout, Linear)`` blocks, ffn.py:37-58) -> same ``state_dict`` keys (``0.0.weight``, ``1.2.weight``
...), same RNG stream, same ``input_dim`` / ``output_dim``.  ``forward`` runs every layer as ``dmpnn_linear_fwd``:

* built-in activation, dropout inactive, no grad: the activation is fused into the epilogue of the PREVIOUS layer's
  kernel (one launch per layer, nothing else);
* otherwise: the same kernels through the autograd wrappers of ``backward.py`` with ``sigma`` / dropout as the torch
  modules themselves between them (their RNG and parameters behave as in the reference).

Inputs must live on the HIP device: there is no CPU fallback.
"""
from __future__ import annotations

from typing import Sequence

import torch
from torch import Tensor, nn

from . import engine
from .nn import classify_activation, get_activation_function


def mlp_forward(seq: nn.Sequential, X: Tensor) -> Tensor:
    """``MLP.forward`` (= ``nn.Sequential.forward`` over the blocks of ffn.py:49-58) on the kernels."""
    engine._require_device(X, "X")
    blocks = list(seq)
    lin0 = blocks[0][-1]
    if lin0.weight.device != X.device:
        raise RuntimeError(f"module is on {lin0.weight.device} but the input is on {X.device}")
    if X.dim() != 2:
        raise RuntimeError(f"MLP: expected a [rows, {lin0.in_features}] matrix, got {tuple(X.shape)}")
    grad = torch.is_grad_enabled() and (X.requires_grad or any(p.requires_grad for p in seq.parameters()))
    fused = not grad
    acts = []
    for b in blocks[1:]:
        act_m, drop_m = b[0], b[1]
        code, slope, slope_t = classify_activation(act_m)
        if code == "custom" or (seq.training and drop_m.p > 0):
            fused = False
        acts.append((code, slope, slope_t))
    if fused:
        H = X
        for i, b in enumerate(blocks):
            lin = b[-1]
            code, slope, slope_t = acts[i] if i < len(acts) else ("none", 0.0, None)  # sigma of the NEXT block, fused here
            H = engine.linear(H, lin.weight, lin.bias, act=code, slope=slope, slope_t=slope_t)
        return H
    from .backward import linear_fn

    H = linear_fn(X, lin0.weight, lin0.bias)
    for b in blocks[1:]:
        H = linear_fn(b[1](b[0](H)), b[2].weight, b[2].bias)
    return H


class MLP(nn.Sequential):
    """State-dict-compatible mirror of ``chemprop.nn.ffn.MLP`` (ffn.py:24-68)."""

    @classmethod
    def build(cls, input_dim: int, output_dim: int, hidden_dim: int | Sequence[int] = 300, n_layers: int = 1,
              dropout: float = 0.0, activation="relu"):
        # Layer widths input -> hidden ... -> output.  The FIRST block is a bare Linear; every later block is
        # (activation, dropout, Linear) with ONE shared activation and ONE shared dropout module — the structure (and hence
        # the state_dict keys "i.0.*" / "i.2.*") and the order in which the Linear layers draw their initial weights are
        # the reference's (ffn.py:37-58).
        widths = [int(hidden_dim)] * int(n_layers) if isinstance(hidden_dim, int) else [int(h) for h in hidden_dim]
        widths = [int(input_dim), *widths, int(output_dim)]
        shared_drop = nn.Dropout(dropout)
        shared_act = get_activation_function(activation)
        blocks = []
        for i in range(len(widths) - 1):
            lin = nn.Linear(widths[i], widths[i + 1])
            blocks.append(nn.Sequential(lin) if i == 0 else nn.Sequential(shared_act, shared_drop, lin))
        return cls(*blocks)

    @property
    def input_dim(self) -> int:
        return self[0][-1].in_features

    @property
    def output_dim(self) -> int:
        return self[-1][-1].out_features

    def forward(self, X: Tensor) -> Tensor:
        return mlp_forward(self, X)
