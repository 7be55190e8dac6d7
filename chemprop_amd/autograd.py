"""Dispatch of one BondMessagePassing forward onto the HIP kernels (+ autograd wiring).

Two routes, both entirely on HIP kernels for the arithmetic of the path:

* **fused** — ``tau`` is one of chemprop's built-in activations (nn/utils.py:43-53) and dropout is
  inactive (p == 0 or eval): one ``dmpnn_forward`` C call enqueues the whole chain.
* **rows**  — arbitrary ``nn.Module`` activation, learnable PReLU under grad, or active dropout:
  the row kernels (linear / message / aggregate) are chained from Python and the activation /
  dropout *modules themselves* run in between (so their RNG and parameters behave as in the
  reference, base.py:135-141,180-194).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import engine


def _params(mp):
    g = lambda lin, n: None if lin is None else getattr(lin, n)
    return dict(W_i=mp.W_i.weight, b_i=mp.W_i.bias, W_h=mp.W_h.weight, b_h=mp.W_h.bias,
                W_o=mp.W_o.weight, b_o=mp.W_o.bias, W_d=g(mp.W_d, "weight"), b_d=g(mp.W_d, "bias"))


def _wants_grad(mp, *tensors) -> bool:
    if not torch.is_grad_enabled():
        return False
    if any(p.requires_grad for p in mp.parameters()):
        return True
    return any(t is not None and t.requires_grad for t in tensors)


def mp_forward(mp, plan: engine.GraphPlan, V: Tensor, E: Tensor, V_d: Optional[Tensor],
               max_level: int = 2) -> Tensor:
    from .nn import classify_activation

    act, slope, slope_t = classify_activation(mp.tau)
    drop_active = mp.training and mp.dropout.p > 0
    grad = _wants_grad(mp, V, E, V_d)
    if grad:
        for name, t in (("bmg.V", V), ("bmg.E", E), ("V_d", V_d)):
            if t is not None and t.requires_grad:
                raise NotImplementedError(
                    f"chemprop_amd: gradient w.r.t. `{name}` is not provided by the engine "
                    "(the reference never asks for it: features are data)")
    p = _params(mp)
    has_vd = mp.W_d is not None and V_d is not None
    # (the dropout scale of the backward pass lives in the backward TILE kernel, which only runs when a gradient of W_i or W_h is
    #  wanted: a block with both frozen — W_o alone trainable — keeps the rows route)
    edge_grad = any(t is not None and t.requires_grad for t in (p["W_i"], p["b_i"], p["W_h"], p["b_h"]))
    if (drop_active and grad and edge_grad and act in ("relu", "leakyrelu") and not has_vd and max_level >= 2
            and type(mp.dropout) is torch.nn.Dropout):
        # ACTIVE dropout inside the tile kernels (round 3): the mask is a counter-based hash of (seed, site, row, column), the seed
        # one draw from torch's CPU generator (so torch.manual_seed fixes the run); a batch that takes another route falls
        # through to the rows route below, where the block's own nn.Dropout runs between the kernels (as does any module that is not
        # exactly nn.Dropout: a subclass has its own semantics)
        from .backward import FusedMP

        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        try:
            return FusedMP.apply(mp, plan, V, E, None, act, slope, (slope_t, max_level, (float(mp.dropout.p), seed)),
                                 *[p[k] for k in ("W_i", "b_i", "W_h", "b_h", "W_o", "b_o", "W_d", "b_d")])
        except engine.RouteUnavailable:
            pass
    use_rows = act == "custom" or drop_active or (act == "prelu" and grad)

    if not use_rows:
        if grad:
            from .backward import FusedMP

            return FusedMP.apply(mp, plan, V, E, V_d if has_vd else None, act, slope, (slope_t, max_level),
                                 *[p[k] for k in ("W_i", "b_i", "W_h", "b_h", "W_o", "b_o", "W_d", "b_d")])
        out, st = engine.forward(plan, V, E, p["W_i"], p["W_h"], p["W_o"], p["b_o"], p["b_i"], p["b_h"],
                                 p["W_d"] if has_vd else None, p["b_d"] if has_vd else None,
                                 V_d if has_vd else None, depth=mp.depth, act=act, slope=slope,
                                 slope_t=slope_t, undirected=mp.undirected, keep=False, max_level=max_level,
                                 wcache=mp.__dict__.setdefault("_dmpnn_wcache", {}))
        mp.__dict__["_dmpnn_last"] = st  # (nn.py builds the replay state of the steady inference path from it)
        return out

    # ---- rows route: kernels for every contraction / segment op, torch modules in between ----
    if grad:
        from .backward import linear_fn, message_fn, aggregate_fn
    else:
        linear_fn = lambda A1, W, b, A2=None, gather=None, n_rows=None, Cadd=None: engine.linear(
            A1, W, b, A2=A2, gather1=gather, n_rows=n_rows, Cadd=Cadd)
        message_fn = lambda plan_, H: engine.message(plan_, H)
        aggregate_fn = lambda plan_, H: engine.aggregate(plan_, H)
    tau, drop = mp.tau, mp.dropout
    rev = None
    H0 = linear_fn(V, p["W_i"], p["b_i"], A2=E, gather=plan.src32, n_rows=plan.n_edges)
    H = tau(H0)
    for _ in range(1, mp.depth):
        if mp.undirected:
            if rev is None:
                rev = plan.rev64
            H = (H + H[rev]) / 2
        M = message_fn(plan, H)
        H = drop(tau(linear_fn(M, p["W_h"], p["b_h"], Cadd=H0)))
    Mv = aggregate_fn(plan, H)
    Hv = drop(tau(linear_fn(V, p["W_o"], p["b_o"], A2=Mv)))
    if has_vd:
        Hv = drop(linear_fn(Hv, p["W_d"], p["b_d"], A2=V_d))
    return Hv
