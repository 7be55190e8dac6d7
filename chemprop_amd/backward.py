"""autograd wiring of the engine (training: chemprop/models/model.py:148-161 calls loss.backward()
through the block).  Gradients are produced by the K6 HIP kernels; torch only routes them.

* :class:`FusedMP`       one ``dmpnn_forward`` (workspace kept) + one ``dmpnn_backward`` per step.
* ``linear_fn`` / ``message_fn`` / ``aggregate_fn``  row-level Functions for the *rows* route
  (arbitrary activation module, learnable PReLU, active dropout run in torch between kernels).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import engine

_PARAM_ORDER = ("W_i", "b_i", "W_h", "b_h", "W_o", "b_o", "W_d", "b_d")


class FusedMP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mp, plan, V, E, V_d, act, slope, slope_and_route, W_i, b_i, W_h, b_h, W_o, b_o, W_d, b_d):
        slope_t, max_level = slope_and_route[0], slope_and_route[1]
        dropout = slope_and_route[2] if len(slope_and_route) > 2 else None
        atom = bool(slope_and_route[3]) if len(slope_and_route) > 3 else False   # AtomMessagePassing on the tile kernels (DMPNN_F_ATOM)
        # the mol-atom-bond blocks (mab.py): the kept H^(depth-1) is a second OUTPUT, for the edge read-out — in the caller's edge order —
        # and its gradient a second input of the backward tile kernel (dmpnn_bwd_args.g_edge); tile-kernel forwards only, depth >= 2
        want_edge = bool(slope_and_route[4]) if len(slope_and_route) > 4 else False
        has_vd = V_d is not None and W_d is not None
        out, st = engine.forward(plan, V, E, W_i, W_h, W_o, b_o, b_i, b_h, W_d if has_vd else None,
                                 b_d if has_vd else None, V_d if has_vd else None, depth=mp.depth, act=act,
                                 slope=slope, slope_t=slope_t, undirected=mp.undirected, keep=True, max_level=max_level, dropout=dropout,
                                 atom=atom, keep_bits=not want_edge)
        ctx.st = st
        ctx.has_vd = has_vd
        ctx.mp = mp
        ctx.edge_rows = None
        if want_edge:
            if st.route != "mega16" or mp.depth < 2:
                raise engine.RouteUnavailable("the edge states as an output: a training forward of the tile kernel, depth >= 2")
            ctx.set_materialize_grads(False)
            H = st.Hs[mp.depth - 2][:, :int(W_h.shape[0])]
            ctx.edge_rows = "caller" if plan.tiles_only else "csr"   # (a tile plan keeps its tensors in the caller's edge order)
            if ctx.edge_rows == "csr":
                H = engine.gather_rows(H, plan.inv32)
            return out, H
        return out

    @staticmethod
    def backward(ctx, gout, g_edge=None):
        st = ctx.st
        if st is None:
            raise RuntimeError("chemprop_amd: backward through this BondMessagePassing forward a second time — the kept workspace "
                               "is released after the first backward (run the forward again; retain_graph=True is honoured only "
                               "with DMPNN_KEEP_WORKSPACE=1)")
        need = {k: ctx.needs_input_grad[8 + i] for i, k in enumerate(_PARAM_ORDER)}
        if not ctx.has_vd:
            need["W_d"] = need["b_d"] = False
        # a flat gradient buffer registered on the module (distributed.GradSync): where a parameter's .grad IS its view of
        # that buffer, the kernels write straight into it and autograd is told there is nothing to accumulate (None).  The
        # kernels OVERWRITE, so only the FIRST backward through a view since the last zero_grad() / allreduce() / optimizer
        # step may do that: a block that runs twice in one step (MulticomponentMessagePassing(shared=True), two forwards
        # before one backward, gradient accumulation) hands its later gradients to autograd, which adds them into the view.
        mp = ctx.mp
        views = mp.__dict__.get("_dmpnn_grad_views") if mp is not None else None
        written = mp.__dict__.get("_dmpnn_grad_written") if mp is not None else None
        direct = {}
        if views:
            for k, v in views.items():
                lin = getattr(mp, "W_" + k[2:], None)
                prm = None if lin is None else (lin.weight if k[0] == "W" else lin.bias)
                if (prm is not None and prm.grad is not None and prm.grad.data_ptr() == v.data_ptr() and need.get(k)
                        and (written is None or v.data_ptr() not in written)):
                    direct[k] = v
        if ctx.edge_rows is not None:
            if gout is None:   # (only the edge read-out reached the loss)
                gout = torch.zeros_like(st.out)
            if g_edge is not None and ctx.edge_rows == "csr":
                g_edge = engine.gather_rows(g_edge.contiguous(), st.plan.perm32)
        grads = engine.backward(st, gout.contiguous(), need, out=direct, g_edge=g_edge)
        if engine._lib.opt("DMPNN_KEEP_WORKSPACE", "0") != "1":
            ctx.st = None  # release the kept workspace
        # (a view the engine did not take — wrong dtype / layout — got a fresh tensor instead: that one goes to autograd)
        took = {k for k, v in direct.items() if grads.get(k) is v}
        if written is not None:
            written.update(direct[k].data_ptr() for k in took)
        return (None,) * 8 + tuple(None if k in took else grads[k] for k in _PARAM_ORDER)


class _Linear(torch.autograd.Function):
    """``[A1[gather] || A2] @ W.T + b + Cadd``; grads for W, b, A1 (ungathered only), A2, Cadd."""

    @staticmethod
    def forward(ctx, A1, W, b, A2, gather, n_rows, Cadd):
        out = engine.linear(A1, W, b, A2=A2, gather1=gather, n_rows=n_rows, Cadd=Cadd)
        ctx.save_for_backward(A1, W, A2, gather)
        ctx.has_b = b is not None
        ctx.has_cadd = Cadd is not None
        return out

    @staticmethod
    def backward(ctx, gZ):
        A1, W, A2, gather = ctx.saved_tensors
        gZ = gZ.contiguous()
        need_A1, need_W, need_b, need_A2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        gW = gb = gA1 = gA2 = None
        if need_W or (need_b and ctx.has_b):
            gW, gb = engine.linear_wgrad(gZ, A1, A2, gather, want_bias=ctx.has_b and need_b)
            if not need_W:
                gW = None
        if need_A1 or need_A2:
            if gather is not None and need_A1:
                raise NotImplementedError("gradient w.r.t. gathered features is not provided")
            gA = engine.linear(gZ, W.t().contiguous(), None)  # data gradient: the same MFMA kernel on W^T
            K1 = A1.shape[1]
            gA1 = gA[:, :K1] if need_A1 else None
            gA2 = gA[:, K1:] if (need_A2 and A2 is not None) else None
        gCadd = gZ if (ctx.has_cadd and ctx.needs_input_grad[6]) else None
        return gA1, gW, gb, gA2, None, None, gCadd


class _Message(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, H):
        ctx.plan = plan
        return engine.message(plan, H)

    @staticmethod
    def backward(ctx, gM):
        return None, engine.message_bwd(ctx.plan, gM.contiguous())


class _Aggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, H):
        ctx.plan = plan
        return engine.aggregate(plan, H)

    @staticmethod
    def backward(ctx, gMv):
        return None, engine.aggregate_bwd(ctx.plan, gMv.contiguous())


class _GatherSrc(torch.autograd.Function):
    """``M[e] = S[src(e)]`` (mixins.py:30); transpose: ``gS[v] = sum_{e: src(e) = v} gM[e]`` which, on a
    symmetric graph (src(e) = dst(rev e)), is the incoming-edge segment sum of the rows taken through rev."""

    @staticmethod
    def forward(ctx, plan, S):
        ctx.plan = plan
        return engine.gather_rows(S, plan.src32)

    @staticmethod
    def backward(ctx, gM):
        plan = ctx.plan
        gS = engine.aggregate(plan, engine.gather_rows(gM.contiguous(), plan.rev32))
        # this transpose holds on a symmetric graph only (src(e) == dst(rev e)); the plan knows (header bit 0, on the
        # device): an asymmetric / hand-built graph gets NaN gradients here — loud, never silently wrong — without a host sync
        asym = (plan.buf[0] & 1).bool()
        return None, torch.where(asym, torch.full_like(gS, float("nan")), gS)


def gather_src_fn(plan, S: Tensor) -> Tensor:
    return _GatherSrc.apply(plan, S)


def linear_fn(A1: Tensor, W: Tensor, b: Optional[Tensor], A2: Optional[Tensor] = None,
              gather: Optional[Tensor] = None, n_rows: Optional[int] = None, Cadd: Optional[Tensor] = None) -> Tensor:
    return _Linear.apply(A1, W, b, A2, gather, n_rows, Cadd)


def message_fn(plan, H: Tensor) -> Tensor:
    return _Message.apply(plan, H)


def aggregate_fn(plan, H: Tensor) -> Tensor:
    return _Aggregate.apply(plan, H)
