"""``torch.export`` / ``torch.compile`` of a model that contains the engine's message-passing block.

The reference's encoders are plain tensor programs and export as such (``tests/integration/test_export.py:17-46``: a whole
``MPNN`` exported on one batch with dynamic ``num_atoms`` / ``num_edges`` and run on another, ``E == 0`` included;
``tests/unit/nn/test_message_passing.py`` does the same per encoder).  The engine's forward hands raw device pointers to a C
library, which no tracer can follow — so under tracing the whole ``BondMessagePassing.forward`` (``base.py:196-212``) is ONE
opaque operator, ``chemprop_amd::bond_message_passing``, with a shape-only fake implementation for the tracer.  The exported
program holds that operator; running it runs the HIP kernels (no tensor-op fallback anywhere: a missing library fails
loudly as everywhere else).

The operator is stateless: it builds the plan (K0) per call and reads the plan's on-device verdict before choosing the
route — one host sync per call, what ``DMPNN_VALIDATE=always`` does for a module (``DMPNN_VALIDATE=never`` trusts the batch
to be a molecular graph, as for modules).  Inference only (no autograd formula is registered).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import _lib, engine

__all__ = ["bond_message_passing_op", "traced_forward"]


@torch.library.custom_op("chemprop_amd::bond_message_passing", mutates_args=())
def bond_message_passing_op(V: Tensor, E: Tensor, edge_index: Tensor, rev_edge_index: Tensor, W_i: Tensor, W_h: Tensor,
                            W_o: Tensor, b_o: Tensor, b_i: Optional[Tensor], b_h: Optional[Tensor], W_d: Optional[Tensor],
                            b_d: Optional[Tensor], V_d: Optional[Tensor], slope_t: Optional[Tensor], depth: int, act: str,
                            slope: float, undirected: bool) -> Tensor:
    engine._require_device(V, "bmg.V")
    n_atoms = int(V.shape[0])
    plan = engine.GraphPlan(edge_index, rev_edge_index, n_atoms)
    level = 0 if undirected else 2
    if level and _lib.opt("DMPNN_VALIDATE", "first") != "never":
        hdr = plan.header()  # (host sync: the plan's verdict on this batch)
        if hdr[0] & 7:
            level = 0
        elif (hdr[0] & 8) or hdr[8] > 0:  # no piece tiles / a molecule beyond the tile: the per-step routes
            level = 1
    out, _ = engine.forward(plan, V, E, W_i, W_h, W_o, b_o, b_i, b_h, W_d, b_d, V_d, depth=depth, act=act, slope=slope,
                            slope_t=slope_t, undirected=undirected, keep=False, max_level=level)
    return out


@bond_message_passing_op.register_fake
def _(V, E, edge_index, rev_edge_index, W_i, W_h, W_o, b_o, b_i, b_h, W_d, b_d, V_d, slope_t, depth, act, slope, undirected):
    d_out = W_d.shape[0] if (W_d is not None and V_d is not None) else W_o.shape[0]
    return V.new_empty((V.shape[0], d_out), dtype=torch.float32)


def traced_forward(mp, bmg, V_d: Optional[Tensor] = None) -> Tensor:
    """``_MessagePassingBase.forward`` (base.py:196-212) while a tracer is recording: the transforms stay tensor programs,
    the message passing is the one operator above."""
    from .nn import InvalidShapeError, classify_activation

    bmg = mp.graph_transform(bmg)
    act, slope, slope_t = classify_activation(mp.tau)
    if act == "custom" or (mp.training and mp.dropout.p > 0):
        raise NotImplementedError("chemprop_amd: torch.export / torch.compile of the message passing needs a built-in activation "
                                  "(relu, leakyrelu, prelu, tanh, elu) and no active dropout")
    has_vd = mp.W_d is not None and V_d is not None
    if V_d is not None:
        V_d = mp.V_d_transform(V_d)
        d_vd = (mp.W_d.in_features - mp.W_o.out_features) if mp.W_d is not None else None
        if mp.W_d is None or V_d.dim() != 2 or V_d.shape[1] != d_vd:
            raise InvalidShapeError("V_d", V_d.shape, [bmg.V.shape[0], d_vd if d_vd is not None else 0])
    return bond_message_passing_op(
        bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias,
        mp.W_i.bias, mp.W_h.bias, mp.W_d.weight if has_vd else None, mp.W_d.bias if has_vd else None, V_d if has_vd else None,
        slope_t, int(mp.depth), act, float(slope), bool(mp.undirected))
