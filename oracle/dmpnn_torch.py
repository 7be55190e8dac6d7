"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement of the reference's D-MPNN bond-message-passing forward as a sequence of the *same
ATen operations in the same order* as the reference issues them, so that on CPU torch it reproduces
the executed reference bit-for-bit (pinned by ``tests/test_oracle.py`` against the golden vectors
frozen from the executed reference by ``tests/golden/make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  It is also the timed CPU baseline (``cpu_baseline.kind == "port"``): because it issues
the reference's own op sequence (index / scatter_reduce_ / sub / cat / linear) it costs what the
reference costs on the same host cores.

Reference lines restated (all under ``/root/reference/chemprop``):
  * ``nn/message_passing/mixins.py:8-9``    initialize   H0 = W_i([V[src] || E])
  * ``nn/message_passing/mixins.py:11-18``  message      M = scatter_sum_dst(H)[src] - H[rev]
  * ``nn/message_passing/base.py:135-141``  update       H = dropout(tau(H0 + W_h(M)))
  * ``nn/message_passing/base.py:180-194``  finalize     tau(W_o([V || M_v])) [, W_d([H || V_d])]
  * ``nn/message_passing/base.py:196-212``  forward      the depth loop + final atom aggregation
  * ``nn/utils.py:43-53``                   activations  (LeakyReLU slope 0.1, PReLU 0.25 init)
Dropout is the identity here (the oracle is evaluated with p = 0 / eval mode).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import torch
import torch.nn.functional as F
from torch import Tensor


@dataclass
class MPWeights:
    """Parameters of one BondMessagePassing block in ``nn.Linear`` layout ``[out, in]``
    (``base.py:238-251``)."""

    W_i: Tensor
    W_h: Tensor
    W_o: Tensor
    b_o: Tensor
    b_i: Optional[Tensor] = None
    b_h: Optional[Tensor] = None
    W_d: Optional[Tensor] = None
    b_d: Optional[Tensor] = None

    @classmethod
    def from_module(cls, m) -> "MPWeights":
        g = lambda lin, n: None if lin is None or getattr(lin, n) is None else getattr(lin, n).detach()
        return cls(g(m.W_i, "weight"), g(m.W_h, "weight"), g(m.W_o, "weight"), g(m.W_o, "bias"),
                   g(m.W_i, "bias"), g(m.W_h, "bias"), g(m.W_d, "weight"), g(m.W_d, "bias"))


def activation_fn(name: str, prelu_weight: Optional[Tensor] = None) -> Callable[[Tensor], Tensor]:
    """nn/utils.py:43-53."""
    name = str(name).lower()
    if name == "relu":
        return torch.relu
    if name == "leakyrelu":
        return lambda x: F.leaky_relu(x, 0.1)
    if name == "prelu":
        w = prelu_weight if prelu_weight is not None else torch.tensor([0.25])
        return lambda x: F.prelu(x, w)
    if name == "tanh":
        return torch.tanh
    if name == "elu":
        return F.elu
    if name in ("identity", "none"):
        return lambda x: x
    raise ValueError(f"unknown activation {name!r}")


def segment_sum_dst(H: Tensor, dst: Tensor, n_atoms: int) -> Tensor:
    """``zeros(V,h).scatter_reduce_(0, dst.repeat, H, "sum", include_self=False)``
    (mixins.py:12-15, base.py:208-211) — sequential over edges on CPU, i.e. each atom's incoming
    rows are summed in increasing edge id."""
    index = dst.unsqueeze(1).repeat(1, H.shape[1])
    return torch.zeros(n_atoms, H.shape[1], dtype=H.dtype, device=H.device).scatter_reduce_(
        0, index, H, reduce="sum", include_self=False)


def initialize(V: Tensor, E: Tensor, src: Tensor, w: MPWeights) -> Tensor:
    return F.linear(torch.cat([V[src], E], dim=1), w.W_i, w.b_i)


def message(H: Tensor, src: Tensor, dst: Tensor, rev: Tensor, n_atoms: int) -> Tensor:
    M_all = segment_sum_dst(H, dst, n_atoms)[src]
    M_rev = H[rev]
    return M_all - M_rev


def update(M: Tensor, H0: Tensor, w: MPWeights, tau) -> Tensor:
    Ht = F.linear(M, w.W_h, w.b_h)
    return tau(H0 + Ht)


def finalize(Mv: Tensor, V: Tensor, V_d: Optional[Tensor], w: MPWeights, tau) -> Tensor:
    H = tau(F.linear(torch.cat((V, Mv), dim=1), w.W_o, w.b_o))
    if V_d is not None:
        H = F.linear(torch.cat((H, V_d), dim=1), w.W_d, w.b_d)  # no activation (base.py:187-188)
    return H


def forward(V: Tensor, E: Tensor, edge_index: Tensor, rev: Tensor, w: MPWeights, depth: int = 3,
            activation="relu", undirected: bool = False, V_d: Optional[Tensor] = None,
            prelu_weight: Optional[Tensor] = None, return_intermediates: bool = False):
    """base.py:196-212 with Identity graph_transform / V_d_transform and dropout 0."""
    tau = activation if callable(activation) else activation_fn(activation, prelu_weight)
    src, dst = edge_index[0], edge_index[1]
    n_atoms = V.shape[0]
    H0 = initialize(V, E, src, w)
    H = tau(H0)
    inter = {"H0": H0, "H": [H], "M": []}
    for _ in range(1, depth):
        if undirected:
            H = (H + H[rev]) / 2
        M = message(H, src, dst, rev, n_atoms)
        H = update(M, H0, w, tau)
        inter["M"].append(M)
        inter["H"].append(H)
    Mv = segment_sum_dst(H, dst, n_atoms)
    out = finalize(Mv, V, V_d, w, tau)
    if return_intermediates:
        inter["Mv"] = Mv
        return out, inter
    return out


def forward_bmg(bmg, w: MPWeights, **kw):
    return forward(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, w, **kw)


# ---- f2: AtomMessagePassing (base.py:254-289 with the atom mixin, mixins.py:21-30) --------------------------
def atom_initialize(V: Tensor, src: Tensor, w: MPWeights) -> Tensor:
    """mixins.py:22-23: ``W_i(V[src])``."""
    return F.linear(V[src], w.W_i, w.b_i)


def atom_message(H: Tensor, E: Tensor, src: Tensor, dst: Tensor, n_atoms: int) -> Tensor:
    """mixins.py:25-30: segment-sum of ``[H || E]`` over incoming edges, gathered at the source atom."""
    HE = torch.cat((H, E), dim=1)
    return segment_sum_dst(HE, dst, n_atoms)[src]


def atom_forward(V: Tensor, E: Tensor, edge_index: Tensor, rev: Tensor, w: MPWeights, depth: int = 3,
                 activation="relu", undirected: bool = False, V_d: Optional[Tensor] = None,
                 prelu_weight: Optional[Tensor] = None, return_intermediates: bool = False):
    """base.py:196-212 for the atom variant (dropout 0, Identity transforms)."""
    tau = activation if callable(activation) else activation_fn(activation, prelu_weight)
    src, dst = edge_index[0], edge_index[1]
    n_atoms = V.shape[0]
    H0 = atom_initialize(V, src, w)
    H = tau(H0)
    inter = {"H0": H0, "M": []}
    for _ in range(1, depth):
        if undirected:
            H = (H + H[rev]) / 2
        M = atom_message(H, E, src, dst, n_atoms)
        H = update(M, H0, w, tau)
        inter["M"].append(M)
    Mv = segment_sum_dst(H, dst, n_atoms)
    out = finalize(Mv, V, V_d, w, tau)
    return (out, inter) if return_intermediates else out


# ---- f2: the mol-atom-bond blocks (mol_atom_bond.py:16-388): the same depth loops with two read-outs ---------
@dataclass
class MABWeights:
    """Parameters of one MAB block in ``nn.Linear`` layout (mol_atom_bond.py:318-335,371-388); a switched-off
    read-out has no ``W_vo`` / ``W_eo``."""

    W_i: Tensor
    W_h: Tensor
    W_vo: Optional[Tensor] = None
    b_vo: Optional[Tensor] = None
    W_eo: Optional[Tensor] = None
    b_eo: Optional[Tensor] = None
    b_i: Optional[Tensor] = None
    b_h: Optional[Tensor] = None
    W_vd: Optional[Tensor] = None
    b_vd: Optional[Tensor] = None
    W_ed: Optional[Tensor] = None
    b_ed: Optional[Tensor] = None

    @classmethod
    def from_state_dict(cls, sd) -> "MABWeights":
        g = lambda k: sd.get(k)
        return cls(g("W_i.weight"), g("W_h.weight"), g("W_vo.weight"), g("W_vo.bias"), g("W_eo.weight"), g("W_eo.bias"),
                   g("W_i.bias"), g("W_h.bias"), g("W_vd.weight"), g("W_vd.bias"), g("W_ed.weight"), g("W_ed.bias"))


def mab_forward(V: Tensor, E: Tensor, edge_index: Tensor, rev: Tensor, w: MABWeights, atom_messages: bool,
                depth: int = 3, activation="relu", undirected: bool = False, V_d: Optional[Tensor] = None,
                E_d: Optional[Tensor] = None, prelu_weight: Optional[Tensor] = None):
    """mol_atom_bond.py:266-282 (dropout 0, Identity transforms) -> (H_v | None, H_e | None)."""
    tau = activation if callable(activation) else activation_fn(activation, prelu_weight)
    src, dst = edge_index[0], edge_index[1]
    n_atoms = V.shape[0]
    wl = MPWeights(W_i=w.W_i, W_h=w.W_h, W_o=w.W_vo, b_o=w.b_vo, b_i=w.b_i, b_h=w.b_h, W_d=w.W_vd, b_d=w.b_vd)
    H0 = atom_initialize(V, src, wl) if atom_messages else initialize(V, E, src, wl)
    H = tau(H0)
    for _ in range(1, depth):
        if undirected:
            H = (H + H[rev]) / 2
        M = atom_message(H, E, src, dst, n_atoms) if atom_messages else message(H, src, dst, rev, n_atoms)
        H = update(M, H0, wl, tau)
    H_v = H_e = None
    if w.W_vo is not None:  # vertex_finalize (:174-219): dropout after W_vd only — no activation there
        H_v = finalize(segment_sum_dst(H, dst, n_atoms), V, V_d, wl, tau)
    if w.W_eo is not None:  # edge_finalize (:221-264)
        H_e = tau(F.linear(torch.cat((E, H), dim=1), w.W_eo, w.b_eo))
        if E_d is not None:
            H_e = F.linear(torch.cat((H_e, E_d), dim=1), w.W_ed, w.b_ed)
    return H_v, H_e
