"""f2, second half: the mol-atom-bond blocks ``chemprop.nn.MABBondMessagePassing`` /
``MABAtomMessagePassing`` (``nn/message_passing/mol_atom_bond.py:16-388``) on the HIP kernels.

They are the bond / atom message-passing blocks with TWO read-outs of the same depth loop
(``mol_atom_bond.py:266-282``):

    H_v = vertex_finalize(segsum_dst H, V, V_d)      (:174-219 — the bond block's finalize)
    H_e = edge_finalize(H, E, E_d)                   (:221-264 — dropout(tau(W_eo [E || H])), then W_ed [. || E_d])

either of which can be switched off (``return_vertex_embeddings`` / ``return_edge_embeddings``; the switched-off
read-out has no ``W_vo`` / ``W_eo``).  Same constructor arguments, parameter names and shapes, ``state_dict`` keys,
construction order (-> same RNG stream -> same initial weights) and error behaviour as the reference.

Two routes, every contraction / segment reduction a HIP kernel in both:

* **tile** (bond variant, built-in activation, dropout inactive, no grad, depth >= 2, both ``W_vo`` present):
  ONE ``dmpnn_forward`` with ``DMPNN_F_KEEP`` — the whole-forward tile kernel leaves ``H^(T-1)`` in its kept slot
  and returns ``H_v`` — plus one ``dmpnn_linear_fwd`` with the activation fused for the edge read-out;
* **rows**: the per-step kernels (``dmpnn_linear_fwd`` / ``dmpnn_message_fwd`` / ``dmpnn_aggregate_fwd`` /
  ``dmpnn_gather_rows``) chained from Python with ``tau`` / dropout as torch modules between them and the autograd
  wrappers of ``backward.py`` (training, custom activations, active dropout, the atom variant).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from . import engine
from .nn import (DEFAULT_ATOM_FDIM, DEFAULT_BOND_FDIM, DEFAULT_HIDDEN_DIM, InvalidShapeError, _HParams,
                 classify_activation, get_activation_function)


class _MABMessagePassingBase(nn.Module):
    """``_MABMessagePassingBase`` (mol_atom_bond.py:16-282): construction and the two read-outs."""

    def __init__(self, d_v: int = DEFAULT_ATOM_FDIM, d_e: int = DEFAULT_BOND_FDIM, d_h: int = DEFAULT_HIDDEN_DIM,
                 bias: bool = False, depth: int = 3, dropout: float = 0.0, activation="relu",
                 undirected: bool = False, d_vd: Optional[int] = None, d_ed: Optional[int] = None,
                 V_d_transform: Optional[nn.Module] = None, E_d_transform: Optional[nn.Module] = None,
                 graph_transform: Optional[nn.Module] = None, return_vertex_embeddings: bool = True,
                 return_edge_embeddings: bool = True):
        super().__init__()
        self.hparams = _HParams(d_v=d_v, d_e=d_e, d_h=d_h, bias=bias, depth=depth, dropout=dropout,
                                activation=activation, undirected=undirected, d_vd=d_vd, d_ed=d_ed,
                                return_vertex_embeddings=return_vertex_embeddings,
                                return_edge_embeddings=return_edge_embeddings,
                                V_d_transform=V_d_transform, E_d_transform=E_d_transform,
                                graph_transform=graph_transform, cls=self.__class__)
        self.return_vertex_embeddings = return_vertex_embeddings
        self.return_edge_embeddings = return_edge_embeddings
        # (mol_atom_bond.py:94-96: attribute order W_i, W_h, W_vo, W_vd, W_eo, W_ed; creation order is setup()'s)
        self.W_i, self.W_h, self.W_vo, self.W_vd, self.W_eo, self.W_ed = self.setup(d_v, d_e, d_h, d_vd, d_ed, bias)
        self.depth = depth
        self.undirected = undirected
        self.dropout = nn.Dropout(dropout)
        self.tau = get_activation_function(activation)
        self.V_d_transform = V_d_transform if V_d_transform is not None else nn.Identity()
        self.E_d_transform = E_d_transform if E_d_transform is not None else nn.Identity()
        self.graph_transform = graph_transform if graph_transform is not None else nn.Identity()

    def setup(self, d_v, d_e, d_h, d_vd, d_ed, bias):
        raise NotImplementedError

    @property
    def output_dims(self) -> tuple[Optional[int], Optional[int]]:
        """(vertex, edge) embedding widths, ``None`` for a switched-off read-out (mol_atom_bond.py:105-119)."""
        v = None if not self.return_vertex_embeddings else (
            self.W_vd.out_features if self.W_vd is not None else self.W_vo.out_features)
        e = None if not self.return_edge_embeddings else (
            self.W_ed.out_features if self.W_ed is not None else self.W_eo.out_features)
        return v, e

    def forward(self, bmg, V_d: Optional[Tensor] = None, E_d: Optional[Tensor] = None):
        return mab_forward(self, bmg, V_d, E_d)


class MABBondMessagePassing(_MABMessagePassingBase):
    """``chemprop.nn.MABBondMessagePassing`` (mol_atom_bond.py:284-335): messages along directed bonds."""

    atom_messages = False

    def setup(self, d_v, d_e, d_h, d_vd, d_ed, bias):  # mol_atom_bond.py:318-335 (same creation order)
        W_i = nn.Linear(d_v + d_e, d_h, bias)
        W_h = nn.Linear(d_h, d_h, bias)
        W_vo = nn.Linear(d_v + d_h, d_h) if self.return_vertex_embeddings else None
        W_eo = nn.Linear(d_e + d_h, d_h) if self.return_edge_embeddings else None
        W_vd = nn.Linear(d_h + d_vd, d_h + d_vd) if d_vd else None
        W_ed = nn.Linear(d_h + d_ed, d_h + d_ed) if d_ed else None
        return W_i, W_h, W_vo, W_vd, W_eo, W_ed


class MABAtomMessagePassing(_MABMessagePassingBase):
    """``chemprop.nn.MABAtomMessagePassing`` (mol_atom_bond.py:338-388): messages along atoms."""

    atom_messages = True

    def setup(self, d_v, d_e, d_h, d_vd, d_ed, bias):  # mol_atom_bond.py:371-388
        W_i = nn.Linear(d_v, d_h, bias)
        W_h = nn.Linear(d_e + d_h, d_h, bias)
        W_vo = nn.Linear(d_v + d_h, d_h) if self.return_vertex_embeddings else None
        W_eo = nn.Linear(d_e + d_h, d_h) if self.return_edge_embeddings else None
        W_vd = nn.Linear(d_h + d_vd, d_h + d_vd) if d_vd else None
        W_ed = nn.Linear(d_h + d_ed, d_h + d_ed) if d_ed else None
        return W_i, W_h, W_vo, W_vd, W_eo, W_ed


def _check_descriptors(name: str, X: Optional[Tensor], transform, W_x, W_o, n_rows: int):
    """The reference applies the transform, then lets ``cat`` + ``Linear`` raise on a bad shape and converts that
    RuntimeError into InvalidShapeError (mol_atom_bond.py:208-217,250-262).  Same outcome, checked up front; a
    descriptor tensor given to a block built without ``d_vd`` / ``d_ed`` fails there with an AttributeError-turned
    TypeError ('NoneType' is not callable) — here it is the same InvalidShapeError."""
    if X is None:
        return None
    X = transform(X)
    d = (W_x.in_features - W_o.out_features) if (W_x is not None and W_o is not None) else None
    if d is None or X.dim() != 2 or X.shape[0] != n_rows or X.shape[1] != d:
        raise InvalidShapeError(name, X.shape, [n_rows, d if d is not None else 0])
    return X


def _tile_route_ok(mp, bmg, V_d, E_d) -> bool:
    if mp.atom_messages or mp.W_vo is None or mp.depth < 2 or torch.is_grad_enabled() and (
            any(p.requires_grad for p in mp.parameters())):
        return False
    if mp.training and mp.dropout.p > 0:
        return False
    act, _, _ = classify_activation(mp.tau)
    if act == "custom":
        return False
    return True


def _tile_train_ok(mp, bmg, V_d) -> bool:
    """A TRAINING forward the tile kernels take (round 4): the vertex read-out present (it is the tile kernel's finalize), built-in
    activation, no active dropout, directed; the edge read-out needs depth >= 2 (it reads a kept H^(depth-1)); atom messages: no V_d,
    an even d_e of 2 .. 16.  Molecules beyond the tile, and everything else: the rows route."""
    if not (torch.is_grad_enabled() and any(p.requires_grad for p in mp.parameters())):
        return False
    if mp.W_vo is None or mp.undirected or (mp.training and mp.dropout.p > 0):
        return False
    if classify_activation(mp.tau)[0] in ("custom", "prelu"):
        return False
    if mp.return_edge_embeddings and mp.depth < 2:
        return False
    n_e, d_e = int(bmg.E.shape[0]), int(bmg.E.shape[1])
    if n_e == 0:
        return False
    if mp.atom_messages and (V_d is not None or not (2 <= d_e <= 16 and d_e % 2 == 0)):
        return False
    if not mp.atom_messages:
        # bond messages: a molecule beyond the tile runs the kernels' generic fp32 path (correct, slow) — a speed question only
        return getattr(bmg, "oversize", None) is not True
    # atom messages: such a molecule is NaN on these training routes, forward and every gradient (the generic path knows bond messages
    # only, dmpnn_mega16_impl.hpp) — the host must KNOW, not guess: counted on the device for a foreign batch (round-4 ADVICE)
    from .nn import batch_oversize

    return batch_oversize(bmg, len(bmg) if hasattr(bmg, "__len__") else 0) is False


def mab_forward(mp, bmg, V_d: Optional[Tensor] = None, E_d: Optional[Tensor] = None):
    """``_MABMessagePassingBase.forward`` (mol_atom_bond.py:266-282) -> ``(H_v | None, H_e | None)``."""
    from .backward import aggregate_fn, gather_src_fn, linear_fn, message_fn

    bmg = mp.graph_transform(bmg)
    engine._require_device(bmg.V, "bmg.V")
    if mp.W_i.weight.device != bmg.V.device:
        raise RuntimeError(f"module is on {mp.W_i.weight.device} but the batch is on {bmg.V.device}")
    n_atoms, n_edges = int(bmg.V.shape[0]), int(bmg.E.shape[0])
    want_v, want_e = mp.return_vertex_embeddings, mp.return_edge_embeddings
    V_d = _check_descriptors("V_d", V_d, mp.V_d_transform, mp.W_vd, mp.W_vo, n_atoms) if want_v else None
    E_d = _check_descriptors("E_d", E_d, mp.E_d_transform, mp.W_ed, mp.W_eo, n_edges) if want_e else None
    for name, t in (("bmg.V", bmg.V), ("bmg.E", bmg.E), ("V_d", V_d), ("E_d", E_d)):
        if t is not None and t.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError(f"chemprop_amd: gradient w.r.t. `{name}` is not provided by the engine "
                                      "(the reference never asks for it: features are data)")
    V, E = bmg.V, bmg.E
    tau, drop = mp.tau, mp.dropout

    if _tile_route_ok(mp, bmg, V_d, E_d):
        from .nn import _route

        plan = engine.GraphPlan.from_bmg(bmg)
        act, slope, slope_t = classify_activation(mp.tau)
        has_vd = V_d is not None
        # the same route policy as the bond block (nn.py: _route): the first batches of a module are checked on the plan's
        # verdict — an oversize molecule switches the tile kernel off for the module, an unusual graph takes the general route
        n_mols = len(bmg) if hasattr(bmg, "__len__") else 0
        plan.oversize = getattr(bmg, "oversize", None)
        level = _route(mp, plan, n_mols, getattr(bmg, "batch", None))
        if plan.oversize is True:  # (host knowledge of the batching code: per-step routes for this batch)
            level = min(level, 1)
        Hv, st = engine.forward(plan, V, E, mp.W_i.weight, mp.W_h.weight, mp.W_vo.weight, mp.W_vo.bias,
                                mp.W_i.bias, mp.W_h.bias, mp.W_vd.weight if has_vd else None,
                                mp.W_vd.bias if has_vd else None, V_d if has_vd else None, depth=mp.depth, act=act,
                                slope=slope, slope_t=slope_t, undirected=mp.undirected, keep=True, max_level=level)
        H_v = Hv if want_v else None
        H_e = None
        if want_e:
            H = st.Hs[mp.depth - 2][:, :mp.W_h.out_features] if n_edges else V.new_zeros(0, mp.W_h.out_features)
            if n_edges and st.route in ("mega", "mega16", "fused"):  # kept rows are the plan's CSR rows (row i = edge perm[i])
                H = engine.gather_rows(H, plan.inv32)
            H_e = engine.linear(E, mp.W_eo.weight, mp.W_eo.bias, A2=H, act=act, slope=slope, slope_t=slope_t)
            if E_d is not None:
                H_e = engine.linear(H_e, mp.W_ed.weight, mp.W_ed.bias, A2=E_d)
        return H_v, H_e

    if _tile_train_ok(mp, bmg, V_d):
        # ---- round 4: TRAINING on the tile kernels — the block's forward is ONE launch that keeps what the backward tile kernel reads,
        # the kept H^(depth-1) leaves it as a second output for the edge read-out (a row kernel under autograd) and that read-out's
        # gradient enters the backward tile kernel beside the vertex one (dmpnn_bwd_args.g_edge).  Both variants (DMPNN_F_ATOM). ----
        import types

        from .backward import FusedMP
        from .nn import _route, _training_plan_kind

        act, slope, slope_t = classify_activation(mp.tau)
        has_vd = V_d is not None
        shim = types.SimpleNamespace(undirected=mp.undirected, W_d=mp.W_vd if has_vd else None, tau=mp.tau, training=mp.training, dropout=mp.dropout, depth=mp.depth,
                                     W_i=mp.W_i, W_h=mp.W_h, W_o=mp.W_vo, _dmpnn_batches_checked=getattr(mp, "_dmpnn_batches_checked", 0),
                                     _dmpnn_no_mega=getattr(mp, "_dmpnn_no_mega", False))
        light = _training_plan_kind(shim, bmg)
        plan = engine.GraphPlan.from_bmg(bmg, light=light, launch="defer" if light == "tiles" else True)
        n_mols = len(bmg) if hasattr(bmg, "__len__") else 0
        plan.oversize = getattr(bmg, "oversize", None)
        if _route(mp, plan, n_mols, getattr(bmg, "batch", None)) >= 2:
            try:
                res = FusedMP.apply(mp, plan, V, E, V_d if has_vd else None, act, slope, (slope_t, 2, None, mp.atom_messages, want_e),
                                    mp.W_i.weight, mp.W_i.bias, mp.W_h.weight, mp.W_h.bias, mp.W_vo.weight, mp.W_vo.bias,
                                    mp.W_vd.weight if has_vd else None, mp.W_vd.bias if has_vd else None)
                plan.ensure_launched()
                H_v, H = res if want_e else (res, None)
                H_e = None
                if want_e:
                    H_e = drop(tau(linear_fn(E, mp.W_eo.weight, mp.W_eo.bias, A2=H)))
                    if E_d is not None:
                        H_e = drop(linear_fn(H_e, mp.W_ed.weight, mp.W_ed.bias, A2=E_d))
                mp.__dict__["_dmpnn_route"] = "mega16/atom" if mp.atom_messages else "mega16"
                return (H_v if want_v else None), H_e
            except engine.RouteUnavailable:
                pass
        plan.pending = None   # (a deferred K0 that nothing ran: the rows route builds its own full plan)

    # ---- rows route ----
    mp.__dict__["_dmpnn_route"] = "rows"
    plan = engine.GraphPlan.from_bmg(bmg)
    nE = plan.n_edges
    rev = None
    if mp.atom_messages:  # mixins.py:21-30
        H0 = linear_fn(V, mp.W_i.weight, mp.W_i.bias, gather=plan.src32, n_rows=nE)
        if mp.depth > 1 and nE:
            with torch.no_grad():
                ME = engine.gather_rows(engine.aggregate(plan, E), plan.src32)  # [E, d_e]: constant over the loop
    else:                 # mixins.py:8-18
        H0 = linear_fn(V, mp.W_i.weight, mp.W_i.bias, A2=E, gather=plan.src32, n_rows=nE)
    H = tau(H0)
    for _ in range(1, mp.depth):
        if mp.undirected:
            if rev is None:
                rev = plan.rev64
            H = (H + H[rev]) / 2
        if not nE:
            H = drop(tau(H0))
        elif mp.atom_messages:
            MH = gather_src_fn(plan, aggregate_fn(plan, H))
            H = drop(tau(linear_fn(MH, mp.W_h.weight, mp.W_h.bias, A2=ME, Cadd=H0)))
        else:
            H = drop(tau(linear_fn(message_fn(plan, H), mp.W_h.weight, mp.W_h.bias, Cadd=H0)))
    H_v = H_e = None
    if want_v:
        Mv = aggregate_fn(plan, H)
        H_v = drop(tau(linear_fn(V, mp.W_vo.weight, mp.W_vo.bias, A2=Mv)))
        if V_d is not None:
            H_v = drop(linear_fn(H_v, mp.W_vd.weight, mp.W_vd.bias, A2=V_d))
    if want_e:
        H_e = drop(tau(linear_fn(E, mp.W_eo.weight, mp.W_eo.bias, A2=H)))
        if E_d is not None:
            H_e = drop(linear_fn(H_e, mp.W_ed.weight, mp.W_ed.bias, A2=E_d))
    return H_v, H_e
