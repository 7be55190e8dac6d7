"""Host-side mirror of the reference's message-passing interface for the accelerated path.

:class:`BondMessagePassing` has the constructor, attributes, parameter names/shapes (hence
``state_dict`` keys), ``hparams`` and ``forward(bmg, V_d=None)`` contract of
``chemprop.nn.BondMessagePassing`` (``chemprop/nn/message_passing/base.py:16-92,196-251``,
interface ``proto.py:9-34``), so a state dict moves between the two unchanged, and the parity tests
read like the reference's own.  Where chemprop itself is importable, use
:mod:`chemprop_amd.integration` instead: it subclasses the *real* class and overrides only
``forward``.

``forward`` runs on the HIP kernels only.  Inputs on a non-HIP device raise (no CPU fallback).
"""
from __future__ import annotations

import copy
import ctypes as _ctypes
import os
from typing import Optional

import torch
from torch import Tensor, nn

from . import _lib, engine
from .autograd import mp_forward
from .data import BatchMolGraph as _OwnBatch

DEFAULT_ATOM_FDIM, DEFAULT_BOND_FDIM = 72, 14  # chemprop/conf.py:8 (v2 featurizers)
DEFAULT_HIDDEN_DIM = 300  # chemprop/conf.py:9


class InvalidShapeError(ValueError):
    """Mirror of ``chemprop/exceptions.py`` (raised for a mis-shaped ``V_d``, base.py:189-192)."""

    def __init__(self, var_name, received, expected):
        super().__init__(f"arg '{var_name}' has incorrect shape! got: `{tuple(received)}`. "
                         f"expected: `{tuple(expected)}`")


def get_activation_function(activation) -> nn.Module:
    """``chemprop/nn/utils.py:19-55``."""
    if isinstance(activation, nn.Module):
        return activation
    name = str(getattr(activation, "value", activation)).lower()
    if name == "selu":
        return nn.SELU()
    try:
        return {"relu": nn.ReLU, "leakyrelu": lambda: nn.LeakyReLU(0.1), "prelu": nn.PReLU,
                "tanh": nn.Tanh, "elu": nn.ELU}[name]()
    except KeyError:
        raise ValueError(f"unknown activation {activation!r}") from None


def classify_activation(tau: nn.Module) -> tuple[str, float, Optional[Tensor]]:
    """Map a ``tau`` module to a kernel activation code: ``(name, slope, slope_tensor)``;
    ``("custom", ...)`` means the module is applied by torch between the kernels."""
    if type(tau) is nn.ReLU:
        return "relu", 0.0, None
    if type(tau) is nn.LeakyReLU and tau.negative_slope > 0:
        return "leakyrelu", float(tau.negative_slope), None
    if type(tau) is nn.PReLU and tau.weight.numel() == 1:
        return "prelu", 0.0, tau.weight
    if type(tau) is nn.Tanh:
        return "tanh", 0.0, None
    if type(tau) is nn.ELU and tau.alpha == 1.0:
        return "elu", 0.0, None
    return "custom", 0.0, None


class _HParams(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


_CACHE_KEYS = ("_dmpnn_replay", "_dmpnn_wcache", "_dmpnn_last", "_dmpnn_mon")
# what a GradSync hangs on a block (views into ITS flat buffer): never pickled / deep-copied along with the module
_STATE_SKIP_KEYS = _CACHE_KEYS + ("_dmpnn_grad_views", "_dmpnn_grad_written")


def invalidate(module: nn.Module) -> None:
    """Drop what the engine remembers about ``module`` between forwards (the replayed argument block, the scratch buffers, the
    spill monitor).  Nothing of it depends on the weights' VALUES — every forward pre-splits the weights afresh (round 4: in K0's
    launch, for free), so an update through ``param.data`` (EMA / SWA swaps) is seen like any other; the replayed argument block
    keys on the parameter tensors' identity, ``data_ptr()`` and device, which a re-allocation or a device move changes."""
    for m in module.modules():
        for k in _CACHE_KEYS:
            m.__dict__.pop(k, None)


class EngineStateMixin:
    """``copy.deepcopy`` / pickling / ``.to()`` of a block must not carry pointers into another module's buffers."""

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in _STATE_SKIP_KEYS}

    def _apply(self, fn, *args, **kwargs):
        for k in _CACHE_KEYS:
            self.__dict__.pop(k, None)
        return super()._apply(fn, *args, **kwargs)


class BondMessagePassing(EngineStateMixin, nn.Module):
    """Directed-bond message passing (D-MPNN encoder) on MI355X HIP kernels."""

    def __init__(self, d_v: int = DEFAULT_ATOM_FDIM, d_e: int = DEFAULT_BOND_FDIM,
                 d_h: int = DEFAULT_HIDDEN_DIM, bias: bool = False, depth: int = 3, dropout: float = 0.0,
                 activation="relu", undirected: bool = False, d_vd: Optional[int] = None,
                 V_d_transform: Optional[nn.Module] = None, graph_transform: Optional[nn.Module] = None):
        super().__init__()
        self.hparams = _HParams(d_v=d_v, d_e=d_e, d_h=d_h, bias=bias, depth=depth, dropout=dropout,
                                activation=activation, undirected=undirected, d_vd=d_vd,
                                V_d_transform=V_d_transform, graph_transform=graph_transform,
                                cls=self.__class__)
        # same construction order as base.py:238-251 -> same RNG stream -> same initial weights
        self.W_i = nn.Linear(d_v + d_e, d_h, bias)
        self.W_h = nn.Linear(d_h, d_h, bias)
        self.W_o = nn.Linear(d_v + d_h, d_h)
        self.W_d = nn.Linear(d_h + d_vd, d_h + d_vd) if d_vd else None
        self.depth = depth
        self.undirected = undirected
        self.dropout = nn.Dropout(dropout)
        self.tau = get_activation_function(activation)
        self.V_d_transform = V_d_transform if V_d_transform is not None else nn.Identity()
        self.graph_transform = graph_transform if graph_transform is not None else nn.Identity()

    @property
    def output_dim(self) -> int:
        return self.W_d.out_features if self.W_d is not None else self.W_o.out_features

    def forward(self, bmg, V_d: Optional[Tensor] = None) -> Tensor:
        return bond_message_passing_forward(self, bmg, V_d)


def _training_plan_kind(mp, bmg, atom: bool = False):
    """``"tiles"`` when a TRAINING forward of ``mp`` on ``bmg`` runs on the tile plan (K0 = the tile table alone, kept tensors in the
    caller's edge order, ``DMPNN_F_TILE_PLAN``), else ``False`` (the full CSR plan).  The SHAPE rule is the library's
    (``dmpnn_train_route``, include/dmpnn.h: directed, built-in activation, no ``W_d``, the shapes of the tile kernel, a plan the
    library can build from what the batch carries, at most 30 directed edges per molecule); the host adds only what it alone knows:
    the environment switches, an activation / dropout MODULE the kernels cannot stand in for, the validation window of the module's
    first batches, whether a gradient of ``W_i`` or ``W_h`` is wanted at all."""
    if _lib.opt("DMPNN_GENERAL", "0") == "1" or _lib.opt("DMPNN_TRAIN_PLAN", "tiles") == "full":
        return False
    act = classify_activation(mp.tau)[0]
    if act == "custom" or (mp.training and mp.dropout.p > 0 and type(mp.dropout) is not nn.Dropout):
        return False
    mode = _lib.opt("DMPNN_VALIDATE", "first")
    if mode == "always" or (mode != "never" and getattr(mp, "_dmpnn_batches_checked", 0) < _VALIDATE_FIRST_N):
        return False   # (the per-batch verdict is read from a full plan)
    if not (mp.W_i.weight.requires_grad or mp.W_h.weight.requires_grad):
        return False
    d_h, d_in = mp.W_h.weight.shape[0], mp.W_i.weight.shape[1]
    d_v = mp.W_o.weight.shape[1] - d_h
    d_e = int(bmg.E.shape[1]) if atom else d_in - d_v
    batch = getattr(bmg, "batch", None)
    info = engine.train_route(int(bmg.V.shape[0]), int(bmg.E.shape[0]), d_v, d_e, d_h, mp.depth, act,
                              len(bmg) if hasattr(bmg, "__len__") else 0, undirected=bool(mp.undirected), has_vd=mp.W_d is not None,
                              dropout_p=float(mp.dropout.p) if mp.training else 0.0, atom=atom,
                              have_batch=batch is not None and batch.dtype == torch.int64 and batch.is_contiguous(),
                              have_table=getattr(bmg, "tiles", None) is not None, oversize=getattr(bmg, "oversize", None),
                              max_level=1 if getattr(mp, "_dmpnn_no_mega", False) else 2)
    return "tiles" if info.plan_kind == 2 else False


_ENV_KEYS = ("DMPNN_GENERAL", "DMPNN_MEGA", "DMPNN_MFMA", "DMPNN_VALIDATE", "DMPNN_STORE")


class _Replay:
    """What a steady inference forward of the tile kernel repeats verbatim: the argument block of the last such forward
    (every pointer that does not belong to the batch), the objects those pointers came from, and the facts the route
    decision was taken on.  A call that finds all of them unchanged only builds the tile plan, allocates ``out`` and
    fills in the batch's pointers — the same two C calls, without the general routing code in between."""

    __slots__ = ("args", "tensors", "ptrs", "tau", "training", "env", "dev", "d_v", "d_e", "d_h", "wsplit", "depth", "block", "flags",
                 "ldh", "spill")


def _param(mp, lin: str, name: str):
    m = mp._modules.get(lin)
    return None if m is None else m._parameters.get(name)


def _make_replay(mp, plan, st) -> None:
    """After a slow-path forward that took the tile kernel on a tile plan: remember how to repeat it."""
    mp.__dict__.pop("_dmpnn_replay", None)
    if st is None or st.route not in ("mega16", "mega16/f16-operands") or not plan.tiles_only or mp.W_d is not None:
        return
    r = _Replay()
    r.args = bytes(st.args)  # template copy of the argument block
    r.tensors = tuple(_param(mp, l, n) for l, n in (("W_i", "weight"), ("W_h", "weight"), ("W_o", "weight"), ("W_i", "bias"),
                                                      ("W_h", "bias"), ("W_o", "bias")))
    r.ptrs = tuple(0 if t is None else t.data_ptr() for t in r.tensors)
    r.tau, r.training, r.depth = mp._modules.get("tau"), mp.training, mp.depth
    r.env = tuple(_lib.opt(k, "") for k in _ENV_KEYS)
    r.dev = plan.device
    r.d_v, r.d_e, r.d_h = st.dims["d_v"], st.dims["d_e"], st.dims["d_h"]
    r.wsplit = st.refs[-1]
    if r.wsplit is None or any(t is not None and (t.dtype != torch.float32 or not t.is_contiguous()) for t in r.tensors):
        return
    # the live argument block of the steady path: everything that does not belong to a batch is set here, once
    blk = r.block = _lib.FwdArgs.from_buffer_copy(r.args)
    blk.ldv, blk.lde, blk.ldout = r.d_v, r.d_e, r.d_h
    # (the pre-split of the weights is redone by every forward — it rides in K0's launch, dmpnn_forward_tiles — so the steady path
    #  needs no promise about the weights' VALUES: the checks below are about identity, shape and layout only)
    r.flags = int(blk.flags) & ~_lib.F_WSPLIT_READY & ~_lib.F_LOADER_TILES
    r.ldh, r.spill = int(blk.ldh), None
    mp.__dict__["_dmpnn_replay"] = r


def _replay_forward(mp, r: "_Replay", bmg):
    """The steady inference path (see :class:`_Replay`); ``None`` when anything it rests on has changed.

    Its host side is as long as its device side (two launches, ~45 us at 512 molecules), so it is kept lean: the argument block
    lives in the replay state and only the batch's fields are rewritten, K0 and the forward are ONE foreign call
    (``dmpnn_forward_tiles``), the scratch of the kernel's generic path is allocated once per module and grows, and a batch made
    by this package's own batching code (``data.BatchMolGraph``: float32 / int64 / contiguous by construction) is not
    re-inspected tensor by tensor."""
    mods = mp._modules
    if (mp.training != r.training or mods.get("tau") is not r.tau or mp.depth != r.depth or mp.undirected
            or type(mods.get("graph_transform")) is not nn.Identity or mods.get("W_d") is not None):
        return None
    ts = r.tensors
    if (_param(mp, "W_i", "weight") is not ts[0] or _param(mp, "W_h", "weight") is not ts[1] or _param(mp, "W_o", "weight") is not ts[2]
            or _param(mp, "W_i", "bias") is not ts[3] or _param(mp, "W_h", "bias") is not ts[4] or _param(mp, "W_o", "bias") is not ts[5]):
        return None
    for t, ptr in zip(ts, r.ptrs):  # same storage, same device (the VALUES may have changed: every forward splits the weights afresh)
        if t is not None and (t.data_ptr() != ptr or t.device != r.dev):
            return None
    if tuple(_lib.opt(k, "") for k in _ENV_KEYS) != r.env:
        return None
    V, E, ei, rev, batch = bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, getattr(bmg, "batch", None)
    dev = r.dev
    if V.device != dev or batch is None or V.dim() != 2 or E.dim() != 2 or V.shape[1] != r.d_v or E.shape[1] != r.d_e:
        return None
    # (`oversize` is set — True / False — only by this package's batching code, whose tensors are float32 / int64 / contiguous by
    #  construction; a batch wrapped around foreign tensors, `from_tensors`, carries None and is inspected)
    if (type(bmg) is not _OwnBatch or bmg.oversize is None) and (
            V.dtype != torch.float32 or E.dtype != torch.float32 or not V.is_contiguous() or not E.is_contiguous() or ei.dtype != torch.int64
            or rev.dtype != torch.int64 or not ei.is_contiguous() or not rev.is_contiguous() or ei.device != dev
            or batch.dtype != torch.int64 or not batch.is_contiguous() or batch.device != dev):
        return None
    nV, nE = V.shape[0], E.shape[0]
    n_mols = len(bmg) if hasattr(bmg, "__len__") else 0
    oversize = getattr(bmg, "oversize", None)
    if oversize is True:  # (the host knows a molecule of this batch exceeds the tile: per-step routes)
        return None
    tiles = getattr(bmg, "tiles", None)
    if tiles is not None and (tiles[0].device != dev or tiles[2] <= 0):
        tiles = None
    small = engine.small_plan_fits(nV, nE)
    lib = _lib.load()
    if (n_mols <= 0 or nE > 30 * n_mols or nE == 0 or (tiles is None and not small and not lib.dmpnn_tile_plan_any_size(nV, nE))
            or batch.numel() != nV
            or ei.shape[1] != nE or rev.numel() != nE or getattr(mp, "_dmpnn_no_mega", False)):
        return None
    nbytes = engine.plan_bytes(nV, nE)
    buf = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    out = torch.empty(nV, r.d_h, dtype=torch.float32, device=dev)
    a = r.block  # (the module's own argument block: consumed by the call below before it returns)
    pb = buf.data_ptr()
    a.plan, a.n_atoms, a.n_edges = pb, nV, nE
    a.V, a.E = V.data_ptr(), E.data_ptr()
    a.out = out.data_ptr()
    a.Mv = a.Hv = pb
    a.edge_index, a.rev_edge_index = ei.data_ptr(), rev.data_ptr()
    if oversize is None:  # bare tensors: scratch for the kernel's generic path, should a molecule exceed the tile (never touched else)
        need = (3 * nE + nV) * r.ldh
        sp = r.spill
        if sp is None or sp.numel() < need:
            sp = r.spill = torch.empty(need + need // 4, dtype=torch.float32, device=dev)
        a.spill_ws, a.spill_bytes = sp.data_ptr(), need * 4
    else:
        a.spill_ws, a.spill_bytes = None, 0
    cur = torch.cuda.current_device()
    ctx = None if dev.index is None or dev.index == cur else torch.cuda.device(dev)
    if ctx is not None:
        ctx.__enter__()
    try:
        stream = engine._raw_stream(dev.index if dev.index is not None else cur) if engine._raw_stream is not None else engine._stream_ptr(dev)
        if tiles is not None:  # the loader's table: K0 is a copy of it
            a.flags = r.flags | _lib.F_LOADER_TILES
            a.n_tiles_launch = tiles[2]
            rc = lib.dmpnn_forward_tiles(_ctypes.byref(a), None, tiles[0].data_ptr(), tiles[1].data_ptr(), tiles[2], nbytes, stream)
        else:
            a.flags = r.flags if small else (r.flags | _lib.F_LOADER_TILES)
            # (the tile count is on the device only; a tile holds at least one molecule, so the batch's molecule count bounds it — a third
            #  fewer idle workgroups than the layout's bound at 512 molecules, 1 us of the launch; a plan with more tiles than that comes back NaN)
            a.n_tiles_launch = n_mols
            rc = lib.dmpnn_forward_tiles(_ctypes.byref(a), batch.data_ptr(), None, None, 0, nbytes, stream)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    if rc:
        _lib.check(rc, "dmpnn_forward_tiles")
    if oversize is None:
        _spill_monitor(mp, buf, dev)
    from .agg import note_batch

    note_batch(batch, n_mols)
    return out


class _Monitor:
    __slots__ = ("host", "event", "pending", "calls", "observed", "with_spill")


def _spill_monitor(mp, plan_buf: Tensor, dev) -> None:
    """Speed heuristic only (correctness never depends on it: the tile kernels carry an oversize molecule through their
    generic path).  Now and then the header of a tile plan is copied to pinned host memory WITHOUT a sync and read at a
    later forward; a module that keeps meeting oversize molecules is moved to the per-step routes, whose time does not
    hinge on the slowest molecule."""
    if torch.cuda.is_current_stream_capturing():
        return  # (nothing of this — pinned allocation, event query, copy — belongs into a hipGraph capture)
    m = mp.__dict__.get("_dmpnn_mon")
    if m is None:
        m = _Monitor()
        m.host = torch.zeros(16, dtype=torch.int32).pin_memory()
        m.event = torch.cuda.Event()
        m.pending, m.calls, m.observed, m.with_spill = False, 0, 0, 0
        mp.__dict__["_dmpnn_mon"] = m
    if m.pending and m.event.query():
        m.pending = False
        _note_spills(mp, m, int(m.host[8]))
    m.calls += 1
    # (never while a hipGraph is being captured: the copy and the event would become nodes of the graph, and the event could not be queried)
    if not m.pending and (m.calls % 8 == 0 if m.calls <= 64 else m.calls % 64 == 0) and not torch.cuda.is_current_stream_capturing():
        m.host.copy_(plan_buf[:16], non_blocking=True)
        m.event.record(torch.cuda.current_stream(dev))
        m.pending = True


def _note_spills(mp, m, n_spill: int) -> None:
    m.observed += 1
    if n_spill > 0:
        m.with_spill += 1
    if m.with_spill >= 2 and 4 * m.with_spill >= m.observed:
        object.__setattr__(mp, "_dmpnn_no_mega", True)


def bond_message_passing_forward(mp, bmg, V_d: Optional[Tensor] = None) -> Tensor:
    """``_MessagePassingBase.forward`` (base.py:196-212) for any module with the reference's
    attributes (``W_i, W_h, W_o, W_d, depth, undirected, dropout, tau, graph_transform,
    V_d_transform``); shared by :class:`BondMessagePassing` and the chemprop subclass."""
    if torch.compiler.is_compiling() or torch.compiler.is_exporting():
        # torch.export / torch.compile: the whole block is one opaque operator with a shape-only fake (chemprop_amd/export.py)
        from .export import traced_forward

        return traced_forward(mp, bmg, V_d)
    if V_d is None and not torch.is_grad_enabled():
        r = mp.__dict__.get("_dmpnn_replay")
        if r is not None:
            out = _replay_forward(mp, r, bmg)
            if out is not None:
                return out
    bmg = mp.graph_transform(bmg)  # Identity, or eval-only scaling on a shallow copy (transforms.py:65-74)
    engine._require_device(bmg.V, "bmg.V")
    if mp.W_i.weight.device != bmg.V.device:
        raise RuntimeError(f"module is on {mp.W_i.weight.device} but the batch is on {bmg.V.device}")
    n_atoms = int(bmg.V.shape[0])
    if V_d is not None:
        V_d = mp.V_d_transform(V_d)
        d_vd = (mp.W_d.in_features - mp.W_o.out_features) if mp.W_d is not None else None
        if mp.W_d is None or V_d.dim() != 2 or V_d.shape[0] != n_atoms or V_d.shape[1] != d_vd:
            raise InvalidShapeError("V_d", V_d.shape, [n_atoms, d_vd if d_vd is not None else 0])
    n_mols = len(bmg) if hasattr(bmg, "__len__") else 0
    # a tile plan at any batch size: the table came with the batch (PackedBatch), or the batch vector is there for the
    # multi-workgroup planner
    loader_tiles = getattr(bmg, "tiles", None) is not None or (
        getattr(bmg, "batch", None) is not None and not engine.small_plan_fits(int(bmg.V.shape[0]), int(bmg.E.shape[0]))
        and bool(_lib.load().dmpnn_tile_plan_any_size(int(bmg.V.shape[0]), int(bmg.E.shape[0]))))
    oversize = getattr(bmg, "oversize", None)  # host knowledge of the batching code (None: bare tensors)
    if oversize is None and mp.training and mp.dropout.p > 0 and torch.is_grad_enabled():
        # (active dropout lives inside the tile kernels, whose generic path for a molecule beyond the tile has none: such a molecule
        #  would be NaN there — and the NaN loss would reach the optimizer.  Counted on the device for a foreign batch.)
        oversize = batch_oversize(bmg, n_mols)
    if oversize is True:
        loader_tiles = False
    # (an inference forward of the fused routes: tile kernel, per-step fused route on the f16 pipe, fp32 fused route)
    light = _light_plan_ok(mp)
    if light and oversize is not True and _tile_plan_ok(mp, int(bmg.V.shape[0]), int(bmg.E.shape[0]), n_mols, loader_tiles):
        light = "tiles"
    if not light and torch.is_grad_enabled() and V_d is None:
        # a training forward bound for the tile kernels: the tile plan (DMPNN_F_TILE_PLAN)
        light = _training_plan_kind(mp, bmg) if oversize is not True else False
    # (a tile plan: K0 is deferred into the forward's own call where the library can run it with the weight pre-split in ONE launch)
    plan = engine.GraphPlan.from_bmg(bmg, light=light, launch="defer" if light == "tiles" else True)
    if n_mols and getattr(bmg, "batch", None) is not None:
        from .agg import note_batch

        note_batch(bmg.batch, n_mols)  # the aggregation that follows (model.py:131) skips its host read of batch.max()
    mp.__dict__.pop("_dmpnn_last", None)
    plan.oversize = oversize
    level = _route(mp, plan, n_mols, getattr(bmg, "batch", None))
    if oversize is True:
        level = min(level, 1)
    out = mp_forward(mp, plan, bmg.V, bmg.E, V_d, max_level=level)
    last = mp.__dict__.pop("_dmpnn_last", None)  # (the forward's workspace must not outlive the call)
    mp.__dict__["_dmpnn_route"] = getattr(last, "route", None)  # diagnostics: the route the last slow-path forward took
    plan.ensure_launched()   # (nothing ran the deferred K0: a route that does not read the plan at all)
    if oversize is None and plan.tiles_only and last is not None and last.route in ("mega16", "mega16/f16-operands"):
        _spill_monitor(mp, plan.buf, plan.device)
    if light == "tiles" and V_d is None and not torch.is_grad_enabled() and _lib.opt("DMPNN_REPLAY", "1") != "0":
        _make_replay(mp, plan, last)
    return out


_VALIDATE_FIRST_N = 2


def batch_oversize(bmg, n_mols: int = 0):
    """``True`` / ``False``: a molecule of this batch exceeds the tile of the whole-forward kernels (> 32 atoms or > 48 directed
    edges).  Our own batching code knows (``bmg.oversize``, free while batching).  For a foreign batch — the reference's
    ``BatchMolGraph`` (``slots=True``: nothing can be attached to it, ``data/collate.py:13``) — it is COUNTED on the device from
    ``bmg.batch`` (two ``bincount`` + one host read): used only by the routes on which an oversize molecule is NaN rather than slow
    (atom messages, the mol-atom-bond blocks, active dropout on the tile kernels: the kernels' generic path implements none of
    them), where a NaN loss would otherwise reach the optimizer (round-4 ADVICE medium).  ``None``: no batch vector to count from."""
    o = getattr(bmg, "oversize", None)
    if o is not None:
        return bool(o)
    batch = getattr(bmg, "batch", None)
    if batch is None or batch.dtype != torch.int64 or batch.numel() != int(bmg.V.shape[0]):
        return None
    from .data import TILE_MAX_ATOMS, TILE_MAX_EDGES

    if batch.numel() == 0:
        return False
    n_mols = int(n_mols) if n_mols else 0
    na = torch.bincount(batch, minlength=n_mols)
    bad = na.max() > TILE_MAX_ATOMS
    if int(bmg.E.shape[0]) > 0:
        ne = torch.bincount(batch[bmg.edge_index[1]], minlength=n_mols)
        bad = bad | (ne.max() > TILE_MAX_EDGES)
    return bool(bad.item())


def _light_plan_ok(mp) -> bool:
    """Inference forward that is certain to take a fused route whatever the batch: the plan may skip the
    arrays only the general route / the backward pass read (``dmpnn_prepare_light``)."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in mp.parameters()):
        return False
    if mp.undirected or (mp.training and mp.dropout.p > 0) or classify_activation(mp.tau)[0] == "custom":
        return False
    if _lib.opt("DMPNN_GENERAL", "0") == "1":
        return False
    if _lib.opt("DMPNN_VALIDATE", "first") != "never" and getattr(mp, "_dmpnn_batches_checked", 0) < _VALIDATE_FIRST_N:
        return False  # the first batches may still be routed to the general kernels by the validation
    d_h, d_in = mp.W_h.weight.shape[0], mp.W_i.weight.shape[1]
    d_v = mp.W_o.weight.shape[1] - d_h
    return d_h % 4 == 0 and d_h <= 320 and d_v % 2 == 0 and (d_in - d_v) % 2 == 0


def _tile_plan_ok(mp, n_atoms: int, n_edges: int, n_mols: int, loader_tiles: bool = False) -> bool:
    """After the validated first batches, an inference forward that is going to take the whole-forward tile kernel
    on the f16 pipe needs only the piece-tile tables (``dmpnn_prepare_tiles``): the kernel reads the batch's own
    index arrays and checks every tile itself (a tile that is not closed returns NaN for its atoms)."""
    if _lib.opt("DMPNN_MEGA", "1") == "0" or _lib.opt("DMPNN_MFMA", "split16") == "f32":
        return False
    if getattr(mp, "_dmpnn_no_mega", False):
        return False
    if _lib.opt("DMPNN_VALIDATE", "first") == "always":
        return False  # (the per-batch verdict is read from a full plan)
    return n_mols > 0 and n_edges <= 30 * n_mols and (loader_tiles or engine.small_plan_fits(n_atoms, n_edges))


def _route(mp, plan, n_mols: int = 0, batch=None) -> int:
    """Cap on the route of this batch: 2 (whole-forward tile kernel), 1 (per-step fused) or 0 (general).

    The fused kernels assume a molecular graph (``rev`` an involution with ``src(rev e) == dst(e)``,
    in-degree <= 24) and the whole-forward tile kernel additionally molecules of <= 48 directed edges /
    <= 32 atoms; ``dmpnn_prepare`` checks all of that ON THE DEVICE and a violating batch makes the
    affected route return NaN (never a silently wrong number).  Reading the verdict needs a host sync,
    so by default (``DMPNN_VALIDATE=first``) only the first batches a module sees are checked
    synchronously — featurizer-produced graphs never violate the graph invariants; ``always`` checks
    every batch (a sync per forward), ``never`` trusts.  A batch found in violation runs the next more
    general route; one oversize molecule switches the tile kernel off for the module (datasets of
    larger molecules use the per-step fused route).

    Molecule SIZE is not a validity question any more: a molecule larger than a tile is a tile of its own that the
    tile kernels run through their generic fp32 path (``csrc/dmpnn_spill_impl.hpp``) — whenever it turns up, validated
    window or not.  The host only decides speed: a batch whose batching code knows it holds such a molecule
    (``bmg.oversize``), and a module that keeps meeting them (``_spill_monitor``), take the per-step routes."""
    mode = _lib.opt("DMPNN_VALIDATE", "first")
    seen = getattr(mp, "_dmpnn_batches_checked", 0)
    no_mega = getattr(mp, "_dmpnn_no_mega", False)
    if n_mols > 0 and plan.n_edges > 30 * n_mols:  # average molecule already near the tile: do not try
        no_mega = True
    if mode == "always" or (mode == "first" and seen < _VALIDATE_FIRST_N):
        object.__setattr__(mp, "_dmpnn_batches_checked", seen + 1)
        hdr = plan.header()
        flags = hdr[0]
        if flags & 7:
            return 0
        if hdr[8] > 0 and getattr(plan, "oversize", None) is None:
            # oversize molecules (the tile kernels carry them through their generic path: correct, SLOW — one workgroup, 1.7 ms
            # for a 40-atom molecule at d_h = 300, whatever the size of the batch around it: scripts/probe_spill_policy.py): a
            # module whose first batches already hold some takes the per-step routes (a speed decision; _spill_monitor keeps
            # watching)
            no_mega = True
            object.__setattr__(mp, "_dmpnn_no_mega", True)
        if flags & 8:
            # no piece tiles: a molecule larger than a tile switches the tile kernel off for the module — but a batch
            # beyond the single-workgroup plan has none either way, which says nothing about its molecules
            if engine.small_plan_fits(plan.n_atoms, plan.n_edges) or (getattr(plan, "any_size", False) and not plan.tiles_only):
                # (a full plan with tiles from the batch vector, dmpnn_prepare_with_batch, carries the verdict itself)
                object.__setattr__(mp, "_dmpnn_no_mega", True)
            elif batch is not None and not no_mega and _lib.load().dmpnn_tile_plan_any_size(plan.n_atoms, plan.n_edges):
                # (validated batches only: the multi-workgroup tile planner's verdict on the molecule sizes of this batch)
                tp = engine.GraphPlan(plan.edge_index, plan.rev_edge_index, plan.n_atoms, light="tiles", batch=batch)
                if tp.tiles_only and tp.flags() & 8:
                    object.__setattr__(mp, "_dmpnn_no_mega", True)
            return 1
    return 1 if no_mega else 2


class AtomMessagePassing(BondMessagePassing):
    """f2: ``chemprop.nn.AtomMessagePassing`` (``base.py:254-289``, ``mixins.py:21-30``) on the
    same kernels — messages pass along atoms:

        H0 = W_i V[src]                      (mixins.py:22-23)
        M  = (segsum_dst [H || E])[src]      (mixins.py:25-30: no reverse-edge subtraction)
        H  = tau(H0 + W_h M)                 (base.py:135-141; W_h is [d_h, d_h + d_e])

    then the same final aggregation / finalize as the bond variant.  Every contraction and segment
    reduction is a HIP kernel (``dmpnn_linear_fwd``, ``dmpnn_aggregate_fwd``, ``dmpnn_gather_rows``); ``tau`` and
    dropout run as torch modules between them (the per-step route of the bond block).  The bond-feature
    half of the message, ``(segsum_dst E)[src]``, does not change over the depth loop and is formed once.
    Same constructor arguments, parameter shapes, ``state_dict`` keys and RNG stream as the reference."""

    def __init__(self, d_v: int = DEFAULT_ATOM_FDIM, d_e: int = DEFAULT_BOND_FDIM,
                 d_h: int = DEFAULT_HIDDEN_DIM, bias: bool = False, depth: int = 3, dropout: float = 0.0,
                 activation="relu", undirected: bool = False, d_vd: Optional[int] = None,
                 V_d_transform: Optional[nn.Module] = None, graph_transform: Optional[nn.Module] = None):
        nn.Module.__init__(self)
        self.hparams = _HParams(d_v=d_v, d_e=d_e, d_h=d_h, bias=bias, depth=depth, dropout=dropout,
                                activation=activation, undirected=undirected, d_vd=d_vd,
                                V_d_transform=V_d_transform, graph_transform=graph_transform,
                                cls=self.__class__)
        # same construction order as base.py:278-289 -> same RNG stream -> same initial weights
        self.W_i = nn.Linear(d_v, d_h, bias)
        self.W_h = nn.Linear(d_e + d_h, d_h, bias)
        self.W_o = nn.Linear(d_v + d_h, d_h)
        self.W_d = nn.Linear(d_h + d_vd, d_h + d_vd) if d_vd else None
        self.depth = depth
        self.undirected = undirected
        self.dropout = nn.Dropout(dropout)
        self.tau = get_activation_function(activation)
        self.V_d_transform = V_d_transform if V_d_transform is not None else nn.Identity()
        self.graph_transform = graph_transform if graph_transform is not None else nn.Identity()

    def forward(self, bmg, V_d: Optional[Tensor] = None) -> Tensor:
        return atom_message_passing_forward(self, bmg, V_d)


def atom_message_passing_forward(mp, bmg, V_d: Optional[Tensor] = None) -> Tensor:
    """``_MessagePassingBase.forward`` (base.py:196-212) with the atom mixin (mixins.py:21-30)."""
    from .backward import aggregate_fn, gather_src_fn, linear_fn

    if V_d is None and not torch.is_grad_enabled():
        r = mp.__dict__.get("_dmpnn_replay")   # the steady inference path of the tile kernel (the bond block's: the argument block carries DMPNN_F_ATOM)
        if r is not None:
            out = _replay_forward(mp, r, bmg)
            if out is not None:
                return out
    bmg = mp.graph_transform(bmg)
    engine._require_device(bmg.V, "bmg.V")
    if mp.W_i.weight.device != bmg.V.device:
        raise RuntimeError(f"module is on {mp.W_i.weight.device} but the batch is on {bmg.V.device}")
    n_atoms = int(bmg.V.shape[0])
    has_vd = False
    if V_d is not None:
        V_d = mp.V_d_transform(V_d)
        d_vd = (mp.W_d.in_features - mp.W_o.out_features) if mp.W_d is not None else None
        if mp.W_d is None or V_d.dim() != 2 or V_d.shape[0] != n_atoms or V_d.shape[1] != d_vd:
            raise InvalidShapeError("V_d", V_d.shape, [n_atoms, d_vd if d_vd is not None else 0])
        has_vd = True
    n_mols = len(bmg) if hasattr(bmg, "__len__") else 0
    if n_mols and getattr(bmg, "batch", None) is not None:
        from .agg import note_batch

        note_batch(bmg.batch, n_mols)
    V, E = bmg.V, bmg.E
    # ---- round 3: the whole forward of a tile of molecules in ONE launch (DMPNN_F_ATOM of the tile kernel) — inference, built-in
    # activation, molecules that fit a tile, d_e <= 16.  Everything else: the per-step kernels chained below. ----
    act, slope, slope_t = classify_activation(mp.tau)
    grad = torch.is_grad_enabled() and any(p.requires_grad for p in mp.parameters())
    # (an oversize molecule is NaN on the atom variant of the tile kernels — their generic path knows bond messages only — so the
    #  host must KNOW: counted on the device for a foreign batch, batch_oversize)
    if (not grad and not has_vd and act != "custom" and not mp.undirected and not (mp.training and mp.dropout.p > 0)
            and 1 <= int(E.shape[1]) <= 16 and int(E.shape[0]) > 0 and batch_oversize(bmg, n_mols) is False):
        nV_, nE_ = int(V.shape[0]), int(E.shape[0])
        loader_tiles = getattr(bmg, "tiles", None) is not None or (
            getattr(bmg, "batch", None) is not None and not engine.small_plan_fits(nV_, nE_) and bool(_lib.load().dmpnn_tile_plan_any_size(nV_, nE_)))
        light = "tiles" if (_light_plan_ok(mp) and _tile_plan_ok(mp, nV_, nE_, n_mols, loader_tiles)) else False
        plan = engine.GraphPlan.from_bmg(bmg, light=light)
        plan.oversize = getattr(bmg, "oversize", None)
        if _route(mp, plan, n_mols, getattr(bmg, "batch", None)) >= 2:
            try:
                out, st = engine.forward(plan, V, E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, mp.W_i.bias, mp.W_h.bias,
                                         depth=mp.depth, act=act, slope=slope, slope_t=slope_t, atom=True,
                                         wcache=mp.__dict__.setdefault("_dmpnn_wcache", {}))
                mp.__dict__["_dmpnn_route"] = st.route + "/atom"
                if light == "tiles" and not torch.is_grad_enabled() and _lib.opt("DMPNN_REPLAY", "1") != "0":
                    _make_replay(mp, plan, st)
                return out
            except engine.RouteUnavailable:
                pass
        if light:
            plan = engine.GraphPlan.from_bmg(bmg)
    elif (grad and not has_vd and act not in ("custom", "prelu") and not mp.undirected and not (mp.training and mp.dropout.p > 0)
          and 2 <= int(E.shape[1]) <= 16 and int(E.shape[1]) % 2 == 0 and int(E.shape[0]) > 0 and batch_oversize(bmg, n_mols) is False
          and not V.requires_grad and not E.requires_grad):
        # ---- round 4: TRAINING on the tile kernels (DMPNN_F_ATOM | DMPNN_F_KEEP): one forward launch that keeps what the backward tile
        # kernel and the weight-gradient products read (sign bits or H^(t), M^(t), the bond-feature half of the messages), one autograd
        # node — the bond block's (backward.FusedMP).  Molecules beyond the tile, W_d, active dropout: the chain below. ----
        from .backward import FusedMP

        light = _training_plan_kind(mp, bmg)
        plan = engine.GraphPlan.from_bmg(bmg, light=light, launch="defer" if light == "tiles" else True)
        plan.oversize = getattr(bmg, "oversize", None)
        if _route(mp, plan, n_mols, getattr(bmg, "batch", None)) >= 2:
            try:
                out = FusedMP.apply(mp, plan, V, E, None, act, slope, (slope_t, 2, None, True), mp.W_i.weight, mp.W_i.bias, mp.W_h.weight,
                                    mp.W_h.bias, mp.W_o.weight, mp.W_o.bias, None, None)
                plan.ensure_launched()
                mp.__dict__["_dmpnn_route"] = "mega16/atom"
                return out
            except engine.RouteUnavailable:
                pass
        if light:
            plan = engine.GraphPlan.from_bmg(bmg)
        else:
            plan.ensure_launched()
    else:
        plan = engine.GraphPlan.from_bmg(bmg)
    mp.__dict__["_dmpnn_route"] = "rows/atom"
    tau, drop = mp.tau, mp.dropout
    nE = plan.n_edges
    H0 = linear_fn(V, mp.W_i.weight, mp.W_i.bias, gather=plan.src32, n_rows=nE)
    H = tau(H0)
    if mp.depth > 1 and nE:
        with torch.no_grad():
            ME = engine.gather_rows(engine.aggregate(plan, E), plan.src32)      # [E, d_e], constant over the loop
    rev = None
    for _ in range(1, mp.depth):
        if mp.undirected:
            if rev is None:
                rev = plan.rev64
            H = (H + H[rev]) / 2
        if nE:
            MH = gather_src_fn(plan, aggregate_fn(plan, H))
            H = drop(tau(linear_fn(MH, mp.W_h.weight, mp.W_h.bias, A2=ME, Cadd=H0)))
        else:
            H = drop(tau(H0))
    Mv = aggregate_fn(plan, H)
    Hv = drop(tau(linear_fn(V, mp.W_o.weight, mp.W_o.bias, A2=Mv)))
    if has_vd:
        Hv = drop(linear_fn(Hv, mp.W_d.weight, mp.W_d.bias, A2=V_d))
    return Hv
