"""Thin host side of the engine: torch tensors in, C-ABI calls (include/dmpnn.h) out.

torch is plumbing here — device memory (caching allocator), streams, autograd bookkeeping.  All
arithmetic on the path runs in the HIP kernels of ``chemprop_amd/csrc``.  There is no CPU or eager
fallback: tensors that are not on a HIP device raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import (ACT, F_FUSED, F_H0_RESIDUAL, F_KEEP, F_LOADER_TILES, F_MEGA, F_ROW_FINALIZE, F_SPLIT16, F_STORE16, F_TILE_PLAN, F_UNDIRECTED, F_WSPLIT_READY,
                   PLAN_NOFUSE_MASK, PLAN_NOMEGA_MASK, FwdArgs, GemmArgs)


# The measured crossovers of the route rule live in the library (dmpnn_forward_route, csrc/dmpnn_abi.hip): from 20 000 directed
# edges on the per-step routes run their contractions on the f16 pipe; from 2 048 on an inference forward that is not the tile
# kernel's takes the per-step fused route on the f16 pipe.  (Mirrors for the host-side plan choice of nn.py.)
STEPS16_MIN_EDGES = 20000
FUSED16_MIN_EDGES = 2048


def small_plan_fits(n_atoms: int, n_edges: int) -> bool:
    """Batches the single-workgroup plan takes (mirror of csrc/dmpnn_common.hpp small_plan_fits): only
    those get piece tiles, hence the whole-forward tile kernel, and a light plan."""
    lds = ((3 * (n_atoms + 2)) * 4 + (5 * n_edges + 2) * 2 + 31) & ~15
    return n_atoms <= 6144 and n_edges <= 10240 and lds <= 160 * 1024 - 2048


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_plan_bytes_cache: dict = {}


def plan_bytes(n_atoms: int, n_edges: int) -> int:
    """``dmpnn_plan_bytes`` (memoised: a pure function of the two sizes)."""
    k = (n_atoms, n_edges)
    v = _plan_bytes_cache.get(k)
    if v is None:
        if len(_plan_bytes_cache) > 4096:
            _plan_bytes_cache.clear()
        v = _plan_bytes_cache[k] = int(_lib.load().dmpnn_plan_bytes(n_atoms, n_edges))
    return v


def _stream_ptr(device) -> int:
    """Raw handle of torch's current stream on ``device`` (the launch stream of every kernel of this call)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


class _OnDevice:
    """``torch.cuda.device(dev)`` only when ``dev`` is not already the current device (the usual case costs nothing)."""

    __slots__ = ("ctx",)

    def __init__(self, dev):
        idx = dev.index
        self.ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _require_device(t: Tensor, name: str) -> None:
    if t.device.type != "cuda":
        raise RuntimeError(
            f"chemprop_amd: `{name}` lives on {t.device}; the MI355X engine runs HIP kernels only "
            "(no CPU fallback) — move the batch and the module to a GPU device first")


def _f32c(t: Tensor, name: str) -> Tensor:
    _require_device(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    if t.dim() == 2 and t.shape[0] <= 1:
        # an empty or one-row matrix counts as contiguous whatever its strides say (a [0, d_e] tensor out of a numpy
        # concatenation has strides (0, 0)): the C ABI wants a leading dimension >= the row length
        return torch.empty(t.shape, dtype=t.dtype, device=t.device).copy_(t)
    return t.contiguous()


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class GraphPlan:
    """K0: int32 indices + stable incoming-edge CSR of one batch, built on device (no host sync)."""

    __slots__ = ("buf", "n_atoms", "n_edges", "device", "light", "tiles_only", "edge_index", "rev_edge_index", "loader_tiles",
                 "any_size", "oversize", "pending", "_job")

    def __init__(self, edge_index: Tensor, rev_edge_index: Tensor, n_atoms: int, light=False, batch: Optional[Tensor] = None,
                 tiles: Optional[tuple] = None, launch: bool = True):
        _require_device(edge_index, "edge_index")
        lib = _lib.load()
        dev = edge_index.device
        n_edges = int(edge_index.shape[1])
        ei = edge_index if edge_index.dtype == torch.int64 else edge_index.long()
        ei = ei.contiguous()
        rev = rev_edge_index if rev_edge_index.dtype == torch.int64 else rev_edge_index.long()
        rev = rev.contiguous()
        nbytes = plan_bytes(n_atoms, n_edges)
        self.buf = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
        self.n_atoms, self.n_edges, self.device = int(n_atoms), n_edges, dev
        self.oversize = None  # host knowledge of the batching code: a molecule exceeds the tile kernel's tile (True / False / unknown)
        # light=True: only what a forward of the fused routes reads (inference); light="tiles": only the piece-tile
        # tables — the whole-forward tile kernel then works on the caller's own index arrays (kept alive here);
        # batches beyond the single-workgroup plan always get the full plan
        small = small_plan_fits(n_atoms, n_edges)
        # tiles = (tile_row, tile_atom, n_tiles): the loader's table (PackedBatch, dmpnn_pack_tiles) — with light="tiles" the
        # plan is a copy of it (dmpnn_prepare_tiles_from_table), whatever the batch size
        self.loader_tiles = 0
        if light == "tiles" and tiles is not None:
            tr, ta, nt = tiles
            if (tr.dtype == torch.int32 and ta.dtype == torch.int32 and tr.device == dev and ta.device == dev
                    and tr.is_contiguous() and ta.is_contiguous() and tr.numel() > nt and ta.numel() > nt and nt > 0):
                self.loader_tiles = int(nt)
        # with the batch vector (int64, like the reference's) the tiles are whole molecules found by binary search — in
        # one workgroup's LDS for small batches, by three multi-workgroup launches beyond (dmpnn_tile_plan_any_size)
        bt = batch if (batch is not None and batch.dtype == torch.int64 and batch.device == dev
                       and batch.numel() == n_atoms and batch.is_contiguous()) else None
        big = (light == "tiles" and not small and not self.loader_tiles and bt is not None and n_atoms > 0
               and bool(lib.dmpnn_tile_plan_any_size(n_atoms, n_edges)))
        self.tiles_only = light == "tiles" and (small or self.loader_tiles > 0 or big)
        self.light = bool(light) and (small or self.loader_tiles > 0 or big)
        # a FULL plan beyond the single-workgroup plan: with the batch vector it carries molecule tiles too
        # (dmpnn_prepare_with_batch) — training on the tile kernels at any batch size
        full_tiles = (not light and not small and bt is not None and n_atoms > 0 and n_edges > 0
                      and bool(lib.dmpnn_full_plan_keeps_tiles(n_atoms, n_edges)))
        self.any_size = (self.tiles_only and not small) or full_tiles  # (the forward's DMPNN_F_LOADER_TILES)
        self.edge_index, self.rev_edge_index = ei, rev
        self.pending, self._job = None, None
        if launch == "defer":
            # K0 NOT launched yet: a tile plan from the batch vector within the single-workgroup plan — the forward that follows
            # runs K0, the pre-split of its weights (in K0's launch) and the tile kernel as ONE foreign call (dmpnn_forward_tiles);
            # anything else that wants the plan first calls ensure_launched()
            if self.tiles_only and small and not self.loader_tiles and bt is not None:
                self.pending = bt
                self._job = (tiles, bt, full_tiles, nbytes)
                return
            launch = True
        if not launch:
            # the buffer and the facts only: dmpnn_train_step runs K0 itself (dmpnn_prepare_with_batch: a FULL plan — with
            # molecule tiles where full_tiles says so; dmpnn_prepare_tiles for a tile plan, DMPNN_F_TILE_PLAN)
            if light and not self.tiles_only:
                raise RuntimeError("GraphPlan(launch=False) is the plan of a training step: full, or the tile plan (light='tiles')")
            self.loader_tiles = 0  # (the C call plans from the batch vector: the launch bound of the batch size, not a loader's count)
            return
        self._job = (tiles, bt, full_tiles, nbytes)
        self._launch()

    def ensure_launched(self) -> None:
        """Run a deferred K0 now (``launch="defer"``: the forward would have run it inside its own call)."""
        if self.pending is not None:
            self.pending = None
            self._launch()

    def _launch(self) -> None:
        lib = _lib.load()
        tiles, bt, full_tiles, nbytes = self._job
        self._job = None
        dev, ei, rev, n_atoms, n_edges = self.device, self.edge_index, self.rev_edge_index, self.n_atoms, self.n_edges
        with _OnDevice(dev):
            if self.loader_tiles:
                _lib.check(lib.dmpnn_prepare_tiles_from_table(tiles[0].data_ptr(), tiles[1].data_ptr(), self.loader_tiles, n_atoms,
                                                              n_edges, self.buf.data_ptr(), nbytes, _stream_ptr(dev)),
                           "dmpnn_prepare_tiles_from_table")
            elif self.tiles_only:
                _lib.check(lib.dmpnn_prepare_tiles(ei.data_ptr(), rev.data_ptr(), bt.data_ptr() if bt is not None else None,
                                                   n_atoms, n_edges, self.buf.data_ptr(), nbytes, _stream_ptr(dev)),
                           "dmpnn_prepare_tiles")
            elif full_tiles:
                _lib.check(lib.dmpnn_prepare_with_batch(ei.data_ptr(), rev.data_ptr(), bt.data_ptr(), n_atoms, n_edges,
                                                        self.buf.data_ptr(), nbytes, _stream_ptr(dev)), "dmpnn_prepare_with_batch")
            else:
                prep = lib.dmpnn_prepare_light if self.light else lib.dmpnn_prepare
                _lib.check(prep(ei.data_ptr(), rev.data_ptr(), n_atoms, n_edges,
                                self.buf.data_ptr(), nbytes, _stream_ptr(dev)), "dmpnn_prepare")

    @classmethod
    def from_bmg(cls, bmg, light=False, use_batch: bool = True, launch=True) -> "GraphPlan":
        return cls(bmg.edge_index, bmg.rev_edge_index, int(bmg.V.shape[0]), light=light,
                   batch=getattr(bmg, "batch", None) if use_batch else None,
                   tiles=getattr(bmg, "tiles", None) if light == "tiles" else None, launch=launch)

    # ---- views for tests / diagnostics (these synchronise) ----
    def arrays(self) -> dict:
        self.ensure_launched()
        off = self._offsets()
        b = self.buf.cpu()
        E, V, T = self.n_edges, self.n_atoms, int(off[11])
        cut = lambda o, n: b[o:o + n]
        return dict(hdr=b[:16], src=cut(off[0], E), dst=cut(off[1], E), rev=cut(off[2], E),
                    row_ptr=cut(off[3], V + 1), perm=cut(off[4], E), inv=cut(off[5], E), srcp=cut(off[6], E),
                    dstp=cut(off[7], E), revp=cut(off[8], E), tile_row=cut(off[9], T + 2),
                    tile_atom=cut(off[10], T + 2), mtile_row=cut(off[12], int(off[14]) + 2),
                    mtile_atom=cut(off[13], int(off[14]) + 2))

    def header(self) -> list:
        """The 16 header words (synchronises): [0] flags, [6] piece tiles, [8] oversize pieces (DMPNN_HDR_NSPILL), ..."""
        self.ensure_launched()
        return self.buf[:16].tolist()

    def flags(self) -> int:
        """Plan flag word (synchronises): bit0 asymmetric, bit1 index out of range, bit2 in-degree > 24."""
        self.ensure_launched()
        return int(self.buf[0].item())

    def fusable(self) -> bool:
        """True when the fused (row-tiled) forward represents this graph exactly (synchronises)."""
        return (self.flags() & PLAN_NOFUSE_MASK) == 0

    def mega_ok(self) -> bool:
        """True when every molecule fits a piece tile (<= 48 edge rows, <= 32 atoms): the whole-forward
        tile kernel applies (synchronises)."""
        return (self.flags() & PLAN_NOMEGA_MASK) == 0

    def _offsets(self):
        off = (C.c_int64 * _lib.PLAN_NOFFSETS)()
        _lib.check(_lib.load().dmpnn_plan_layout(self.n_atoms, self.n_edges, off), "dmpnn_plan_layout")
        return off

    def _view(self, k: int) -> Tensor:
        self.ensure_launched()
        off = self._offsets()
        return self.buf[off[k]:off[k] + self.n_edges]

    @property
    def src32(self) -> Tensor:
        return self._view(0)

    @property
    def dst32(self) -> Tensor:
        return self._view(1)

    @property
    def rev32(self) -> Tensor:
        return self._view(2)

    @property
    def rev64(self) -> Tensor:
        return self._view(2).long()

    @property
    def inv32(self) -> Tensor:
        """edge id -> CSR row (``X_rows[inv]`` brings a fused-forward edge tensor back to the caller's edge order)."""
        return self._view(5)

    @property
    def perm32(self) -> Tensor:
        """CSR row -> edge id, int32 (``X_edges[perm]`` takes an edge tensor in the caller's order to the kept tensors' row order)."""
        return self._view(4)

    @property
    def perm64(self) -> Tensor:
        """CSR row -> edge id (row i of a fused-forward edge tensor is edge perm[i])."""
        return self._view(4).long()


def fuse16_shapes(a) -> bool:
    """``dmpnn_forward_can_fuse16`` on the SHAPES of ``a`` (for a training forward its workspace requirements — kept slots,
    ``msplit`` — are met by the allocation that follows the route decision, so they are left out of the question here)."""
    flags = a.flags
    a.flags = flags & ~_lib.F_KEEP
    try:
        return bool(_lib.load().dmpnn_forward_can_fuse16(C.byref(a)))
    finally:
        a.flags = flags


def _lean16_bits(lib, a) -> int:
    """``dmpnn_forward_keep_bits_bytes`` for the per-step fused route's LEAN training forward on the shapes / options of ``a`` (0: not
    taken — another activation class, W_d, d_h > 320, depth 1)"""
    flags = a.flags
    a.flags = (flags | F_FUSED | F_SPLIT16 | F_KEEP) & ~F_MEGA
    try:
        return int(lib.dmpnn_forward_keep_bits_bytes(C.byref(a)))
    finally:
        a.flags = flags


def lean_sign_bits(st: "ForwardState") -> Tensor:
    """``[depth, n_edges, d_h]`` bool: ``tau(z) > 0`` at H0 (site 0) and every H^(t), rows in the plan's CSR-row order — what a lean
    training forward of the per-step fused route keeps instead of the fp32 tensors (tests / diagnostics)."""
    if not st.route.startswith("fused16/lean"):
        raise RuntimeError("lean_sign_bits: not a lean forward of the per-step fused route")
    bits = st.refs[16]
    d_h, depth = st.dims["d_h"], int(st.args.depth)
    bn = (d_h + 63) // 64 * 64
    b = bits.view(depth, st.plan.n_edges, bn // 8)
    sh = torch.arange(8, device=b.device, dtype=torch.uint8)
    return ((b.unsqueeze(-1) >> sh) & 1).bool().reshape(depth, st.plan.n_edges, bn)[:, :, :d_h]


KEEP_ROWS_MIN = 4096   # (= DMPNN_KEEP_ROWS_MIN of include/dmpnn.h: the rule itself is the library's, dmpnn_train_route)


def train_route(n_atoms: int, n_edges: int, d_v: int, d_e: int, d_h: int, depth: int, act: str, n_mols: int = 0, *, undirected: bool = False,
                has_vd: bool = False, dropout_p: float = 0.0, atom: bool = False, have_batch: bool = False, have_table: bool = False,
                oversize=None, max_level: int = 2) -> "_lib.TrainRouteInfo":
    """``dmpnn_train_route`` (include/dmpnn.h): for a TRAINING forward of these shapes, which plan K0 builds (``plan_kind`` 0 full /
    2 tiles), which route the forward takes on it, and the form of the kept tensors (``keep_rows`` / ``keep_bits`` / ``lean``) — the ONE
    training-side rule, in the library beside ``dmpnn_forward_route``; the host only contributes what it alone knows (the caller's
    cap, the environment switches: ``DMPNN_MEGA`` / ``DMPNN_MFMA`` / ``DMPNN_KEEP_ROWS``)."""
    a = FwdArgs()
    a.n_atoms, a.n_edges, a.d_v, a.d_e, a.d_h, a.depth = int(n_atoms), int(n_edges), int(d_v), int(d_e), int(d_h), int(depth)
    a.ldh, a.ldv, a.lde = (int(d_h) + 3) // 4 * 4, int(d_v), int(d_e)
    a.flags = (F_UNDIRECTED if undirected else 0) | (_lib.F_ATOM if atom else 0)
    a.act = act_code(act)
    a.dropout_p = float(dropout_p)
    a.W_d = 16 if has_vd else None      # (only its presence is read)
    cap = min(int(max_level), 1) if _lib.opt("DMPNN_MEGA", "1") == "0" else int(max_level)
    kr = {"1": 1, "0": 0}.get(_lib.opt("DMPNN_KEEP_ROWS", "auto"), -1)   # "1" / "0": always / never (tests, A/B measurements)
    info = _lib.TrainRouteInfo()
    _lib.check(_lib.load().dmpnn_train_route(C.byref(a), int(n_mols), (1 if have_batch else 0) | (2 if have_table else 0),
                                             -1 if oversize is None else (1 if oversize else 0), cap,
                                             1 if _lib.opt("DMPNN_MFMA", "split16") == "f32" else 0, kr, C.byref(info)), "dmpnn_train_route")
    return info


def split_rows_to_float(rows: Tensor, n_cols: int) -> Tensor:
    """Split rows (``[..., n_rows, row_floats]`` float32 storage: chunks of ``[hi 32 halfs | lo 32 halfs]`` + a 16-byte tail with the row's
    power-of-two scale, csrc/dmpnn_step16_impl.hpp) back to fp32 ``[..., n_rows, n_cols]`` (tests / diagnostics)."""
    lead, rf = rows.shape[:-1], rows.shape[-1]
    h = rows.contiguous().view(torch.float16).reshape(*lead, 2 * rf)
    nc = (n_cols + 31) // 32
    ch = h[..., :nc * 64].reshape(*lead, nc, 2, 32).float()
    scale = rows[..., rf - 4:rf - 3]
    return ((ch[..., 0, :] + ch[..., 1, :]).reshape(*lead, nc * 32)[..., :n_cols]) / scale


def kept_messages(st: "ForwardState") -> Tensor:
    """The kept ``M^(t)`` of a training forward as fp32 ``[depth - 1, n_edges, d_h]`` whatever form the route keeps them in: fp32 rows
    (``st.Ms``) or split rows (the tile kernel with ``msplit``, the lean per-step route).  Tests / diagnostics."""
    if st.args.msplit and st.refs[15] is not None and not st.dims.get("atom"):
        return split_rows_to_float(st.refs[15], st.dims["d_h"])
    return st.Ms[:, :, :st.dims["d_h"]]


class RouteUnavailable(RuntimeError):
    """A demanded kernel feature does not exist on the route this batch takes (the caller has another way)."""


def act_code(name: str) -> int:
    return ACT[str(name).lower()]


# ------------------------------------------------------------------------------------------------
# row kernels (used by per-row parity tests, by the custom-activation / dropout path, and by bench)
# ------------------------------------------------------------------------------------------------
def storage_f16() -> bool:
    """``DMPNN_STORE=f16``: the per-step fused route keeps the message tensor between the depth steps as one f16 per element
    with a power-of-two row scale (2 bytes instead of the exact 4-byte hi + lo pair) — an opt-in storage mode in the spirit of
    BASELINE configs[1]'s "bf16" (11-bit significands instead of bf16's 8).  NOT fp32-class: ~1e-4 relative on the output
    (tests hold 2e-3); everything else — H0, weights, accumulation, the output, every other route — is unchanged."""
    return _lib.opt("DMPNN_STORE", "f32") == "f16"


def message(plan: GraphPlan, H: Tensor, act_on_load: str = "none", slope: float = 0.0,
            slope_t: Optional[Tensor] = None, undirected: bool = False, out: Optional[Tensor] = None) -> Tensor:
    H = _f32c(H, "H")
    M = out if out is not None else torch.empty_like(H)
    with _OnDevice(H.device):
        _lib.check(_lib.load().dmpnn_message_fwd(
            plan.buf.data_ptr(), plan.n_atoms, plan.n_edges, H.shape[1], H.data_ptr(), H.stride(0),
            M.data_ptr(), M.stride(0), act_code(act_on_load), float(slope), _ptr(slope_t),
            F_UNDIRECTED if undirected else 0, _stream_ptr(H.device)), "dmpnn_message_fwd")
    return M


def aggregate(plan: GraphPlan, H: Tensor, act_on_load: str = "none", slope: float = 0.0,
              slope_t: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    H = _f32c(H, "H")
    Mv = out if out is not None else torch.empty(plan.n_atoms, H.shape[1], dtype=torch.float32, device=H.device)
    with _OnDevice(H.device):
        _lib.check(_lib.load().dmpnn_aggregate_fwd(
            plan.buf.data_ptr(), plan.n_atoms, plan.n_edges, H.shape[1], H.data_ptr(), H.stride(0),
            Mv.data_ptr(), Mv.stride(0), act_code(act_on_load), float(slope), _ptr(slope_t),
            _stream_ptr(H.device)), "dmpnn_aggregate_fwd")
    return Mv


def gather_rows(X: Tensor, idx32: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """``out[i] = X[idx32[i]]`` (``dmpnn_gather_rows``; an index out of range gives a NaN row)."""
    X = _f32c(X, "X")
    _require_device(idx32, "idx")
    if idx32.dtype != torch.int32:
        raise RuntimeError("gather_rows: int32 indices (the plan's arrays)")
    n = int(idx32.shape[0])
    O = out if out is not None else torch.empty(n, X.shape[1], dtype=torch.float32, device=X.device)
    with _OnDevice(X.device):
        _lib.check(_lib.load().dmpnn_gather_rows(X.data_ptr(), X.stride(0), X.shape[0], idx32.data_ptr(), n, X.shape[1],
                                                 O.data_ptr(), O.stride(0) if n else X.shape[1], _stream_ptr(X.device)),
                   "dmpnn_gather_rows")
    return O


def linear(A1: Tensor, W: Tensor, bias: Optional[Tensor] = None, A2: Optional[Tensor] = None,
           gather1: Optional[Tensor] = None, n_rows: Optional[int] = None, Cadd: Optional[Tensor] = None,
           act: str = "none", slope: float = 0.0, slope_t: Optional[Tensor] = None,
           out: Optional[Tensor] = None, zpre: Optional[Tensor] = None, mfma: Optional[str] = None) -> Tensor:
    """``act([A1[gather] || A2] @ W.T + bias + Cadd)``: the fp32-MFMA kernel, or with ``mfma="split16"`` the
    f16-pipe kernel with the exact 3-term operand split (``dmpnn_linear16_fwd``; raises if the shapes /
    alignments are not taken)."""
    A1 = _f32c(A1, "A1")
    W = _f32c(W, "W")
    if A2 is not None:
        A2 = _f32c(A2, "A2")
    M = int(n_rows) if n_rows is not None else (int(gather1.shape[0]) if gather1 is not None else int(A1.shape[0]))
    N = int(W.shape[0])
    K1, K2 = int(A1.shape[1]), (int(A2.shape[1]) if A2 is not None else 0)
    if W.shape[1] != K1 + K2:
        raise RuntimeError(f"linear: W has in_features {W.shape[1]} but operands give {K1}+{K2}")
    C_ = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=A1.device)
    g = GemmArgs()
    g.M, g.N, g.K1, g.K2 = M, N, K1, K2
    g.A1, g.lda1, g.gather1 = A1.data_ptr(), A1.stride(0), _ptr(gather1)
    g.gather1_rows = int(A1.shape[0])
    g.A2, g.lda2 = _ptr(A2), (A2.stride(0) if A2 is not None else 0)
    g.W, g.ldw = W.data_ptr(), W.stride(0)
    g.bias = _ptr(bias)
    g.Cadd, g.ldcadd = _ptr(Cadd), (Cadd.stride(0) if Cadd is not None else 0)
    g.C, g.ldc = C_.data_ptr(), C_.stride(0)
    g.Zpre, g.ldz = _ptr(zpre), (zpre.stride(0) if zpre is not None else 0)
    g.act, g.act_slope, g.act_slope_ptr = act_code(act), float(slope), _ptr(slope_t)
    lib = _lib.load()
    with _OnDevice(A1.device):
        if mfma == "split16":
            if not lib.dmpnn_linear16_ok(C.byref(g)):
                raise RuntimeError("linear(split16): shapes / alignments not taken by the split kernel")
            nb = int(lib.dmpnn_linear16_wsplit_bytes(N, K1 + K2))
            ws = torch.empty(nb, dtype=torch.uint8, device=A1.device)
            _lib.check(lib.dmpnn_linear16_fwd(C.byref(g), ws.data_ptr(), nb, 0, _stream_ptr(A1.device)), "dmpnn_linear16_fwd")
        else:
            _lib.check(lib.dmpnn_linear_fwd(C.byref(g), _stream_ptr(A1.device)), "dmpnn_linear_fwd")
    return C_


def update_fused(plan: GraphPlan, M: Tensor, H0: Tensor, W_h: Tensor, b_h: Optional[Tensor] = None,
                 act: str = "relu", slope: float = 0.0, slope_t: Optional[Tensor] = None, want_H: bool = False,
                 want_M: bool = True, want_Mv: bool = False, H_out: Optional[Tensor] = None,
                 M_next: Optional[Tensor] = None, Mv: Optional[Tensor] = None):
    """One fused depth step on CSR-row-ordered tensors: ``H' = tau(H0 + M W_h^T + b)`` and, from the
    same kernel's epilogue, the next message ``M_next`` and / or the per-atom aggregate ``Mv``."""
    M, H0, W_h = _f32c(M, "M"), _f32c(H0, "H0"), _f32c(W_h, "W_h").contiguous()
    dev, (nE, h) = M.device, M.shape
    if want_H and H_out is None:
        H_out = torch.empty_like(M)
    if want_M and M_next is None:
        M_next = torch.empty_like(M)
    if want_Mv and Mv is None:
        Mv = torch.empty(plan.n_atoms, h, dtype=torch.float32, device=dev)
    ld = lambda t: t.stride(0) if t is not None else 0
    with _OnDevice(dev):
        _lib.check(_lib.load().dmpnn_update_fwd(
            plan.buf.data_ptr(), plan.n_atoms, plan.n_edges, h, M.data_ptr(), M.stride(0), H0.data_ptr(), H0.stride(0),
            W_h.data_ptr(), _ptr(b_h), _ptr(H_out), ld(H_out), _ptr(M_next), ld(M_next), _ptr(Mv), ld(Mv),
            act_code(act), float(slope), _ptr(slope_t), _stream_ptr(dev)), "dmpnn_update_fwd")
    return H_out, M_next, Mv


# ------------------------------------------------------------------------------------------------
# whole forward (builtin activation, dropout inactive): one C call, kernels chained on the stream
# ------------------------------------------------------------------------------------------------
class ForwardState:
    """Workspace of one forward; kept alive for the backward pass when ``keep`` is set."""

    __slots__ = ("plan", "H0", "Hs", "Ms", "Mv", "Hv", "ldh", "n_hslots", "n_mslots", "out", "args", "refs", "dims",
                 "fused", "route")


def forward(plan: GraphPlan, V: Tensor, E: Tensor, W_i: Tensor, W_h: Tensor, W_o: Tensor, b_o: Tensor,
            b_i: Optional[Tensor] = None, b_h: Optional[Tensor] = None, W_d: Optional[Tensor] = None,
            b_d: Optional[Tensor] = None, V_d: Optional[Tensor] = None, depth: int = 3, act: str = "relu",
            slope: float = 0.0, slope_t: Optional[Tensor] = None, undirected: bool = False,
            keep: bool = False, fused: Optional[bool] = None, route: Optional[str] = None,
            max_level: int = 2, mfma: Optional[str] = None, wcache: Optional[dict] = None,
            launch: bool = True, form: int = 0, dropout: Optional[tuple] = None, atom: bool = False,
            keep_bits: bool = True) -> tuple[Tensor, ForwardState]:
    """One ``dmpnn_forward`` call.  Routes (``route`` = ``"mega" | "fused" | "general"``, default: the best
    the shapes allow):

    * ``mega``    — the whole forward of every tile of whole molecules in ONE launch (small batches,
                    molecules of <= 24 bonds);
    * ``fused``   — per depth step one contraction whose epilogue forms the segment sums (CSR-row order);
    * ``general`` — arbitrary index arrays / undirected / any ``d_h`` (caller's edge order).

    ``wcache``: a dict owned by the caller (one per module) that keeps the SCRATCH BUFFER of the weights' pre-split between calls
    (no allocation per forward); its contents are rewritten by every forward — nothing about the weights is cached.

    ``mfma`` picks the matrix arithmetic of the mega route: ``"split16"`` (default; fp32-equivalent exact
    3-term f16 split on the f16 matrix pipe) or ``"f32"`` (the exact fp32 MFMA); env ``DMPNN_MFMA``.
    ``form``: ``DMPNN_F_H0_RESIDUAL`` / ``DMPNN_F_ROW_FINALIZE`` bits for the per-step fused route on the f16 pipe (its other
    form of the residual / of the finalize, include/dmpnn.h; what training and wide hidden layers use anyway).
    ``atom=True``: ``AtomMessagePassing`` semantics (``DMPNN_F_ATOM``: ``W_i [d_h, d_v]``, ``W_h [d_h, d_h + d_e]``) — the tile
    kernel, inference or (round 4) training; raises :class:`RouteUnavailable` when this batch takes another route.
    ``dropout = (p, seed)``: ACTIVE dropout inside the kernels (``dmpnn_fwd_args.dropout_p``) — a training forward (``keep``) of
    the tile kernel with a ReLU-class activation and no ``W_d``; raises :class:`RouteUnavailable` when this batch takes another
    route (the caller then runs its own ``nn.Dropout`` between the row kernels).
    ``launch=False`` prepares the argument block and the workspace without enqueuing anything (``trainer.FusedTrainer``).
    ``route`` is a demand (raises when the shapes do not allow it); ``max_level`` (0 general, 1 fused,
    2 mega) only caps the automatic choice.  ``fused=False`` is shorthand for ``route="general"``;
    ``fused=True`` demands at least ``fused``.
    Graph properties are decided on the device by the plan: a graph a route cannot represent makes that
    route return NaN (see ``GraphPlan.fusable`` / ``GraphPlan.mega_ok``)."""
    lib = _lib.load()
    V = _f32c(V, "V")
    E = _f32c(E, "E")
    dev = V.device
    nV, nE = plan.n_atoms, plan.n_edges
    if V.shape[0] != nV or E.shape[0] != nE:
        raise RuntimeError(f"forward: plan is for V={nV}, E={nE} but got V={V.shape[0]}, E={E.shape[0]}")
    d_v, d_e, d_h = int(V.shape[1]), int(E.shape[1]), int(W_h.shape[0])
    if atom and (int(W_i.shape[1]) != d_v or int(W_h.shape[1]) != d_h + d_e):
        raise RuntimeError("forward(atom): W_i must be [d_h, d_v] and W_h [d_h, d_h + d_e] (base.py:278-289)")
    d_vd = int(V_d.shape[1]) if (W_d is not None and V_d is not None) else 0
    ldh = (d_h + 3) // 4 * 4
    n_steps = max(depth - 1, 0)

    a = FwdArgs()
    a.plan, a.n_atoms, a.n_edges = plan.buf.data_ptr(), nV, nE
    a.edge_index, a.rev_edge_index = plan.edge_index.data_ptr(), plan.rev_edge_index.data_ptr()
    a.d_v, a.d_e, a.d_h, a.d_vd = d_v, d_e, d_h, d_vd
    a.depth, a.flags = int(depth), (F_UNDIRECTED if undirected else 0) | (int(form) & (F_H0_RESIDUAL | F_ROW_FINALIZE))
    if getattr(plan, "loader_tiles", 0) or getattr(plan, "any_size", False):
        a.flags |= F_LOADER_TILES              # a tile plan of any batch size
        a.n_tiles_launch = plan.loader_tiles   # (0: the launch bound — the tile count is on the device only)
    a.act, a.act_slope, a.act_slope_ptr = act_code(act), float(slope), _ptr(slope_t)
    a.V, a.ldv = V.data_ptr(), V.stride(0)
    a.E, a.lde = E.data_ptr(), E.stride(0)
    if d_vd:
        V_d = _f32c(V_d, "V_d")
        a.V_d, a.ldvd = V_d.data_ptr(), V_d.stride(0)
    Wc = lambda t, n: None if t is None else _f32c(t, n).contiguous()
    W_i, W_h, W_o, b_o, b_i, b_h = Wc(W_i, "W_i"), Wc(W_h, "W_h"), Wc(W_o, "W_o"), Wc(b_o, "b_o"), Wc(b_i, "b_i"), Wc(b_h, "b_h")
    a.W_i, a.b_i, a.W_h, a.b_h, a.W_o, a.b_o = _ptr(W_i), _ptr(b_i), _ptr(W_h), _ptr(b_h), _ptr(W_o), _ptr(b_o)
    if d_vd:
        W_d, b_d = Wc(W_d, "W_d"), Wc(b_d, "b_d")
        a.W_d, a.b_d = _ptr(W_d), _ptr(b_d)
    a.ldh = ldh
    out = torch.empty(nV, d_h + d_vd, dtype=torch.float32, device=dev)
    a.out, a.ldout = out.data_ptr(), out.stride(0)

    # ---- route.  The DEFAULT policy is ONE shape rule in the library (dmpnn_forward_route: the measured crossovers are its
    # constants, tests/test_host.py enumerates it); `route` / `fused` / `mfma` are demands of tests and A/B measurements ----
    if fused is False:
        route = "general"
    if _lib.opt("DMPNN_GENERAL", "0") == "1":
        route = "general"
    a.H0 = a.Ms = a.Mv = plan.buf.data_ptr()  # any 16-byte aligned pointer: the real workspace is allocated below
    mf = mfma or _lib.opt("DMPNN_MFMA", "split16")
    light, tiles_only = bool(getattr(plan, "light", False)), bool(getattr(plan, "tiles_only", False))
    if route is None and fused is None and mfma is None:
        cap = min(int(max_level), 1) if _lib.opt("DMPNN_MEGA", "1") == "0" else int(max_level)
        rc = int(lib.dmpnn_forward_route(C.byref(a), 1 if keep else 0, cap, 2 if tiles_only else (1 if light else 0), 1 if mf == "f32" else 0))
        if rc < 0:
            raise RuntimeError("forward: " + ("a tile plan (light='tiles') only serves the whole-forward tile kernel on the f16 pipe" if tiles_only else
                                              "a light GraphPlan only serves inference forwards of the fused routes (build the plan with "
                                              "light=False for the general route or for training)"))
        name = _lib.ROUTES[rc]
        use_mega, use_fused16 = name.startswith("mega"), name == "fused16"
        use_fused = use_mega or use_fused16 or name == "fused"
        want16 = name.endswith("16")
    else:
        level = 0 if (route == "general" or undirected) else int(lib.dmpnn_forward_can_fuse(C.byref(a)))
        if route == "fused" or _lib.opt("DMPNN_MEGA", "1") == "0":
            level = min(level, 1)
        if route is None:  # (`fused=` / `mfma=` alone demand an arithmetic, not a route: the caller's cap still holds)
            level = min(level, int(max_level))
        if (fused is True or route in ("fused", "mega")) and level < (2 if route == "mega" else 1):
            raise RuntimeError(f"forward: route {route or 'fused'!r} requested but the shapes do not allow it "
                               "(fused: d_h % 4 == 0, d_h <= 320, even d_v / d_e, directed; mega: additionally a batch within the "
                               "single-workgroup plan, or a plan with molecule tiles)")
        # the per-step FUSED route on the f16 pipe on demand — also for a TRAINING forward (keep): k_step16 then writes H^(t) and an
        # fp32 copy of every message for the backward pass.  Built and measured in round 3 (40-atom x 4 096: step 6 989 us against
        # 6 949 us with the general forward): those stores eat what the fused forward saves, so the default rule does not take it
        use_fused16 = (route == "fused16" and not undirected and not tiles_only and mf == "split16" and not (keep and light) and fuse16_shapes(a))
        if route == "fused16" and not use_fused16:
            raise RuntimeError("forward: route 'fused16' requested but not available (directed, d_h % 4 == 0, d_h <= 640, "
                               "even d_v / d_e, a full or light plan — training: a full plan)")
        if use_fused16:
            level = 1
        use_fused, use_mega = level >= 1, level >= 2
        if light and (not use_fused or (keep and not (tiles_only and use_mega and not d_vd))):
            raise RuntimeError("forward: a light GraphPlan only serves inference forwards of the fused routes "
                               "(build the plan with light=False for the general route or for training; the tile plan also "
                               "serves a training forward of the tile kernel without W_d)")
        if tiles_only and (not use_mega or mf == "f32"):
            raise RuntimeError("forward: a tile plan (light='tiles') only serves the whole-forward tile kernel on the f16 pipe")
        # the per-step routes' contractions on the f16 pipe: mfma="split16" forces them, "f32" forbids them, else the rule's crossover
        want16 = use_mega or use_fused16 or (not use_fused and (mfma == "split16" or (mfma is None and mf == "split16" and (d_h > 320 or nE >= STEPS16_MIN_EDGES))))
        if mf == "f32":
            want16 = False

    if atom:
        if not (use_mega and want16 and not d_vd and 1 <= d_e <= 16 and (not keep or (d_e % 2 == 0 and d_v % 2 == 0 and d_h % 2 == 0))):
            raise RouteUnavailable("atom messages inside the kernels: the tile kernel, 1 <= d_e <= 16, no W_d (training: even d_v / d_e / d_h)")
        a.flags |= _lib.F_ATOM
    if dropout is not None and float(dropout[0]) > 0.0:
        if not (use_mega and want16 and keep and not d_vd and act in ("relu", "leakyrelu")):
            raise RouteUnavailable("dropout inside the kernels: training forward of the tile kernel, ReLU-class activation, no W_d")
        a.dropout_p, a.dropout_seed = float(dropout[0]), int(dropout[1]) & 0xFFFFFFFFFFFFFFFF
    bits = None
    lean16 = False
    st = ForwardState()
    st.fused = use_fused
    st.route = "mega" if use_mega else ("fused16" if use_fused16 else ("fused" if use_fused else "general"))
    if use_mega:
        n_hslots = n_steps if keep else 0           # inference: H / M never leave the CU
        n_mslots = n_steps if keep else 0
    elif use_fused:
        n_hslots = n_steps if keep else 0           # H^(t) only matters to the backward pass
        n_mslots = n_steps if keep else min(n_steps, 2)
    else:
        n_hslots = max(n_steps, 1) if keep else 1
        n_mslots = max(n_steps, 1) if keep else 1
    st.plan, st.ldh, st.n_hslots, st.n_mslots = plan, ldh, n_hslots, n_mslots
    need_h0 = not (use_mega and not keep)
    spill_ws = None
    if use_mega and not keep and getattr(plan, "oversize", None) is not False and nV:
        # scratch of the tile kernel's generic path, should a molecule exceed its tile (never touched otherwise)
        spill_ws = torch.empty((3 * nE + nV) * ldh, dtype=torch.float32, device=dev)
        a.spill_ws, a.spill_bytes = spill_ws.data_ptr(), spill_ws.numel() * 4
    split_ms = None
    if use_mega and not keep and not d_vd:  # inference tile kernel: nothing leaves the CU but `out`
        edge_ws = atom_ws = None
    elif use_fused16 and keep and keep_bits and _lean16_bits(lib, a) > 0:
        # LEAN training forward (round 4; ReLU-class activation, no W_d, d_h <= 320): nothing is kept that the backward step kernels
        # (csrc/dmpnn_bstep16.hip) do not read — the split message rows of every step (depth - 1 slots), the split K1 operand (in
        # the H0 buffer: no H0 tensor, the residual is recomputed per step as in inference), one sign bit per element of H0 / H^(t)
        srf = int(lib.dmpnn_split_row_floats(d_h))
        lean16 = True
        n_hslots = n_mslots = 0
        # (the H0 buffer holds the split K1 operand, rows of ceil((d_v + d_e) / 32) * 128 + 16 bytes: include/dmpnn.h, keep_bits)
        xrow = ((d_v + d_e + 31) // 32 * 128 + 16) // 4
        edge_ws = torch.empty((1, nE, max(ldh, xrow)), dtype=torch.float32, device=dev)
        split_ms = torch.empty((max(n_steps, 1), nE, srf), dtype=torch.float32, device=dev)
        atom_ws = torch.empty((2, nV, ldh), dtype=torch.float32, device=dev)
        bits = torch.empty(_lean16_bits(lib, a), dtype=torch.uint8, device=dev)
    elif use_fused16 and keep:
        # training: H0 | H^(t) | M^(t) fp32 (what dmpnn_backward reads, CSR-row order) + the two split ping-pong slots in `msplit`
        srf = int(lib.dmpnn_split_row_floats(d_h))
        n_hslots = n_mslots = n_steps
        edge_ws = torch.empty((1 + 2 * n_steps, nE, ldh), dtype=torch.float32, device=dev)
        split_ms = torch.empty((2, nE, srf), dtype=torch.float32, device=dev)
        atom_ws = torch.empty((2, nV, ldh), dtype=torch.float32, device=dev)
    elif use_fused16:
        # H0 fp32 | two slots of split message rows (row = chunks of [hi | lo] halfs + a 16-byte tail with the row's scale)
        srf = int(lib.dmpnn_split_row_floats(d_h))
        # round 5: with a buffer of dmpnn_forward_h0_bytes() the route keeps H0 as ROW QUADS — the step kernel's accumulator-fragment
        # layout, written by K1 from its registers, read back by every depth step with coalesced 16-byte loads — instead of recomputing
        # W_i x in every step (d_h <= 320) or gathering fp32 rows word by word (d_h > 320): include/dmpnn.h, dmpnn_fwd_args.h0_bytes.
        # DMPNN_H0=x keeps the forms of ABI <= 11 (A/B measurements)
        h0_rows = nE
        if _lib.opt("DMPNN_H0", "quads") != "x":
            flags = a.flags
            a.flags = (flags | F_FUSED | F_SPLIT16) & ~(F_MEGA | F_KEEP)   # (form bits stay: DMPNN_F_H0_RESIDUAL asks for fp32 rows)
            try:
                need = int(lib.dmpnn_forward_h0_bytes(C.byref(a)))
            finally:
                a.flags = flags
            if need:
                h0_rows = max(nE, (need // 4 + ldh - 1) // ldh)
                a.h0_bytes = h0_rows * ldh * 4
        edge_ws = torch.empty((1, h0_rows, ldh), dtype=torch.float32, device=dev)
        split_ms = torch.empty((2, nE, srf), dtype=torch.float32, device=dev)
        atom_ws = torch.empty((2, nV, ldh), dtype=torch.float32, device=dev)
        n_hslots, n_mslots = 0, 2
    else:
        edge_ws = torch.empty(((1 if need_h0 else 0) + n_hslots + n_mslots, nE, ldh), dtype=torch.float32, device=dev)
        atom_ws = torch.empty((2, nV, ldh), dtype=torch.float32, device=dev)
        if atom and keep:
            # the bond-feature half of the atom messages, kept for W_h's gradient: depth - 1 slots of [n_edges][16] (include/dmpnn.h, DMPNN_F_ATOM)
            split_ms = torch.empty((max(n_steps, 1), nE, 16), dtype=torch.float32, device=dev)
        elif use_mega and want16 and keep and n_steps and train_route(nV, nE, d_v, d_e, d_h, depth, act, 1, max_level=2).keep_rows:
            # round 4: the tile kernel keeps M^(t) as SPLIT ROWS (depth - 1 slots of n_edges rows of dmpnn_split_row_floats(d_h) floats) —
            # what the weight-gradient products read as they are (csrc/dmpnn_wgrad16.hip: k_wgrad16r); the fp32 Ms slots above then only
            # serve a molecule beyond the tile.  From KEEP_ROWS_MIN message rows on — 4 096 since round 6 (k_wgrad16r rebuilt for one
            # workgroup per CU: 49 -> 31.7 us per launch at 512 QM9-shaped molecules); whole-model step, split rows | blocks, same box
            # (profiles/r06_rows_crossover.txt): 128 molecules 145 | 149 us, 256 156 | 162, 512 177 | 195, 768 253 | 300, 1 024 281 | 364;
            # 64 molecules (2 184 message rows) 140 | 137: the rule's lower end
            split_ms = torch.empty((n_steps, nE, int(lib.dmpnn_split_row_floats(d_h))), dtype=torch.float32, device=dev)
    st.out = out
    if edge_ws is None:
        st.H0 = st.Hs = st.Ms = st.Mv = st.Hv = None
        a.H0 = a.Hs = a.Ms = None
        a.n_hslots, a.n_mslots = 0, 1
        a.Mv = a.Hv = plan.buf.data_ptr()  # (never written: the tile kernel stores kept tensors with DMPNN_F_KEEP only)
    else:
        if ldh != d_h:  # pad columns are read by vector loads of later kernels: keep them finite
            edge_ws.zero_()
            atom_ws.zero_()
        i0 = 1 if need_h0 else 0
        st.H0, st.Hs, st.Ms = (edge_ws[0] if need_h0 else None), edge_ws[i0:i0 + n_hslots], edge_ws[i0 + n_hslots:]
        st.Mv, st.Hv = atom_ws[0], atom_ws[1]
        a.H0 = st.H0.data_ptr() if need_h0 else None
        a.Hs, a.n_hslots = (st.Hs.data_ptr() if n_hslots else None), n_hslots
        a.Ms, a.n_mslots = (st.Ms.data_ptr() if n_mslots else None), max(n_mslots, 1)
        a.Mv, a.Hv = st.Mv.data_ptr(), st.Hv.data_ptr()
        if lean16:
            a.msplit, a.msplit_bytes = split_ms.data_ptr(), split_ms.numel() * 4
            a.keep_bits, a.keep_bits_bytes = bits.data_ptr(), bits.numel()
            a.Hs = a.Ms = None
            st.Ms = split_ms      # (the kept M^(t) as split rows, CSR-row order)
        elif split_ms is not None and keep:
            a.msplit, a.msplit_bytes = split_ms.data_ptr(), split_ms.numel() * 4
        elif split_ms is not None:
            st.Ms = split_ms
            a.Ms, a.n_mslots = split_ms.data_ptr(), 2
    if use_fused:
        a.flags |= F_FUSED
    wsplit = None
    if want16:
        if use_mega:
            a.flags |= F_MEGA
        a.flags |= F_SPLIT16
        if keep:
            a.flags |= F_KEEP  # (the size below then includes the backward tile kernel's two transposed matrices)
        nb = int(lib.dmpnn_forward_wsplit_bytes(C.byref(a)))
        # scratch of the weights' pre-split — redone by EVERY forward (round 4: it rides in K0's launch on the steady path, so keeping
        # it between forwards on the strength of the tensors' autograd versions — which a write through `param.data` does not bump —
        # bought nothing and could go stale silently); `wcache`, when given, only keeps the BUFFER of a module between its calls
        # (a TRAINING forward gets a buffer of its own: the backward pass reads the transposed matrices in it later)
        wsplit = wcache.get("buf") if (wcache is not None and not keep) else None
        if wsplit is None or wsplit.numel() != nb or wsplit.device != dev:
            wsplit = torch.empty(nb, dtype=torch.uint8, device=dev)
            if wcache is not None and not keep:
                wcache["buf"] = wsplit
        a.wsplit, a.wsplit_bytes = wsplit.data_ptr(), nb
        st.route = "mega16" if use_mega else (("fused16/lean" if lean16 else "fused16") if use_fused16 else "general16")
        if use_fused16 and storage_f16() and not keep:
            # OPT-IN half storage of the message tensor between the steps (DMPNN_F_STORE16): not fp32-class, see include/dmpnn.h
            a.flags |= F_STORE16
            st.route = "fused16/f16-storage"
        elif use_mega and storage_f16() and not keep and not atom:
            # ... and on the whole-forward tile kernel (round 6): every product on the hi halves alone — operands, messages and weights as
            # one f16 per element under the same power-of-two scales, one MFMA pass instead of three (k_mpnn_tile16<..., LP>)
            a.flags |= F_STORE16
            st.route = "mega16/f16-operands"
    elif use_mega:
        a.flags |= F_MEGA
    if keep:
        a.flags |= F_KEEP
        if tiles_only:
            # training on a tile plan: K0 is the tile table alone; the kept tensors are in the caller's edge order and
            # dmpnn_backward runs the tile kernel on the caller's index arrays (it cannot see the plan's kind: this flag says so)
            if not (use_mega and want16 and not d_vd):
                raise RuntimeError("forward: a tile plan serves a training forward only on the tile kernel (f16 pipe), without W_d")
            a.flags |= F_TILE_PLAN
            # ... and for a ReLU-class activation without dropout the backward tile kernel needs only the SIGN of H0 / H^(t): one bit
            # per element straight from the matrix-pipe fragments (2 KB per tile and tensor instead of 57.6 KB of fp32 rows through an
            # LDS transpose); the fp32 slots stay allocated — a molecule beyond the tile keeps its rows there — but are not touched
            nbits = int(lib.dmpnn_forward_keep_bits_bytes(C.byref(a))) if keep_bits else 0
            if nbits:
                bits = torch.empty(nbits, dtype=torch.uint8, device=dev)
                a.keep_bits, a.keep_bits_bytes = bits.data_ptr(), nbits
    if launch:  # (launch=False: the argument block and the workspace only — dmpnn_train_step enqueues the forward itself)
        pend = getattr(plan, "pending", None)
        with _OnDevice(dev):
            if pend is not None and use_mega and want16 and tiles_only:
                # K0 was deferred to this call: tile table + weight pre-split (ONE launch) + tile kernel as one foreign call
                plan.pending = None
                _lib.check(lib.dmpnn_forward_tiles(C.byref(a), pend.data_ptr(), None, None, 0, plan.buf.numel() * 4, _stream_ptr(dev)), "dmpnn_forward_tiles")
            else:
                if pend is not None:
                    plan.ensure_launched()
                _lib.check(lib.dmpnn_forward(C.byref(a), _stream_ptr(dev)), "dmpnn_forward")
    st.args = a
    st.refs = (V, E, V_d, W_i, W_h, W_o, b_o, b_i, b_h, W_d, b_d, slope_t, edge_ws, atom_ws, spill_ws, split_ms, bits, wsplit)  # (wsplit last: nn._make_replay)
    st.dims = dict(d_v=d_v, d_e=d_e, d_h=d_h, d_vd=d_vd, has_bi=b_i is not None, has_bh=b_h is not None, atom=bool(atom))
    return out, st


def backward(st: ForwardState, gout: Tensor, need: dict, out: Optional[dict] = None, launch: bool = True, g_edge: Optional[Tensor] = None):
    """K6: parameter gradients of a kept forward.  ``need`` maps W_i/b_i/W_h/b_h/W_o/b_o/W_d/b_d -> bool.
    ``out``: optional ``{name: tensor}`` the kernels write the gradients INTO (contiguous fp32 of the parameter's shape, on
    the device — e.g. views of one flat gradient buffer, ``distributed.GradSync``); names missing there are allocated.
    ``g_edge``: a second gradient input, w.r.t. the kept ``H^(depth-1)`` rows (``dmpnn_bwd_args.g_edge``: the edge read-out of the
    mol-atom-bond blocks), ``[n_edges, d_h]`` in the kept tensors' row order — the backward tile kernel only."""
    from ._lib import BwdArgs

    lib = _lib.load()
    gout = _f32c(gout, "grad_output")
    dev = gout.device
    d = st.dims
    h, dv, de, dvd = d["d_h"], d["d_v"], d["d_e"], d["d_vd"]
    atom = bool(d.get("atom"))   # (AtomMessagePassing, base.py:278-289: W_i [d_h, d_v], W_h [d_h, d_e + d_h])
    shapes = dict(W_i=(h, dv if atom else dv + de), b_i=(h,), W_h=(h, h + de if atom else h), b_h=(h,), W_o=(h, dv + h), b_o=(h,),
                  W_d=(h + dvd, h + dvd), b_d=(h + dvd,))
    present = dict(W_i=True, b_i=d["has_bi"], W_h=True, b_h=d["has_bh"], W_o=True, b_o=True, W_d=dvd > 0, b_d=dvd > 0)
    def _buf(k):
        t = out.get(k) if out else None
        if (t is not None and t.dtype == torch.float32 and t.device == dev and tuple(t.shape) == shapes[k] and t.is_contiguous()):
            return t
        return torch.empty(shapes[k], dtype=torch.float32, device=dev)

    grads = {k: (_buf(k) if (need.get(k) and present[k]) else None) for k in shapes}
    b = BwdArgs()
    b.f = st.args
    b.gout, b.ldgout = gout.data_ptr(), gout.stride(0)
    for k, g in grads.items():
        setattr(b, "g" + k, _ptr(g))
    nbytes = lib.dmpnn_backward_ws_bytes(C.byref(st.args))
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dev)
    b.ws, b.ws_bytes = ws.data_ptr(), nbytes
    if g_edge is not None:
        g_edge = _f32c(g_edge, "g_edge").contiguous()
        if g_edge.shape[0] != st.args.n_edges or g_edge.shape[1] < h or g_edge.stride(0) % 4:
            raise RuntimeError("backward: g_edge must be [n_edges, d_h] with a row stride that is a multiple of 4")
        b.g_edge, b.ld_gedge = g_edge.data_ptr(), g_edge.stride(0)
    if not launch:  # (trainer.FusedTrainer: the argument block only; `refs` keeps the scratch alive)
        return grads, b, (ws, gout)
    with _OnDevice(dev):
        _lib.check(lib.dmpnn_backward(C.byref(b), _stream_ptr(dev)), "dmpnn_backward")
    return grads


def message_bwd(plan: GraphPlan, gM: Tensor) -> Tensor:
    gM = _f32c(gM, "gM")
    gH = torch.empty_like(gM)
    with _OnDevice(gM.device):
        _lib.check(_lib.load().dmpnn_message_bwd(plan.buf.data_ptr(), plan.n_atoms, plan.n_edges, gM.shape[1],
                                                 gM.data_ptr(), gM.stride(0), gH.data_ptr(), gH.stride(0),
                                                 _stream_ptr(gM.device)), "dmpnn_message_bwd")
    return gH


def aggregate_bwd(plan: GraphPlan, gMv: Tensor) -> Tensor:
    gMv = _f32c(gMv, "gMv")
    gH = torch.empty(plan.n_edges, gMv.shape[1], dtype=torch.float32, device=gMv.device)
    with _OnDevice(gMv.device):
        _lib.check(_lib.load().dmpnn_aggregate_bwd(plan.buf.data_ptr(), plan.n_atoms, plan.n_edges, gMv.shape[1],
                                                   gMv.data_ptr(), gMv.stride(0), gH.data_ptr(), gH.stride(0),
                                                   _stream_ptr(gMv.device)), "dmpnn_aggregate_bwd")
    return gH


def linear_wgrad(gZ: Tensor, A1: Tensor, A2: Optional[Tensor] = None, gather1: Optional[Tensor] = None,
                 want_bias: bool = False) -> tuple[Tensor, Optional[Tensor]]:
    """``gW = gZ^T @ [A1[gather] || A2]`` (and ``gb = gZ.sum(0)``) on the split-M MFMA kernel."""
    gZ = _f32c(gZ, "gZ")
    A1 = _f32c(A1, "A1")
    if A2 is not None:
        A2 = _f32c(A2, "A2")
    lib = _lib.load()
    M, N = int(gZ.shape[0]), int(gZ.shape[1])
    K1, K2 = int(A1.shape[1]), (int(A2.shape[1]) if A2 is not None else 0)
    g = GemmArgs()
    g.M, g.N, g.K1, g.K2 = M, N, K1, K2
    g.A1, g.lda1, g.gather1 = A1.data_ptr(), A1.stride(0), _ptr(gather1)
    g.gather1_rows = int(A1.shape[0])
    g.A2, g.lda2 = _ptr(A2), (A2.stride(0) if A2 is not None else 0)
    gW = torch.empty(N, K1 + K2, dtype=torch.float32, device=gZ.device)
    gb = torch.empty(N, dtype=torch.float32, device=gZ.device) if want_bias else None
    nbytes = lib.dmpnn_linear_wgrad_ws_bytes(M, N, K1 + K2, 1 if want_bias else 0)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=gZ.device)
    with _OnDevice(gZ.device):
        _lib.check(lib.dmpnn_linear_wgrad(C.byref(g), gZ.data_ptr(), gZ.stride(0), gW.data_ptr(), gW.stride(0),
                                          _ptr(gb), ws.data_ptr(), nbytes, _stream_ptr(gZ.device)), "dmpnn_linear_wgrad")
    return gW, gb
