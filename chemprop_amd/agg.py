"""f1: the aggregation right after the message-passing block inside ``MPNN.fingerprint``
(``models/model.py:131``) on the MI355X kernels — mirrors of ``chemprop/nn/agg.py``.

``MeanAggregation`` / ``SumAggregation`` / ``NormAggregation`` (``agg.py:66-113``) are ONE segment
reduction over the sorted ``batch`` vector (``dmpnn_molagg_*``: no ``[V, h]`` int64 index, rows added in
increasing atom order = the reference's sequential scatter order, so results are bit-identical);
``AttentiveAggregation`` (``agg.py:116-133``) is the same reduction twice around its small linear layer.
Same constructor arguments, ``hparams`` and (for the attentive variant) ``state_dict`` keys as the
reference.  No CPU fallback: tensors must live on a HIP device.

``n_mols``: the reference reads ``batch.max()`` back from the device on every call.  When the batch tensor
is the one the message-passing block just saw (``bond_message_passing_forward`` notes ``len(bmg)`` for it),
that host sync is skipped; otherwise it is read back exactly like the reference does.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional

import torch
from torch import Tensor, nn

from . import _lib, engine

MODES = {"sum": 0, "mean": 1, "norm": 2}

_noted = None  # (weakref to the batch tensor, n_mols) of the most recent BatchMolGraph seen by the block


def note_batch(batch: Tensor, n_mols: int) -> None:
    """Remember the molecule count of ``batch`` (identity of the tensor object, not its address)."""
    global _noted
    try:
        _noted = (weakref.ref(batch), int(n_mols))
    except TypeError:  # pragma: no cover
        _noted = None


def n_molecules(batch: Tensor) -> int:
    if _noted is not None and _noted[0]() is batch and _noted[1] > 0:
        return _noted[1]
    if batch.numel() == 0:
        return 0
    return int(batch.max()) + 1  # agg.py:75: the reference's own host read


class MolBounds:
    """First / one-past-last atom of every molecule and the validity flag, in device memory."""

    def __init__(self, batch: Tensor, n_mols: int):
        engine._require_device(batch, "batch")
        if batch.dtype != torch.int64:
            batch = batch.to(torch.int64)
        self.batch = batch.contiguous()
        self.n_atoms, self.n_mols = int(batch.numel()), int(n_mols)
        lib = _lib.load()
        nb = int(lib.dmpnn_molagg_ws_bytes(self.n_mols))
        self.ws = torch.empty(nb, dtype=torch.uint8, device=batch.device)
        with engine._OnDevice(batch.device):
            _lib.check(lib.dmpnn_molagg_bounds(self.batch.data_ptr(), self.n_atoms, self.n_mols, self.ws.data_ptr(), nb,
                                               engine._stream_ptr(batch.device)), "dmpnn_molagg_bounds")


def mol_reduce(H: Tensor, b: MolBounds, mode: str = "sum", norm: float = 1.0) -> Tensor:
    H = engine._f32c(H, "H")
    if H.dim() != 2 or H.shape[0] != b.n_atoms:
        raise ValueError(f"H must be [n_atoms={b.n_atoms}, d]; got {tuple(H.shape)}")
    out = torch.empty(b.n_mols, H.shape[1], dtype=torch.float32, device=H.device)
    lib = _lib.load()
    with engine._OnDevice(H.device):
        _lib.check(lib.dmpnn_molagg_fwd(H.data_ptr(), H.stride(0), b.n_atoms, H.shape[1], b.n_mols, b.ws.data_ptr(),
                                        MODES[mode], C.c_float(norm), out.data_ptr(), out.stride(0) if b.n_mols else H.shape[1],
                                        engine._stream_ptr(H.device)), "dmpnn_molagg_fwd")
    return out


def mol_expand(G: Tensor, b: MolBounds, mode: str = "sum", norm: float = 1.0) -> Tensor:
    """Rows of ``G [n_mols, d]`` gathered back to the atoms (the transpose of :func:`mol_reduce`)."""
    G = engine._f32c(G, "G")
    out = torch.empty(b.n_atoms, G.shape[1], dtype=torch.float32, device=G.device)
    lib = _lib.load()
    with engine._OnDevice(G.device):
        _lib.check(lib.dmpnn_molagg_bwd(G.data_ptr(), G.stride(0) if b.n_mols else G.shape[1], b.batch.data_ptr(), b.n_atoms,
                                        G.shape[1], b.n_mols, b.ws.data_ptr(), MODES[mode], C.c_float(norm), out.data_ptr(),
                                        out.stride(0) if b.n_atoms else G.shape[1], engine._stream_ptr(G.device)),
                   "dmpnn_molagg_bwd")
    return out


class _Reduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, H, b, mode, norm):
        ctx.b, ctx.mode, ctx.norm = b, mode, norm
        return mol_reduce(H, b, mode, norm)

    @staticmethod
    def backward(ctx, g):
        return mol_expand(g.contiguous(), ctx.b, ctx.mode, ctx.norm), None, None, None


class _Expand(torch.autograd.Function):
    @staticmethod
    def forward(ctx, G, b):
        ctx.b = b
        return mol_expand(G, b)

    @staticmethod
    def backward(ctx, g):
        return mol_reduce(g.contiguous(), ctx.b), None


def aggregation_forward(mod, H: Tensor, batch: Tensor, mode: str) -> Tensor:
    """``forward`` of any module with the reference's attributes (``norm`` for the norm variant, ``W`` for the
    attentive one); shared by the mirrors below and by the subclasses of the real chemprop classes
    (``chemprop_amd.integration``)."""
    engine._require_device(H, "H")
    b = MolBounds(batch, n_molecules(batch))
    if mode == "attentive":
        from .backward import linear_fn

        logits = linear_fn(H, mod.W.weight, mod.W.bias).exp()
        Z = _Reduce.apply(logits, b, "sum", 1.0)
        alphas = logits / _Expand.apply(Z, b)
        return _Reduce.apply(alphas * H, b, "sum", 1.0)
    return _Reduce.apply(H, b, mode, float(getattr(mod, "norm", 1.0)))


class Aggregation(nn.Module):
    """``chemprop.nn.agg.Aggregation`` (agg.py:19-57): ``forward(H [V, d], batch [V]) -> [n_mols, d]``."""

    mode = "sum"

    def __init__(self, dim: int = 0, *args, **kwargs):
        super().__init__()
        if dim != 0:
            raise NotImplementedError("chemprop_amd: aggregation over dim 0 (atoms) only, like every caller of the reference")
        self.dim = dim
        self.hparams = {"dim": dim, "cls": self.__class__}

    def forward(self, H: Tensor, batch: Tensor) -> Tensor:
        return aggregation_forward(self, H, batch, self.mode)


class MeanAggregation(Aggregation):
    """agg.py:66-80."""
    mode = "mean"


class SumAggregation(Aggregation):
    """agg.py:83-98."""
    mode = "sum"


class NormAggregation(SumAggregation):
    """agg.py:101-113: sum / ``norm``."""
    mode = "norm"

    def __init__(self, dim: int = 0, *args, norm: float = 100.0, **kwargs):
        super().__init__(dim, **kwargs)
        self.norm = norm
        self.hparams["norm"] = norm


class AttentiveAggregation(Aggregation):
    """agg.py:116-133: ``alpha_v = exp(W h_v) / sum_{u in mol} exp(W h_u)``; ``out = sum_v alpha_v h_v``."""

    def __init__(self, dim: int = 0, *args, output_size: int, **kwargs):
        super().__init__(dim, *args, **kwargs)
        self.hparams["output_size"] = output_size
        self.W = nn.Linear(output_size, 1)

    mode = "attentive"


REGISTRY = {"mean": MeanAggregation, "sum": SumAggregation, "norm": NormAggregation}
