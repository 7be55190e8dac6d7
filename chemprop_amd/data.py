"""Host-side mirror of the reference's hot-path INPUT contract.

* :class:`MolGraph`       mirrors ``chemprop/data/molgraph.py:6-16`` (numpy, one molecule).
* :class:`BatchMolGraph`  mirrors ``chemprop/data/collate.py:13-73`` (five torch tensors + size).

The engine itself is duck-typed: anything exposing ``V, E, edge_index, rev_edge_index, batch`` works
(the reference's own ``BatchMolGraph`` and ``BatchCuikMolGraph`` —
``chemprop/featurizers/molgraph/molecule.py:95-123`` — included).  These classes exist so the package
is usable, testable and benchmarkable where chemprop itself is not installed (the GPU box).

Batching semantics reproduced (collate.py:37-62): molecules are concatenated in order; atom ids in
``edge_index`` are shifted by the running atom count, ``rev_edge_index`` by the running edge count;
``batch[a]`` is the molecule id of atom ``a`` (non-decreasing); features become ``float32``,
indices ``int64``.
"""
from __future__ import annotations

from typing import NamedTuple, Sequence

import numpy as np
import torch
from torch import Tensor


class MolGraph(NamedTuple):
    """Graph featurization of one molecule (numpy)."""

    V: np.ndarray  # [n_atoms, d_v]
    E: np.ndarray  # [n_edges, d_e]   (two directed edges per bond)
    edge_index: np.ndarray  # [2, n_edges]  row 0 = source atom, row 1 = destination atom
    rev_edge_index: np.ndarray  # [n_edges]  id of the reverse directed edge


TILE_MAX_ATOMS, TILE_MAX_EDGES = 32, 48  # kMegaBA / kMegaBM of csrc/dmpnn_common.hpp


def molecules_oversize(n_atoms, n_edges) -> bool:
    """True when a molecule of the batch exceeds the tile of the whole-forward tile kernel (host knowledge, free while batching)."""
    n_atoms, n_edges = np.asarray(n_atoms), np.asarray(n_edges)
    return bool(n_atoms.size and (int(n_atoms.max()) > TILE_MAX_ATOMS or int(n_edges.max()) > TILE_MAX_EDGES))


class BatchMolGraph:
    """A batch of :class:`MolGraph` as five tensors.  ``len()`` is the number of molecules."""

    __slots__ = ("V", "E", "edge_index", "rev_edge_index", "batch", "_size", "tiles", "oversize")

    def __init__(self, mgs: Sequence[MolGraph]):
        self._size = len(mgs)
        self.tiles = None  # (tile_row, tile_atom, n_tiles) device int32 views: only batches from PackedBatch.to_device
        n_atoms = np.fromiter((len(mg.V) for mg in mgs), dtype=np.int64, count=len(mgs))
        n_edges = np.fromiter((mg.edge_index.shape[1] for mg in mgs), dtype=np.int64, count=len(mgs))
        # what the host knows for free while batching: does a molecule exceed the whole-forward tile kernel's tile?
        # (True: this batch takes the per-step routes; False: the tile kernel with no oversize molecule in sight;
        # None — batches that arrive as bare tensors — : the kernel's own generic path covers whatever turns up)
        self.oversize = molecules_oversize(n_atoms, n_edges)
        atom_off = np.concatenate([[0], np.cumsum(n_atoms)[:-1]]) if len(mgs) else np.zeros(0, np.int64)
        edge_off = np.concatenate([[0], np.cumsum(n_edges)[:-1]]) if len(mgs) else np.zeros(0, np.int64)

        self.V = torch.from_numpy(np.concatenate([mg.V for mg in mgs])).float()
        self.E = torch.from_numpy(np.concatenate([mg.E for mg in mgs])).float()
        ei = np.hstack([mg.edge_index + o for mg, o in zip(mgs, atom_off)])
        self.edge_index = torch.from_numpy(ei).long()
        rev = np.concatenate([mg.rev_edge_index + o for mg, o in zip(mgs, edge_off)])
        self.rev_edge_index = torch.from_numpy(rev).long()
        self.batch = torch.from_numpy(np.repeat(np.arange(len(mgs), dtype=np.int64), n_atoms))

    def __len__(self) -> int:
        return self._size

    def to(self, device) -> None:
        """In-place device move, returns ``None`` exactly like collate.py:68-73."""
        self.V = self.V.to(device)
        self.E = self.E.to(device)
        self.edge_index = self.edge_index.to(device)
        self.rev_edge_index = self.rev_edge_index.to(device)
        self.batch = self.batch.to(device)
        self.tiles = None  # (views of the packed buffer on its own device)

    @classmethod
    def from_tensors(cls, V: Tensor, E: Tensor, edge_index: Tensor, rev_edge_index: Tensor,
                     batch: Tensor, size: int | None = None) -> "BatchMolGraph":
        """Wrap already-batched tensors (the cuik-molmaker style hand-off, molecule.py:95-123)."""
        self = object.__new__(cls)
        self.V, self.E = V, E
        self.edge_index, self.rev_edge_index, self.batch = edge_index, rev_edge_index, batch
        self._size = int(size) if size is not None else (int(batch[-1]) + 1 if batch.numel() else 0)
        self.tiles = None
        self.oversize = None
        return self

    def __copy__(self):
        b = BatchMolGraph.from_tensors(self.V, self.E, self.edge_index, self.rev_edge_index, self.batch, self._size)
        b.tiles = self.tiles  # (a graph_transform scales V / E of a shallow copy: the connectivity is the same)
        b.oversize = self.oversize
        return b


# ---- f3: one-buffer wire format + device-side batching (collate.py:37-62,68-73) -----------------
_WIRE_MAGIC = 0x31424D44  # "DMB1"
_WIRE_HEADER = 8          # int64 words: magic, n_mols, n_atoms, n_edges, d_v, d_e, n_tiles (-1: no table), rows of E


def _align16(n: int) -> int:
    return (n + 15) // 16 * 16


class PackedBatch:
    """A batch of :class:`MolGraph` in ONE host buffer, built in the DataLoader worker (``collate_fn``):

        int64  header[8]            magic, n_mols, n_atoms, n_edges, d_v, d_e, n_tiles (-1: no tile table), rows of E
        int32  atom_off[n_mols+1]   running atom count   (collate.py:46,56: ``num_nodes``)
        int32  edge_off[n_mols+1]   running edge count   (collate.py:47,57: ``num_edges``)
        int32  src[n_edges] | dst[n_edges] | rev[n_edges]      molecule-LOCAL ids, as the featurizer made them
        f32    V[n_atoms, d_v] | E[rows of E, d_e]   (the reference concatenates ``mg.E`` as given, collate.py:59: one row per
                                    directed edge from its featurizers, whatever a hand-built MolGraph holds)
        int32  tile_row[n_tiles+1] | tile_atom[n_tiles+1]     whole molecules packed greedily into tiles of <= 48 directed
                                    edges / <= 32 atoms (``dmpnn_pack_tiles``, a host function of the library): the tile
                                    plan of the whole-forward tile kernel, made where the molecule sizes are known
                                    anyway — the forward then needs no plan kernel (K0) beyond a copy of this table and
                                    takes the tile kernel at ANY batch size; absent when a molecule exceeds a tile

    every section 16-byte aligned.  Against the reference's hand-off (five tensors, int64 indices, five copies) this
    is one copy and 12 instead of 24 index bytes per directed edge; ``to_device`` issues that copy (asynchronous
    when the buffer is pinned) and ONE ``dmpnn_collate`` launch, and returns a :class:`BatchMolGraph` whose
    ``V`` / ``E`` are views of the copied buffer and whose index tensors are the int64 tensors the reference builds,
    bit for bit.  No arithmetic of the batching happens on the host beyond the two running sums."""

    __slots__ = ("buf", "n_mols", "n_atoms", "n_edges", "n_erows", "d_v", "d_e", "sections", "n_tiles", "oversize")

    def __init__(self, mgs: Sequence[MolGraph], pin: bool = False, tiles: bool = True):
        n_mols = len(mgs)
        n_at = np.fromiter((len(mg.V) for mg in mgs), dtype=np.int64, count=n_mols)
        n_ed = np.fromiter((mg.edge_index.shape[1] for mg in mgs), dtype=np.int64, count=n_mols)
        nV, nE = int(n_at.sum()), int(n_ed.sum())
        nEr = int(sum(len(mg.E) for mg in mgs))  # rows of E as given (== nE for featurizer-made graphs)
        if nV >= 2 ** 31 or nE >= 2 ** 31:
            raise ValueError("PackedBatch: more than 2^31 atoms or edges in one batch")
        d_v = int(mgs[0].V.shape[1]) if n_mols else 0
        d_e = int(mgs[0].E.shape[1]) if n_mols else 0
        atom_off = np.zeros(n_mols + 1, dtype=np.int32)
        edge_off = np.zeros(n_mols + 1, dtype=np.int32)
        atom_off[1:] = np.cumsum(n_at)
        edge_off[1:] = np.cumsum(n_ed)
        n_tiles, trow, tatom = -1, None, None
        self.oversize = molecules_oversize(n_at, n_ed)  # such a batch ships no table and takes the per-step routes
        if tiles and n_mols and nV and not self.oversize:
            from . import _lib

            lib = _lib.load()
            cap = int(lib.dmpnn_max_tiles(nV, nE)) + 1
            trow, tatom = np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.int32)
            n_tiles = int(lib.dmpnn_pack_tiles(atom_off.ctypes.data, edge_off.ctypes.data, n_mols, trow.ctypes.data,
                                               tatom.ctypes.data, cap))
            if n_tiles < 0:
                n_tiles = -1  # (table capacity: the device plans decide)
        nt = n_tiles + 1 if n_tiles >= 0 else 0
        sec, o = {}, _WIRE_HEADER * 8
        for name, nbytes in (("atom_off", 4 * (n_mols + 1)), ("edge_off", 4 * (n_mols + 1)), ("src", 4 * nE), ("dst", 4 * nE),
                             ("rev", 4 * nE), ("V", 4 * nV * d_v), ("E", 4 * nEr * d_e), ("tile_row", 4 * nt), ("tile_atom", 4 * nt)):
            sec[name] = (o, nbytes)
            o = _align16(o + nbytes)
        # (numpy's calloc-backed zeros, not torch.zeros: no intra-op thread pool is woken for a 2 MB fill in a loader worker)
        if pin:
            buf = torch.empty(o, dtype=torch.uint8).pin_memory()
            raw = buf.numpy()
            raw[:] = 0
        else:
            raw = np.zeros(o, dtype=np.uint8)
            buf = torch.from_numpy(raw)
        view = lambda name, dt: raw[sec[name][0]:sec[name][0] + sec[name][1]].view(dt)
        raw[:_WIRE_HEADER * 8].view(np.int64)[:8] = (_WIRE_MAGIC, n_mols, nV, nE, d_v, d_e, n_tiles, nEr)
        view("atom_off", np.int32)[:] = atom_off
        view("edge_off", np.int32)[:] = edge_off
        if nt:
            view("tile_row", np.int32)[:] = trow[:nt]
            view("tile_atom", np.int32)[:] = tatom[:nt]
        if n_mols:
            if nE:
                view("src", np.int32)[:] = np.concatenate([mg.edge_index[0] for mg in mgs])
                view("dst", np.int32)[:] = np.concatenate([mg.edge_index[1] for mg in mgs])
                view("rev", np.int32)[:] = np.concatenate([mg.rev_edge_index for mg in mgs])
            if nEr and d_e:
                view("E", np.float32).reshape(nEr, d_e)[:] = np.concatenate([mg.E for mg in mgs])
            if nV and d_v:
                view("V", np.float32).reshape(nV, d_v)[:] = np.concatenate([mg.V for mg in mgs])
        self.buf, self.sections = buf, sec
        self.n_mols, self.n_atoms, self.n_edges, self.d_v, self.d_e = n_mols, nV, nE, d_v, d_e
        self.n_tiles, self.n_erows = n_tiles, nEr

    def __len__(self) -> int:
        return self.n_mols

    def pin_memory(self) -> "PackedBatch":
        """(torch's DataLoader calls this on custom batch types when ``pin_memory=True``)"""
        self.buf = self.buf.pin_memory()
        return self

    def to_device(self, device) -> "BatchMolGraph":
        """One host-to-device copy + one ``dmpnn_collate`` launch -> the five tensors of collate.py:58-62 on ``device``."""
        from . import _lib, engine

        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("PackedBatch.to_device: batching runs in a HIP kernel; use BatchMolGraph(mgs) on the host")
        dbuf = self.buf.to(device, non_blocking=True)
        nV, nE = self.n_atoms, self.n_edges
        sl = lambda name: dbuf[self.sections[name][0]:self.sections[name][0] + self.sections[name][1]]
        V = sl("V").view(torch.float32).view(nV, self.d_v)
        E = sl("E").view(torch.float32).view(self.n_erows, self.d_e)
        edge_index = torch.empty(2, nE, dtype=torch.int64, device=device)
        rev = torch.empty(nE, dtype=torch.int64, device=device)
        batch = torch.empty(nV, dtype=torch.int64, device=device)
        base = dbuf.data_ptr()
        at = lambda name: base + self.sections[name][0]
        with engine._OnDevice(device):
            _lib.check(_lib.load().dmpnn_collate(at("atom_off"), at("edge_off"), self.n_mols, at("src"), at("dst"), at("rev"),
                                                 nV, nE, edge_index.data_ptr(), rev.data_ptr(), batch.data_ptr(),
                                                 engine._stream_ptr(device)), "dmpnn_collate")
        bmg = BatchMolGraph.from_tensors(V, E, edge_index, rev, batch, self.n_mols)
        bmg.oversize = self.oversize
        if self.n_tiles >= 0:
            bmg.tiles = (sl("tile_row").view(torch.int32), sl("tile_atom").view(torch.int32), self.n_tiles)
        return bmg
