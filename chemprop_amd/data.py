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


class BatchMolGraph:
    """A batch of :class:`MolGraph` as five tensors.  ``len()`` is the number of molecules."""

    __slots__ = ("V", "E", "edge_index", "rev_edge_index", "batch", "_size")

    def __init__(self, mgs: Sequence[MolGraph]):
        self._size = len(mgs)
        n_atoms = np.fromiter((len(mg.V) for mg in mgs), dtype=np.int64, count=len(mgs))
        n_edges = np.fromiter((mg.edge_index.shape[1] for mg in mgs), dtype=np.int64, count=len(mgs))
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

    @classmethod
    def from_tensors(cls, V: Tensor, E: Tensor, edge_index: Tensor, rev_edge_index: Tensor,
                     batch: Tensor, size: int | None = None) -> "BatchMolGraph":
        """Wrap already-batched tensors (the cuik-molmaker style hand-off, molecule.py:95-123)."""
        self = object.__new__(cls)
        self.V, self.E = V, E
        self.edge_index, self.rev_edge_index, self.batch = edge_index, rev_edge_index, batch
        self._size = int(size) if size is not None else (int(batch[-1]) + 1 if batch.numel() else 0)
        return self

    def __copy__(self):
        return BatchMolGraph.from_tensors(self.V, self.E, self.edge_index, self.rev_edge_index,
                                          self.batch, self._size)
