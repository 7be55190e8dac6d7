"""Synthetic molecule-shaped graphs (no RDKit / datasets are available offline).

Shapes follow SURVEY.md §8(d): per molecule ``n_atoms ~ max(2, round(N(mu, 0.2 mu)))``; a random
tree (atom *i* bonds to a random earlier atom of degree < 4) plus ``Poisson(0.12 n)`` ring-closure
bonds (degree <= 4), so E/V ~ 2.1 as in QM9/ZINC.  Directed edges are emitted interleaved
``(u->v, v->u)`` per bond with ``rev = [1,0,3,2,...]`` exactly as the reference featurizers do
(``chemprop/featurizers/molgraph/molecule.py:75-89``); ``layout="block"`` reproduces the hand-built
layout of ``tests/unit/nn/test_message_passing.py:15-26`` and ``layout="shuffled"`` permutes the
edges of each molecule at random (still a valid graph).

Features are multi-hot in the v2 layout: atom blocks 38,7,6,5,6,8 + aromatic bit + 0.01*mass
(``chemprop/featurizers/atom.py:74-109``), bond = null bit, 4 bond types, conjugated, in-ring,
7 stereo (``chemprop/featurizers/bond.py:60-84``), identical for both directions of a bond.
``kind="cgr"`` gives the condensed-graph-of-reaction widths (d_v 106, d_e 28,
``chemprop/featurizers/molgraph/reaction.py:77-78``) with signed differences in the second half.
"""
from __future__ import annotations

import numpy as np

from .data import BatchMolGraph, MolGraph

_ATOM_BLOCKS = (38, 7, 6, 5, 6, 8)
SHAPES = {"qm9": 9.0, "zinc": 23.2, "synth40": 40.0, "cgr": 22.0}


def _atom_features(rng: np.random.Generator, n: int, d_v: int) -> np.ndarray:
    V = np.zeros((n, d_v), dtype=np.float32)
    off = 0
    for b in _ATOM_BLOCKS:
        if off + b > d_v:
            break
        # heavy-tailed choice so a few columns are hot, like real element/degree distributions
        idx = np.minimum((rng.exponential(1.5, size=n)).astype(np.int64), b - 1)
        V[np.arange(n), off + idx] = 1.0
        off += b
    if off + 2 <= d_v:
        V[:, off] = rng.integers(0, 2, size=n)
        V[:, off + 1] = rng.uniform(0.01, 0.35, size=n)
        off += 2
    if off < d_v:  # extra / CGR-difference columns: signed small integers
        V[:, off:] = rng.integers(-1, 2, size=(n, d_v - off)).astype(np.float32)
    return V


def _bond_features(rng: np.random.Generator, n_bonds: int, d_e: int) -> np.ndarray:
    Eb = np.zeros((n_bonds, d_e), dtype=np.float32)
    if d_e >= 14:
        Eb[np.arange(n_bonds), 1 + rng.integers(0, 4, size=n_bonds)] = 1.0
        Eb[:, 5] = rng.integers(0, 2, size=n_bonds)
        Eb[:, 6] = rng.integers(0, 2, size=n_bonds)
        Eb[np.arange(n_bonds), 7 + np.minimum(rng.exponential(0.5, size=n_bonds).astype(np.int64), 6)] = 1.0
        if d_e > 14:
            Eb[:, 14:] = rng.integers(-1, 2, size=(n_bonds, d_e - 14)).astype(np.float32)
    else:
        Eb[:] = rng.integers(0, 2, size=(n_bonds, d_e)).astype(np.float32)
    return Eb


def random_molgraph(rng: np.random.Generator, mu: float = 9.0, d_v: int = 72, d_e: int = 14,
                    layout: str = "interleaved", n_atoms: int | None = None) -> MolGraph:
    n = int(n_atoms) if n_atoms is not None else max(2, int(round(rng.normal(mu, 0.2 * mu))))
    deg = np.zeros(n, dtype=np.int64)
    bonds: list[tuple[int, int]] = []
    for i in range(1, n):
        cand = np.flatnonzero(deg[:i] < 4)
        if len(cand) == 0:  # cannot happen for a tree, kept for safety
            cand = np.arange(i)
        j = int(cand[rng.integers(len(cand))])
        bonds.append((j, i))
        deg[i] += 1
        deg[j] += 1
    have = set(bonds)
    for _ in range(int(rng.poisson(0.12 * n))):
        cand = np.flatnonzero(deg < 4)
        if len(cand) < 2:
            break
        u, v = (int(x) for x in rng.choice(cand, size=2, replace=False))
        u, v = min(u, v), max(u, v)
        if (u, v) in have:
            continue
        have.add((u, v))
        bonds.append((u, v))
        deg[u] += 1
        deg[v] += 1
    nb = len(bonds)
    b = np.asarray(bonds, dtype=np.int64).reshape(nb, 2)
    Eb = _bond_features(rng, nb, d_e)
    if layout == "block":
        src = np.concatenate([b[:, 0], b[:, 1]])
        dst = np.concatenate([b[:, 1], b[:, 0]])
        rev = np.concatenate([np.arange(nb, 2 * nb), np.arange(nb)])
        E = np.concatenate([Eb, Eb])
    else:
        src = np.stack([b[:, 0], b[:, 1]], axis=1).ravel()
        dst = np.stack([b[:, 1], b[:, 0]], axis=1).ravel()
        rev = np.arange(2 * nb).reshape(-1, 2)[:, ::-1].ravel()
        E = np.repeat(Eb, 2, axis=0)
        if layout == "shuffled" and nb:
            p = rng.permutation(2 * nb)  # new position k holds old edge p[k]
            inv = np.empty_like(p)
            inv[p] = np.arange(2 * nb)
            src, dst, E = src[p], dst[p], E[p]
            rev = inv[rev[p]]
    V = _atom_features(rng, n, d_v)
    return MolGraph(V, E.astype(np.float32), np.stack([src, dst]).astype(np.int64), rev.astype(np.int64))


def random_molgraphs(n_mols: int, kind: str = "qm9", seed: int = 0, layout: str = "interleaved",
                     d_v: int | None = None, d_e: int | None = None) -> list[MolGraph]:
    rng = np.random.default_rng(seed)
    mu = SHAPES[kind]
    if d_v is None:
        d_v = 106 if kind == "cgr" else 72
    if d_e is None:
        d_e = 28 if kind == "cgr" else 14
    return [random_molgraph(rng, mu, d_v, d_e, layout) for _ in range(n_mols)]


def random_batch(n_mols: int, kind: str = "qm9", seed: int = 0, layout: str = "interleaved",
                 d_v: int | None = None, d_e: int | None = None) -> BatchMolGraph:
    return BatchMolGraph(random_molgraphs(n_mols, kind, seed, layout, d_v, d_e))


def chain_molgraph(n: int, d_v: int = 72, d_e: int = 14) -> MolGraph:
    """The hand-built chain of ``tests/unit/nn/test_message_passing.py:15-26`` (all-ones features,
    block edge layout)."""
    V = np.ones((n, d_v), dtype=np.float32)
    E = np.ones((2 * (n - 1), d_e), dtype=np.float32)
    a = np.arange(n - 1)
    edge_index = np.stack([np.concatenate([a, a + 1]), np.concatenate([a + 1, a])]).astype(np.int64)
    rev = np.concatenate([np.arange(n - 1, 2 * n - 2), np.arange(n - 1)]).astype(np.int64)
    return MolGraph(V, E, edge_index, rev)
