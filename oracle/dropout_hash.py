"""TEST INFRASTRUCTURE — the dropout mask of the engine's fused dropout, restated in numpy.

chemprop's dropout is ``nn.Dropout`` (``base.py:85,139,182``): a Bernoulli(1 - p) mask scaled by 1 / (1 - p), drawn from torch's
generator.  A mask drawn inside a HIP kernel cannot reproduce that stream (the reference on a GPU does not reproduce its own CPU
stream either), so the engine defines its own: a counter-based hash of ``(seed, site, row, column)`` (``csrc/dmpnn_common.hpp:
drop_hash``), documented in ``include/dmpnn.h`` (``dmpnn_fwd_args.dropout_p``).  This module restates it so that the tests can
replay the engine's exact masks in the EXECUTED reference: parity of a stochastic op is parity given the mask.  Pinned on the
CPU against the library's own host function ``dmpnn_dropout_keep`` (``tests/test_host.py``).
"""
from __future__ import annotations

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & M32


def drop_hash(seed: int, site: int, rows, cols) -> np.ndarray:
    """uint32 hash of every (row, col) pair: ``rows`` [R], ``cols`` [C] -> [R, C]  (all arithmetic modulo 2^32)."""
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    r = _u32(rows).reshape(-1, 1)
    c = _u32(cols).reshape(1, -1)
    h = _u32(r * np.uint64(1024) + c) ^ lo ^ _u32(np.uint64(site) * np.uint64(0x9E3779B9))
    h ^= h >> np.uint64(16)
    h = _u32(h * np.uint64(0x7FEB352D))
    h ^= h >> np.uint64(15)
    h = _u32(h + hi)
    h = _u32(h * np.uint64(0x846CA68B))
    h ^= h >> np.uint64(16)
    h = _u32(h * np.uint64(0x9E3779B1))
    h ^= h >> np.uint64(15)
    return h.astype(np.uint32)


def threshold(p: float) -> int:
    """floor(p 2^32) computed like the library (float32 p widened to double)."""
    t = float(np.float32(p)) * 4294967296.0
    return 4294967295 if t >= 4294967295.0 else int(t)


def keep_mask(seed: int, site: int, n_rows: int, n_cols: int, p: float, rows=None) -> np.ndarray:
    """Boolean ``[n_rows, n_cols]``: True where the element is KEPT (and scaled by 1 / (1 - p)).  ``rows``: the row ids (the
    plan's row of an edge for the update sites, the atom id for the finalize site); default ``arange(n_rows)``."""
    rows = np.arange(n_rows) if rows is None else np.asarray(rows)
    return drop_hash(seed, site, rows, np.arange(n_cols)) >= np.uint32(threshold(p))
