"""TEST INFRASTRUCTURE — numpy restatement of the ARITHMETIC the f16-pipe kernels use for an fp32 contraction
(chemprop_amd/csrc/dmpnn_mega16_impl.hpp: scale_for, split4, k_split_weights, contract; dmpnn_rows16_impl.hpp): the
"exact 3-term split".  Not a reference of chemprop (the reference computes ``F.linear`` in fp32); it pins, on the CPU,
the claim the bench line's ``dtype`` makes — fp32-class accuracy from three f16 MFMA passes.  Only tests/ may import it.

    s        = 2^(14 - e)  with  max|x| = m 2^e, m in [0.5, 1)          power of two: scaling is exact, max|x s| in [2^13, 2^14)
    x s      = hi + lo + r,   hi = f16(x s),  lo = f16(x s - hi)          |lo| <= 2^-11 |hi|,  |r| <= 2^-22 |x s|  (or one f16
                                                                          subnormal step 2^-24 for tiny entries)
    (a s_a)(w s_w) ~ hi_a hi_w + hi_a lo_w + lo_a hi_w                    every product of two 11-bit significands is exact
                                                                          in fp32; the dropped lo_a lo_w is <= 2^-22 relative
    out      = fp32 accumulation of the three passes  /  (s_a s_w)        exact un-scaling

Activations take one scale per row tile (the tile kernel: the tile's rows), weights one scale per output row.
"""
from __future__ import annotations

import numpy as np


def scale_for(maxabs: float) -> float:
    """dmpnn_mega16_impl.hpp:56-62."""
    if not (maxabs > 0.0) or not np.isfinite(maxabs):
        return 1.0
    _, e = np.frexp(np.float32(maxabs))
    return float(np.ldexp(np.float32(1.0), 14 - int(e)))


def split(x: np.ndarray, s) -> tuple[np.ndarray, np.ndarray]:
    """(hi, lo) f16 pair of ``x * s`` (split4, :63-67; k_split_weights :99-102)."""
    a = (x.astype(np.float32) * np.float32(s)).astype(np.float32)
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def linear_split16(A: np.ndarray, W: np.ndarray, tile_rows: int = 48) -> np.ndarray:
    """``A @ W.T`` ([M,K] x [N,K]) in the kernels' arithmetic: per-tile activation scale, per-row weight scale, three
    f16 x f16 -> fp32 passes (hi hi, hi lo, lo hi), fp32 accumulation, exact un-scaling."""
    A = np.asarray(A, np.float32)
    W = np.asarray(W, np.float32)
    sW = np.array([scale_for(float(np.abs(W[n]).max())) for n in range(W.shape[0])], np.float32)
    Wh, Wl = split(W, sW[:, None])
    Whf, Wlf = Wh.astype(np.float32), Wl.astype(np.float32)
    out = np.empty((A.shape[0], W.shape[0]), np.float32)
    for r0 in range(0, A.shape[0], tile_rows):
        At = A[r0:r0 + tile_rows]
        sA = np.float32(scale_for(float(np.abs(At).max()) if At.size else 0.0))
        Ah, Al = split(At, sA)
        Ahf, Alf = Ah.astype(np.float32), Al.astype(np.float32)
        acc = (Ahf @ Whf.T + Ahf @ Wlf.T + Alf @ Whf.T).astype(np.float32)
        out[r0:r0 + tile_rows] = acc / (sA * sW[None, :])
    return out
