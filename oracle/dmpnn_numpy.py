"""ORACLE (test infrastructure, never shipped, never measured as the product).

Second, independent CPU restatement of the same path in numpy, written in the *CSR / atom-centric
form the HIP kernels use* rather than the scatter form the reference uses.  It exists to prove —
on CPU, against the executed reference — that the restructured algorithm is the same function:

  * stable counting sort of edges by destination  ->  ``row_ptr[V+1]``, ``perm[E]``
    (reference has no counterpart; it re-derives a dense ``[E,h]`` int64 index every call,
    ``mixins.py:12``, ``base.py:208``).  Stability keeps each atom's summation order equal to the
    reference's sequential ``scatter_reduce_`` order (increasing edge id), so the segment sums are
    bit-identical to ATen's on CPU.
  * atom-centric message: for atom ``v`` with incoming edges ``e'_1..e'_d``:
    ``S = ((H[e'_1] + H[e'_2]) + ...)``, then ``M[rev(e'_i)] = S - H[e'_i]``.  For a valid molecular
    graph (``rev`` an involution with ``src(rev(e)) == dst(e)``) this equals
    ``M[e] = S[src(e)] - H[rev(e)]`` (``mixins.py:11-18``) with every ``H`` row read exactly once.
  * analytic backward of the whole block (what ``K6`` implements), checked against torch autograd of
    :mod:`oracle.dmpnn_torch`.

Only ``tests/`` may import this file.
"""
from __future__ import annotations

import numpy as np


def build_csr(dst: np.ndarray, n_atoms: int):
    """Incoming-edge CSR by destination atom; stable (edge ids ascending inside a row)."""
    dst = np.asarray(dst, dtype=np.int64)
    perm = np.argsort(dst, kind="stable").astype(np.int64)
    counts = np.bincount(dst, minlength=n_atoms).astype(np.int64)
    row_ptr = np.zeros(n_atoms + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    return row_ptr, perm


def graph_is_symmetric(src, dst, rev) -> bool:
    """The invariants every featurizer-produced graph satisfies: ``rev`` is an
    involution and the reverse edge runs dst -> src."""
    src, dst, rev = (np.asarray(a, dtype=np.int64) for a in (src, dst, rev))
    if len(rev) == 0:
        return True
    if rev.min() < 0 or rev.max() >= len(rev):
        return False
    return bool(np.all(rev[rev] == np.arange(len(rev))) and np.all(src[rev] == dst) and np.all(dst[rev] == src))


def row_coordinates(src, dst, rev, perm):
    """The graph in CSR-row coordinates (row i = edge perm[i]; what the fused forward's edge tensors
    use): ``inv`` (edge -> row), ``srcp/dstp`` (atoms of row i), ``revp`` (row of the reverse edge)."""
    src, dst, rev, perm = (np.asarray(a, dtype=np.int64) for a in (src, dst, rev, perm))
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    return dict(inv=inv, srcp=src[perm], dstp=dst[perm], revp=inv[rev[perm]])


def tile_tables(row_ptr, n_edges, n_slots, bm=48, max_deg_supported=24):
    """Row tiles of WHOLE atoms for the fused contraction (restates csrc/dmpnn_prepare.hip
    ``tile_geom`` / ``write_tiles``): nominal stride ``b0 = bm - maxdeg + 1``; atom ``v`` belongs to
    tile ``min(row_ptr[v] // b0, n_tiles - 1)``; a tile therefore spans at most ``bm`` rows.
    Returns (tile_row[n_slots + 2], tile_atom[n_slots + 2], n_tiles, b0)."""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    n_atoms = len(row_ptr) - 1
    deg = np.diff(row_ptr)
    maxdeg = int(deg.max()) if n_atoms else 0
    md = max(maxdeg, 1)
    b0 = bm - md + 1
    n_tiles = 0 if (md > max_deg_supported or n_edges == 0) else -(-n_edges // b0)
    tile_row = np.full(n_slots + 2, n_edges, dtype=np.int64)
    tile_atom = np.full(n_slots + 2, n_atoms, dtype=np.int64)
    if n_tiles:
        t_of = np.minimum(row_ptr[:-1] // b0, n_tiles - 1)
        for t in range(n_tiles):
            atoms = np.flatnonzero(t_of >= t)
            if len(atoms):
                tile_atom[t] = atoms[0]
                tile_row[t] = row_ptr[atoms[0]]
    return tile_row, tile_atom, n_tiles, b0


def piece_tiles(src, dst, row_ptr, n_slots, bm=48, ba=32, block=64):
    """Row tiles made of WHOLE connected pieces (restates ``build_piece_tiles`` of
    csrc/dmpnn_prepare.hip): a cut after atom ``v`` is safe when no edge joins atoms <= v with atoms
    > v; consecutive pieces are packed greedily into tiles of <= ``bm`` rows and <= ``ba`` atoms, and a
    tile also starts at every ``block``-th piece (the kernel walks blocks of pieces in parallel).
    Returns (mtile_row[n_slots + 2], mtile_atom[n_slots + 2], n_tiles); n_tiles = -1 if the table does not
    hold them.  A piece that alone exceeds a tile gets a tile of its own (the kernels' generic path)."""
    src, dst, row_ptr = (np.asarray(a, dtype=np.int64) for a in (src, dst, row_ptr))
    n_atoms, n_edges = len(row_ptr) - 1, len(src)
    mrow = np.full(n_slots + 2, n_edges, dtype=np.int64)
    matom = np.full(n_slots + 2, n_atoms, dtype=np.int64)
    maxnbr = np.arange(n_atoms, dtype=np.int64)
    if n_edges:
        np.maximum.at(maxnbr, dst, src)
        np.maximum.at(maxnbr, src, dst)
    pm = np.maximum.accumulate(maxnbr) if n_atoms else maxnbr
    starts = [u for u in range(n_atoms) if u == 0 or pm[u - 1] == u - 1] + [n_atoms]
    n_pieces = len(starts) - 1
    tiles, p = [], 0
    while p < n_pieces:
        v = starts[p]
        q = p + 1
        lim = min(n_pieces, (p // block + 1) * block)
        # (a piece that alone exceeds a tile is a tile of its own: the tile kernels run their generic path on it)
        if not (starts[p + 1] - v > ba or row_ptr[starts[p + 1]] - row_ptr[v] > bm):
            while q < lim and starts[q + 1] - v <= ba and row_ptr[starts[q + 1]] - row_ptr[v] <= bm:
                q += 1
        tiles.append(v)
        p = q
    if len(tiles) > n_slots:
        return mrow, matom, -1
    for t, a in enumerate(tiles):
        matom[t] = a
        mrow[t] = row_ptr[a]
    return mrow, matom, len(tiles)


def segment_sum_csr(H: np.ndarray, row_ptr: np.ndarray, perm: np.ndarray) -> np.ndarray:
    """S[v] = sum of H[perm[row_ptr[v]:row_ptr[v+1]]] taken left to right in float32."""
    n_atoms = len(row_ptr) - 1
    S = np.zeros((n_atoms, H.shape[1]), dtype=H.dtype)
    deg = np.diff(row_ptr)
    for r in range(int(deg.max()) if n_atoms and len(perm) else 0):
        rows = np.flatnonzero(deg > r)
        e = perm[row_ptr[rows] + r]
        if r == 0:
            S[rows] = H[e]  # include_self=False: the first addend is copied, not added to 0
        else:
            S[rows] = S[rows] + H[e]
    return S


def message_edge_form(H, src, rev, row_ptr, perm):
    """mixins.py:11-18 literally: M[e] = S[src(e)] - H[rev(e)]."""
    S = segment_sum_csr(H, row_ptr, perm)
    return S[src] - H[rev]


def message_atom_form(H, rev, row_ptr, perm):
    """Atom-centric form used by the HIP kernel (valid graphs): M[rev(e')] = S[dst(e')] - H[e']."""
    S = segment_sum_csr(H, row_ptr, perm)
    dst_of = np.repeat(np.arange(len(row_ptr) - 1), np.diff(row_ptr))  # dst of perm[k]
    M = np.empty_like(H)
    M[rev[perm]] = S[dst_of] - H[perm]
    return M


def _act(name, x, slope=None):
    name = str(name).lower()
    if name == "relu":
        return np.maximum(x, 0)
    if name == "leakyrelu":
        return np.where(x > 0, x, np.float32(0.1) * x)
    if name == "prelu":
        return np.where(x > 0, x, np.float32(0.25 if slope is None else slope) * x)
    if name == "tanh":
        return np.tanh(x)
    if name == "elu":
        return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))
    if name in ("identity", "none"):
        return x
    raise ValueError(name)


def _act_grad(name, z, y, slope=None):
    """d tau / dz given pre-activation z and output y."""
    name = str(name).lower()
    one = np.ones_like(z)
    if name == "relu":
        return (z > 0).astype(z.dtype)
    if name == "leakyrelu":
        return np.where(z > 0, one, np.float32(0.1) * one)
    if name == "prelu":
        return np.where(z > 0, one, np.float32(0.25 if slope is None else slope) * one)
    if name == "tanh":
        return 1 - y * y
    if name == "elu":
        return np.where(z > 0, one, y + 1)
    if name in ("identity", "none"):
        return one
    raise ValueError(name)


def forward(V, E, edge_index, rev, W, depth=3, activation="relu", undirected=False, V_d=None,
            atom_form=True, keep=False):
    """``W`` is a dict with W_i, W_h, W_o, b_o and optional b_i, b_h, W_d, b_d (``[out,in]``)."""
    V = np.asarray(V, np.float32)
    E = np.asarray(E, np.float32)
    src = np.asarray(edge_index[0], np.int64)
    dst = np.asarray(edge_index[1], np.int64)
    rev = np.asarray(rev, np.int64)
    nV = V.shape[0]
    row_ptr, perm = build_csr(dst, nV)
    use_atom = atom_form and graph_is_symmetric(src, dst, rev)
    f32 = lambda k: None if W.get(k) is None else np.asarray(W[k], np.float32)
    W_i, W_h, W_o, b_o, b_i, b_h, W_d, b_d = (f32(k) for k in ("W_i", "W_h", "W_o", "b_o", "b_i", "b_h", "W_d", "b_d"))
    X = np.concatenate([V[src], E], axis=1)
    H0 = X @ W_i.T
    if b_i is not None:
        H0 = H0 + b_i
    H = _act(activation, H0)
    saved = {"X": X, "H0": H0, "Hs": [H], "Ms": [], "Zs": [H0], "Hbar": []}
    for _ in range(1, depth):
        Hb = (H + H[rev]) / np.float32(2) if undirected else H
        M = message_atom_form(Hb, rev, row_ptr, perm) if use_atom else message_edge_form(Hb, src, rev, row_ptr, perm)
        Z = M @ W_h.T
        if b_h is not None:
            Z = Z + b_h
        Z = H0 + Z
        H = _act(activation, Z)
        saved["Ms"].append(M); saved["Zs"].append(Z); saved["Hs"].append(H); saved["Hbar"].append(Hb)
    Mv = segment_sum_csr(H, row_ptr, perm)
    XO = np.concatenate([V, Mv], axis=1)
    ZO = XO @ W_o.T + b_o
    HO = _act(activation, ZO)
    out = HO
    if V_d is not None:
        XD = np.concatenate([HO, np.asarray(V_d, np.float32)], axis=1)
        out = XD @ W_d.T + b_d
        saved["XD"] = XD
    if keep:
        saved.update(Mv=Mv, XO=XO, ZO=ZO, HO=HO, row_ptr=row_ptr, perm=perm, src=src, dst=dst, rev=rev)
        return out, saved
    return out


def backward(gout, saved, W, depth=3, activation="relu", undirected=False, has_Vd=False):
    """Analytic gradient of ``sum(out * gout)`` w.r.t. the parameters (float64 accumulation is NOT
    used: this mirrors the fp32 kernels).  Derivation (SURVEY §7 'Backward of message'):

      S[v] = sum_{dst(e)=v} H[e],  M[e] = S[src(e)] - H[rev(e)]
      => gH[e'] = gS[dst(e')] - gM[rev(e')],   gS[v] = sum_{src(e)=v} gM[e]
    """
    g = {}
    src, dst, rev = saved["src"], saved["dst"], saved["rev"]
    nV = len(saved["row_ptr"]) - 1
    h = W["W_h"].shape[0]
    W_o = np.asarray(W["W_o"], np.float32)
    W_h = np.asarray(W["W_h"], np.float32)
    gout = np.asarray(gout, np.float32)
    if has_Vd:
        W_d = np.asarray(W["W_d"], np.float32)
        g["W_d"] = gout.T @ saved["XD"]
        g["b_d"] = gout.sum(0)
        gHO = (gout @ W_d)[:, :h]
    else:
        gHO = gout
    gZO = gHO * _act_grad(activation, saved["ZO"], saved["HO"])
    g["W_o"] = gZO.T @ saved["XO"]
    g["b_o"] = gZO.sum(0)
    d_v = saved["XO"].shape[1] - h
    gMv = (gZO @ W_o)[:, d_v:]
    gH = gMv[dst]  # final aggregation: every edge receives its destination atom's gradient
    gH0 = np.zeros_like(saved["H0"])
    g["W_h"] = np.zeros_like(W_h)
    if W.get("b_h") is not None:
        g["b_h"] = np.zeros(h, np.float32)
    for t in range(depth - 1, 0, -1):
        gZ = gH * _act_grad(activation, saved["Zs"][t], saved["Hs"][t])
        g["W_h"] += gZ.T @ saved["Ms"][t - 1]
        if "b_h" in g:
            g["b_h"] += gZ.sum(0)
        gH0 += gZ
        gM = gZ @ W_h
        gS = np.zeros((nV, h), np.float32)
        np.add.at(gS, src, gM)
        gHb = gS[dst] - gM[rev]
        gH = (gHb + gHb[rev]) / np.float32(2) if undirected else gHb
    gH0 += gH * _act_grad(activation, saved["Zs"][0], saved["Hs"][0])
    g["W_i"] = gH0.T @ saved["X"]
    if W.get("b_i") is not None:
        g["b_i"] = gH0.sum(0)
    return g
