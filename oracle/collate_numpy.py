"""TEST INFRASTRUCTURE — CPU restatement of the reference's batching, ``BatchMolGraph.__post_init__``
(chemprop/data/collate.py:37-62), in numpy.  Only tests/ may import it.

Pinned: checked bit-for-bit against the index tensors the EXECUTED reference ``BatchMolGraph`` produced for the same
molecule lists (frozen in tests/golden/mab/*.npz by tests/golden/make_golden_mab.py; tests/test_collate.py).
"""
from __future__ import annotations

import numpy as np


def collate(mgs):
    """-> dict(V f32 [nV,d_v], E f32 [nE,d_e], edge_index i64 [2,nE], rev_edge_index i64 [nE], batch i64 [nV])."""
    Vs, Es, eis, revs, bs = [], [], [], [], []
    num_nodes = num_edges = 0
    for i, mg in enumerate(mgs):                      # collate.py:48-57
        Vs.append(mg.V)
        Es.append(mg.E)
        eis.append(mg.edge_index + num_nodes)
        revs.append(mg.rev_edge_index + num_edges)
        bs.append(np.full(len(mg.V), i, dtype=np.int64))
        num_nodes += mg.V.shape[0]
        num_edges += mg.edge_index.shape[1]
    return dict(V=np.concatenate(Vs).astype(np.float32), E=np.concatenate(Es).astype(np.float32),   # :58-62
                edge_index=np.hstack(eis).astype(np.int64), rev_edge_index=np.concatenate(revs).astype(np.int64),
                batch=np.concatenate(bs).astype(np.int64))


def unpack_wire(raw: np.ndarray):
    """Decode the one-buffer wire format of chemprop_amd/data.py (PackedBatch) on the host and batch it the reference's
    way: what ``dmpnn_collate`` must produce from the same bytes."""
    hdr = raw[:64].view(np.int64)
    magic, n_mols, nV, nE, d_v, d_e, n_tiles, nEr = (int(x) for x in hdr[:8])
    nt = n_tiles + 1 if n_tiles >= 0 else 0
    assert magic == 0x31424D44
    a16 = lambda n: (n + 15) // 16 * 16
    o = 64
    out = {}
    for name, dt, count in (("atom_off", np.int32, n_mols + 1), ("edge_off", np.int32, n_mols + 1), ("src", np.int32, nE),
                            ("dst", np.int32, nE), ("rev", np.int32, nE), ("V", np.float32, nV * d_v), ("E", np.float32, nEr * d_e),
                            ("tile_row", np.int32, nt), ("tile_atom", np.int32, nt)):
        nbytes = count * np.dtype(dt).itemsize
        out[name] = raw[o:o + nbytes].view(dt)
        o = a16(o + nbytes)
    assert o == raw.size
    ao, eo = out["atom_off"].astype(np.int64), out["edge_off"].astype(np.int64)
    m_of_edge = np.searchsorted(eo, np.arange(nE), side="right") - 1
    m_of_atom = np.searchsorted(ao, np.arange(nV), side="right") - 1
    return dict(tile_row=out["tile_row"], tile_atom=out["tile_atom"], n_tiles=n_tiles,
                V=out["V"].reshape(nV, d_v), E=out["E"].reshape(nEr, d_e),
                edge_index=np.stack([out["src"] + ao[m_of_edge], out["dst"] + ao[m_of_edge]]).astype(np.int64),
                rev_edge_index=(out["rev"] + eo[m_of_edge]).astype(np.int64), batch=m_of_atom.astype(np.int64))


def greedy_molecule_tiles(n_atoms, n_edges, max_rows: int = 48, max_atoms: int = 32):
    """The loader-side tile table (``dmpnn_pack_tiles``): consecutive whole molecules packed greedily into tiles of at most
    ``max_rows`` directed edges and ``max_atoms`` atoms (the tile limits of the whole-forward tile kernel).
    -> (tile_row [n+1], tile_atom [n+1]); a molecule that alone exceeds a tile is a tile of its own.  Molecules without atoms and
    bonds belong to no tile."""
    rows, atoms = [], []
    r = a = 0          # running offsets
    tr = ta = 0        # start of the open tile
    open_tile = False
    for na, ne in zip(n_atoms, n_edges):
        na, ne = int(na), int(ne)
        if open_tile and (a + na - ta > max_atoms or r + ne - tr > max_rows):
            open_tile = False
        if not open_tile and (na or ne):
            rows.append(r); atoms.append(a)
            tr, ta, open_tile = r, a, True
        r += ne; a += na
        if na > max_atoms or ne > max_rows:
            open_tile = False  # an oversize molecule is a tile of its own (the tile kernel's generic path)
    rows.append(r); atoms.append(a)
    return np.array(rows, dtype=np.int32), np.array(atoms, dtype=np.int32)


def blocked_molecule_tiles(n_atoms, n_edges, block: int = 64, max_rows: int = 48, max_atoms: int = 32):
    """The DEVICE tile planners' packing (k_prepare_tiles_batch for small batches, dmpnn_tiles_large.hip beyond): greedy as
    above, except that a tile also starts at every block of ``block`` consecutive molecules (the blocks are walked in
    parallel).  -> (tile_row [n+1], tile_atom [n+1]); a molecule that alone exceeds a tile is a tile of its own."""
    n_atoms, n_edges = [int(x) for x in n_atoms], [int(x) for x in n_edges]
    n = len(n_atoms)
    ao = np.concatenate([[0], np.cumsum(n_atoms)]).astype(np.int64)
    eo = np.concatenate([[0], np.cumsum(n_edges)]).astype(np.int64)
    rows, atoms = [], []
    for base in range(0, n, block):
        lim = min(base + block, n)
        p = base
        while p < lim:
            rows.append(int(eo[p])); atoms.append(int(ao[p]))
            q = p + 1
            if n_atoms[p] <= max_atoms and n_edges[p] <= max_rows:  # (an oversize molecule is a tile of its own)
                while q < n and ao[q + 1] - ao[p] <= max_atoms and eo[q + 1] - eo[p] <= max_rows:
                    q += 1
            p = q
    rows.append(int(eo[n])); atoms.append(int(ao[n]))
    return np.array(rows, dtype=np.int32), np.array(atoms, dtype=np.int32)


def full_plan_tiles_ok(src, dst, n_atoms: int, tile_row, tile_atom) -> bool:
    """What the full plan WITH molecule tiles (``dmpnn_prepare_with_batch``; csrc/dmpnn_prepare.hip: k_rows_tiles, keep_mtiles)
    verifies on the device before the tile kernels may read it by rows: every table entry is the row offset of its atom in
    the incoming-edge CSR (the caller-order edge range of a tile IS its row range: edges in molecule order,
    ``chemprop/data/collate.py:51-56``), and every edge's source atom lies in the tile of its destination atom (no bond
    between two tiles).  ``tile_row`` / ``tile_atom``: [n_tiles + 1] as returned by :func:`blocked_molecule_tiles`."""
    src, dst = np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)
    tile_row, tile_atom = np.asarray(tile_row, dtype=np.int64), np.asarray(tile_atom, dtype=np.int64)
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n_atoms))]).astype(np.int64)
    if (tile_atom < 0).any() or (tile_atom > n_atoms).any() or not np.array_equal(tile_row, row_ptr[tile_atom]):
        return False
    t_dst = np.searchsorted(tile_atom, dst, side="right") - 1   # last tile whose first atom is <= dst
    t_dst = np.clip(t_dst, 0, len(tile_atom) - 2)
    return bool(((src >= tile_atom[t_dst]) & (src < tile_atom[t_dst + 1])).all())

