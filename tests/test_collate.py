"""f3 — batching (chemprop/data/collate.py:37-62): the one-buffer wire format of chemprop_amd/data.py and the device
kernel ``dmpnn_collate`` against the oracle restatement and against the index tensors the EXECUTED reference
``BatchMolGraph`` produced (frozen in tests/golden/mab/*.npz, whose meta records the molecule lists).  Integer work:
bit-exact."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

MAB = sorted(glob.glob(os.path.join(GOLDEN_DIR, "mab", "*.npz")))
KEYS = ("V", "E", "edge_index", "rev_edge_index", "batch")


def _golden(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    return {k: z[k] for k in KEYS}, meta


def _molgraphs(meta):
    from chemprop_amd import synth

    n, kind, seed = meta["graphs"]
    return synth.random_molgraphs(n, kind, seed=seed)


def _same(got: dict, want: dict):
    for k in KEYS:
        g, w = np.asarray(got[k]), np.asarray(want[k])
        assert g.dtype == w.dtype and g.shape == w.shape, (k, g.dtype, w.dtype, g.shape, w.shape)
        assert np.array_equal(g, w), k


@pytest.mark.parametrize("path", MAB, ids=[os.path.basename(p)[:-4] for p in MAB])
def test_oracle_and_wire_format_vs_executed_reference(path):
    from chemprop_amd.data import BatchMolGraph, PackedBatch
    from oracle import collate_numpy as oc

    want, meta = _golden(path)
    mgs = _molgraphs(meta)
    _same(oc.collate(mgs), want)                                     # the restatement is pinned
    pb = PackedBatch(mgs)
    assert (pb.n_mols, pb.n_atoms, pb.n_edges) == (meta["n_mols"], want["V"].shape[0], want["E"].shape[0])
    _same(oc.unpack_wire(pb.buf.numpy()), want)                      # what dmpnn_collate must make of these bytes
    host = BatchMolGraph(mgs)                                        # the host mirror batches the same way
    _same({k: getattr(host, k).numpy() for k in KEYS}, want)


def _odd_molgraphs():
    """Edge cases of the batching: a lone atom (no bonds), an empty molecule, a two-atom molecule, a ring."""
    from chemprop_amd.data import MolGraph

    f = lambda n, d: np.arange(n * d, dtype=np.float32).reshape(n, d) / 7
    e = lambda pairs: (np.array([[a for a, b in pairs] + [b for a, b in pairs], [b for a, b in pairs] + [a for a, b in pairs]], dtype=np.int64),
                       np.concatenate([np.arange(len(pairs)) + len(pairs), np.arange(len(pairs))]).astype(np.int64))
    out = []
    for n, pairs in ((1, []), (0, []), (2, [(0, 1)]), (5, [(0, 1), (1, 2), (2, 3), (3, 4), (4, 0)]), (1, []), (3, [(0, 2), (2, 1)])):
        ei, rev = e(pairs) if pairs else (np.zeros((2, 0), np.int64), np.zeros(0, np.int64))
        out.append(MolGraph(V=f(n, 4), E=f(2 * len(pairs), 3), edge_index=ei, rev_edge_index=rev))
    return out


def test_wire_format_edge_cases():
    from chemprop_amd.data import PackedBatch
    from oracle import collate_numpy as oc

    mgs = _odd_molgraphs()
    pb = PackedBatch(mgs)
    _same(oc.unpack_wire(pb.buf.numpy()), oc.collate(mgs))
    assert len(pb) == 6 and pb.buf.numel() % 16 == 0
    empty = PackedBatch([])
    assert (empty.n_mols, empty.n_atoms, empty.n_edges) == (0, 0, 0)
    with pytest.raises(RuntimeError):  # batching is a HIP kernel: no host fallback behind to_device
        pb.to_device("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("path", MAB[:4], ids=[os.path.basename(p)[:-4] for p in MAB[:4]])
def test_device_collate_vs_executed_reference(path, gpu_device):
    from chemprop_amd.data import PackedBatch

    want, meta = _golden(path)
    bmg = PackedBatch(_molgraphs(meta), pin=True).to_device(gpu_device)
    assert len(bmg) == meta["n_mols"]
    _same({k: getattr(bmg, k).cpu().numpy() for k in KEYS}, want)


@pytest.mark.gpu
def test_device_collate_edge_cases(gpu_device):
    from chemprop_amd.data import PackedBatch
    from oracle import collate_numpy as oc

    mgs = _odd_molgraphs()
    bmg = PackedBatch(mgs).to_device(gpu_device)
    _same({k: getattr(bmg, k).cpu().numpy() for k in KEYS}, oc.collate(mgs))
    e = PackedBatch([]).to_device(gpu_device)
    assert e.V.shape[0] == 0 and e.edge_index.shape == (2, 0) and e.batch.numel() == 0 and len(e) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_mols,kind", [(512, "qm9"), (4096, "zinc")])
def test_device_collate_full_size_and_forward(n_mols, kind, gpu_device):
    """BASELINE sizes: bit-exact against the oracle, and the block fed from the packed batch gives the same bits as
    the block fed from the host-built batch."""
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph, PackedBatch
    from chemprop_amd.nn import BondMessagePassing
    from oracle import collate_numpy as oc

    mgs = synth.random_molgraphs(n_mols, kind, seed=11)
    bmg = PackedBatch(mgs, pin=True).to_device(gpu_device)
    _same({k: getattr(bmg, k).cpu().numpy() for k in KEYS}, oc.collate(mgs))
    host = BatchMolGraph(mgs)
    host.to(gpu_device)
    torch.manual_seed(0)
    mp = BondMessagePassing().eval().to(gpu_device)
    with torch.no_grad():
        assert torch.equal(mp(bmg), mp(host))


# ---- the loader-side tile table (dmpnn_pack_tiles, a HOST function of the library: runs without a GPU) ----
def _pack_tiles(n_atoms, n_edges):
    import ctypes as C

    from chemprop_amd import _lib

    lib = _lib.load()
    ao = np.zeros(len(n_atoms) + 1, np.int32); ao[1:] = np.cumsum(n_atoms)
    eo = np.zeros(len(n_edges) + 1, np.int32); eo[1:] = np.cumsum(n_edges)
    cap = int(lib.dmpnn_max_tiles(int(ao[-1]), int(eo[-1]))) + 1
    tr, ta = np.empty(cap, np.int32), np.empty(cap, np.int32)
    n = int(lib.dmpnn_pack_tiles(ao.ctypes.data, eo.ctypes.data, len(n_atoms), tr.ctypes.data, ta.ctypes.data, cap))
    return n, tr[:max(n, 0) + 1], ta[:max(n, 0) + 1]


@pytest.mark.parametrize("kind,n_mols,seed", [("qm9", 1, 0), ("qm9", 64, 1), ("qm9", 512, 2), ("qm9", 4096, 3), ("zinc", 40, 4)])
def test_pack_tiles_vs_oracle(kind, n_mols, seed):
    from chemprop_amd import synth
    from oracle import collate_numpy as oc

    mgs = synth.random_molgraphs(n_mols, kind, seed=seed)
    n_at = [len(m.V) for m in mgs]
    n_ed = [m.edge_index.shape[1] for m in mgs]
    want = oc.greedy_molecule_tiles(n_at, n_ed)
    n, tr, ta = _pack_tiles(n_at, n_ed)
    assert n == len(want[0]) - 1 and np.array_equal(tr, want[0]) and np.array_equal(ta, want[1])
    # what the tile kernel needs: whole molecules, a partition of the batch; <= 48 rows and <= 32 atoms per tile — or ONE
    # molecule beyond that (a tile of its own for the kernel's generic path)
    assert tr[0] == 0 and ta[0] == 0 and tr[-1] == sum(n_ed) and ta[-1] == sum(n_at)
    starts = np.concatenate([[0], np.cumsum(n_at)])
    assert set(ta.tolist()) <= set(starts.tolist()) and (np.diff(ta) > 0).all()
    for t in range(n):
        if tr[t + 1] - tr[t] > 48 or ta[t + 1] - ta[t] > 32:
            m = int(np.searchsorted(starts, ta[t]))
            assert starts[m] == ta[t] and starts[m + 1] == ta[t + 1], "an oversize tile holds exactly one molecule"
    if kind == "zinc":
        assert any(tr[t + 1] - tr[t] > 48 for t in range(n))


def test_pack_tiles_edge_cases():
    from oracle import collate_numpy as oc

    for n_at, n_ed in (([1, 0, 2, 5, 1, 3], [0, 0, 2, 10, 0, 4]), ([0, 0], [0, 0]), ([32], [48]), ([16, 16, 1], [24, 24, 0]), ([3], [0])):
        want = oc.greedy_molecule_tiles(n_at, n_ed)
        n, tr, ta = _pack_tiles(n_at, n_ed)
        assert n == len(want[0]) - 1 and np.array_equal(tr, want[0]) and np.array_equal(ta, want[1]), (n_at, n_ed)
    # a molecule larger than a tile: a tile of its own (the tile kernel's generic path), never a refusal
    for n_at, n_ed in (([33], [10]), ([4, 20], [6, 50]), ([4, 40, 3, 3], [6, 86, 4, 4]), ([40, 40], [86, 90])):
        want = oc.greedy_molecule_tiles(n_at, n_ed)
        n, tr, ta = _pack_tiles(n_at, n_ed)
        assert n == len(want[0]) - 1 and np.array_equal(tr, want[0]) and np.array_equal(ta, want[1]), (n_at, n_ed)
    assert _pack_tiles([4, 40, 3, 3], [6, 86, 4, 4])[1].tolist() == [0, 6, 92, 100]


def test_packed_batch_carries_the_tile_table():
    from chemprop_amd import synth
    from chemprop_amd.data import PackedBatch
    from oracle import collate_numpy as oc

    mgs = synth.random_molgraphs(100, "qm9", seed=8)
    pb = PackedBatch(mgs)
    w = oc.unpack_wire(pb.buf.numpy())
    want = oc.greedy_molecule_tiles([len(m.V) for m in mgs], [m.edge_index.shape[1] for m in mgs])
    assert pb.n_tiles == w["n_tiles"] == len(want[0]) - 1
    assert np.array_equal(w["tile_row"], want[0]) and np.array_equal(w["tile_atom"], want[1])
    assert PackedBatch(mgs, tiles=False).n_tiles == -1
    assert pb.oversize is False
    big = synth.random_molgraphs(6, "synth40", seed=1)  # 40-atom molecules: the host knows, no table, per-step routes
    assert PackedBatch(big).n_tiles == -1 and PackedBatch(big).oversize is True


@pytest.mark.gpu
@pytest.mark.parametrize("n_mols", [512, 4096])
def test_forward_on_loader_tiles(n_mols, gpu_device):
    """The tile plan copied from the loader's table gives the same function as the device-built plans (the tile
    composition differs, so the per-tile f16 scales do: fp32-class agreement, not bits), at any batch size; and it is
    the whole-forward tile kernel that runs."""
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph, PackedBatch
    from chemprop_amd.nn import BondMessagePassing
    from conftest import TOL, parity_err
    from oracle import dmpnn_torch as ot

    mgs = synth.random_molgraphs(n_mols, "qm9", seed=13)
    host = BatchMolGraph(mgs)
    torch.manual_seed(1)
    mp = BondMessagePassing().eval()
    with torch.no_grad():
        ref = ot.forward_bmg(host, ot.MPWeights.from_module(mp), depth=mp.depth).numpy()
    mp = mp.to(gpu_device)
    packed = PackedBatch(mgs, pin=True)
    assert packed.n_tiles > 0
    with torch.no_grad():
        for i in range(4):  # the first batches of a module are validated on full plans; then tile plans, then the replay path
            bmg = packed.to_device(gpu_device)
            out = mp(bmg)
            assert parity_err(out.cpu().numpy(), ref) <= TOL, i
        assert mp.__dict__.get("_dmpnn_replay") is not None
        from chemprop_amd import engine

        plan = engine.GraphPlan.from_bmg(bmg, light="tiles")
        assert plan.loader_tiles == packed.n_tiles and plan.tiles_only
        hdr = plan.arrays()["hdr"]
        assert int(hdr[7]) == 2 and int(hdr[6]) == packed.n_tiles and int(hdr[0]) == 16


@pytest.mark.gpu
def test_a_wrong_loader_table_is_loud(gpu_device):
    """Tiles that split a molecule (not closed) poison the affected output with NaN; a tile beyond the matrix-pipe limits
    that is closed is computed by the generic path."""
    from chemprop_amd import engine, synth
    from chemprop_amd.data import PackedBatch
    from chemprop_amd.nn import BondMessagePassing

    bmg = PackedBatch(synth.random_molgraphs(64, "qm9", seed=2), pin=True).to_device(gpu_device)
    torch.manual_seed(1)
    mp = BondMessagePassing(d_h=64).eval().to(gpu_device)
    p = {k: v for k, v in mp.named_parameters()}
    fwd = lambda plan: engine.forward(plan, bmg.V, bmg.E, p["W_i.weight"], p["W_h.weight"], p["W_o.weight"], p["W_o.bias"],
                                      depth=3, act="relu", route="mega")[0]
    tr, ta, n = bmg.tiles
    with torch.no_grad():
        good = fwd(engine.GraphPlan.from_bmg(bmg, light="tiles"))
        assert torch.isfinite(good).all()
        tr2, ta2 = tr.clone(), ta.clone()
        tr2[1] += 2  # tile 0 now takes two edge rows of the next molecule: neither tile is closed
        bad = fwd(engine.GraphPlan(bmg.edge_index, bmg.rev_edge_index, bmg.V.shape[0], light="tiles", batch=bmg.batch, tiles=(tr2, ta2, n)))
        a1, a2 = int(ta[1]), int(ta[2])
        assert torch.isnan(bad[:a2]).any() and torch.equal(bad[a2:], good[a2:]) and a1 < a2
        tr3, ta3 = tr.clone(), ta.clone()
        tr3[1:n] = tr[n]; ta3[1:n] = ta[n]  # one tile of everything: beyond the matrix-pipe tile, but CLOSED — the kernel's
        # generic path computes it (a table can only cost speed, or be loud: never a wrong number)
        whole = fwd(engine.GraphPlan(bmg.edge_index, bmg.rev_edge_index, bmg.V.shape[0], light="tiles", batch=bmg.batch, tiles=(tr3, ta3, n)))
        assert float((whole - good).abs().max()) <= 3e-6 * max(1.0, float(good.abs().max()))
        tr4 = tr3.clone()
        tr4[1:n] = tr[n] - 2  # ... the same tile without the last two edge rows: not closed -> NaN for its atoms (all of them)
        assert torch.isnan(fwd(engine.GraphPlan(bmg.edge_index, bmg.rev_edge_index, bmg.V.shape[0], light="tiles", batch=bmg.batch,
                                                tiles=(tr4, ta3, n)))).all()


# ---- the multi-workgroup tile planner: batches beyond the single-workgroup plan, handed over as five tensors ----
@pytest.mark.gpu
@pytest.mark.parametrize("n_mols,seed", [(2000, 5), (4096, 6), (32768, 7)])
def test_large_tile_plan_vs_oracle(n_mols, seed, gpu_device):
    """dmpnn_prepare_tiles with the batch vector on a batch the single-workgroup plan does not take: the blocked greedy
    tables, bit for bit (integer work), as a tile plan."""
    from chemprop_amd import engine, synth
    from chemprop_amd.data import BatchMolGraph
    from oracle import collate_numpy as oc

    mgs = synth.random_molgraphs(n_mols, "qm9", seed=seed)
    bmg = BatchMolGraph(mgs)
    bmg.to(gpu_device)
    nV, nE = int(bmg.V.shape[0]), int(bmg.E.shape[0])
    assert not engine.small_plan_fits(nV, nE)
    plan = engine.GraphPlan.from_bmg(bmg, light="tiles")
    assert plan.tiles_only and plan.any_size and plan.loader_tiles == 0
    arr = plan.arrays()
    want = oc.blocked_molecule_tiles([len(m.V) for m in mgs], [m.edge_index.shape[1] for m in mgs])
    n = len(want[0]) - 1
    hdr = arr["hdr"]
    assert int(hdr[0]) == 16 and int(hdr[7]) == 2 and int(hdr[6]) == n and int(hdr[2]) == nV and int(hdr[3]) == nE
    assert np.array_equal(arr["mtile_row"][:n + 1].numpy(), want[0]) and np.array_equal(arr["mtile_atom"][:n + 1].numpy(), want[1])
    assert (arr["mtile_row"][n:] == nE).all() and (arr["mtile_atom"][n:] == nV).all()


@pytest.mark.gpu
def test_large_batch_of_five_tensors_takes_the_tile_kernel(gpu_device):
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph
    from chemprop_amd.nn import BondMessagePassing
    from conftest import TOL, parity_err
    from oracle import dmpnn_torch as ot

    mgs = synth.random_molgraphs(4096, "qm9", seed=17)
    host = BatchMolGraph(mgs)
    torch.manual_seed(2)
    mp = BondMessagePassing().eval()
    with torch.no_grad():
        ref = ot.forward_bmg(host, ot.MPWeights.from_module(mp), depth=mp.depth).numpy()
    mp = mp.to(gpu_device)
    host.to(gpu_device)
    with torch.no_grad():
        for i in range(4):  # validated batches (full plan, per-step route), then the tile plan, then the replay path
            assert parity_err(mp(host).cpu().numpy(), ref) <= TOL, i
    assert mp.__dict__.get("_dmpnn_replay") is not None and not getattr(mp, "_dmpnn_no_mega", False)


@pytest.mark.gpu
def test_large_tile_plan_carries_an_oversize_molecule(gpu_device):
    """One 40-atom molecule among 3000 small ones, handed over as bare tensors (multi-workgroup tile planner): no error flag,
    one spill tile, the tile route computes it; a module that meets it in its validated batches, and the same batch from the
    host-side batching code (which knows), take the per-step route.  All within the bar."""
    from chemprop_amd import engine, synth
    from chemprop_amd.data import BatchMolGraph
    from chemprop_amd.nn import BondMessagePassing
    from conftest import TOL, parity_err
    from oracle import dmpnn_torch as ot

    mgs = synth.random_molgraphs(3000, "qm9", seed=3)
    mgs[1234] = synth.random_molgraphs(1, "synth40", seed=9)[0]
    host = BatchMolGraph(mgs)
    assert host.oversize is True
    torch.manual_seed(2)
    mp = BondMessagePassing().eval()
    with torch.no_grad():
        ref = ot.forward_bmg(host, ot.MPWeights.from_module(mp), depth=mp.depth).numpy()
    host.to(gpu_device)
    bare = BatchMolGraph.from_tensors(host.V, host.E, host.edge_index, host.rev_edge_index, host.batch, len(host))
    plan = engine.GraphPlan.from_bmg(bare, light="tiles")
    hdr = plan.header()
    assert plan.tiles_only and hdr[0] & 15 == 0 and hdr[8] == 1
    mp = mp.to(gpu_device)
    with torch.no_grad():
        p = mp.state_dict()
        out, st = engine.forward(plan, bare.V, bare.E, p["W_i.weight"], p["W_h.weight"], p["W_o.weight"], p["W_o.bias"], depth=mp.depth)
        assert st.route == "mega16" and parity_err(out.cpu().numpy(), ref) <= TOL   # (the tile route, oversize molecule included)
        # a module: its validated first batches are full plans WITH the molecule tiles (dmpnn_prepare_with_batch), so the
        # oversize molecule is seen there and the module takes the per-step routes, as with small batches (the generic path
        # of one such molecule takes 1.7 ms whatever the batch around it: scripts/probe_spill_policy.py)
        for i in range(4):
            assert parity_err(mp(bare).cpu().numpy(), ref) <= TOL, i
        assert getattr(mp, "_dmpnn_no_mega", False) and mp.__dict__.get("_dmpnn_replay") is None
        mp2 = BondMessagePassing().eval().to(gpu_device)
        mp2.load_state_dict(mp.state_dict())
        for i in range(3):
            assert parity_err(mp2(host).cpu().numpy(), ref) <= TOL, i
        assert mp2.__dict__.get("_dmpnn_replay") is None


def test_oracle_statement_of_the_full_plan_tile_check():
    """The device check of ``dmpnn_prepare_with_batch`` restated (oracle/collate_numpy.py: full_plan_tiles_ok): molecule
    tiles of a collated batch pass; a bond between two tiles, or edges out of molecule order, do not."""
    from chemprop_amd import synth
    from oracle import collate_numpy as oc

    mgs = synth.random_molgraphs(700, "qm9", seed=8)
    b = oc.collate(mgs)
    src, dst = b["edge_index"][0], b["edge_index"][1]
    nV = len(b["batch"])
    tr, ta = oc.blocked_molecule_tiles([len(m.V) for m in mgs], [m.edge_index.shape[1] for m in mgs])
    assert oc.full_plan_tiles_ok(src, dst, nV, tr, ta)
    # a bond between an atom of the first tile and one of the last
    s2, d2 = src.copy(), dst.copy()
    e = int(len(src) // 2)
    r = int(b["rev_edge_index"][e])
    s2[e] = ta[-2]
    d2[r] = ta[-2]
    assert not oc.full_plan_tiles_ok(s2, d2, nV, tr, ta)
    # a table whose row offsets are not the CSR offsets of its atoms (what a planner working on edges that are NOT in molecule
    # order would produce): rejected; the CSR itself does not depend on the order of the edges
    tr2 = tr.copy()
    tr2[len(tr2) // 2] += 2
    assert not oc.full_plan_tiles_ok(src, dst, nV, tr2, ta)
    perm = np.arange(len(src)).reshape(-1, 2)[::-1].reshape(-1)
    assert oc.full_plan_tiles_ok(src[perm], dst[perm], nV, tr, ta)


# ---- properties of the host-side packers over arbitrary molecule-size sequences (hypothesis; no GPU) ----
def test_pack_tiles_properties():
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from oracle import collate_numpy as oc

    sizes = st.lists(st.tuples(st.integers(0, 36), st.integers(0, 56)), min_size=0, max_size=300)

    @settings(max_examples=300, deadline=None)
    @given(sizes)
    def prop(mols):
        n_at = [a for a, _ in mols]
        n_ed = [2 * (e // 2) if a >= 2 else 0 for a, e in mols]  # directed edges come in pairs; none without two atoms
        want = oc.greedy_molecule_tiles(n_at, n_ed)
        n, tr, ta = _pack_tiles(n_at, n_ed)
        assert n == len(want[0]) - 1 and np.array_equal(tr, want[0]) and np.array_equal(ta, want[1])
        # a partition of the batch into consecutive whole molecules within the tile limits, and greedy: no two consecutive
        # tiles could have been one
        ao = np.concatenate([[0], np.cumsum(n_at)]); eo = np.concatenate([[0], np.cumsum(n_ed)])
        assert tr[-1] == eo[-1] and ta[-1] == ao[-1]
        if n:
            assert (np.diff(tr) >= 0).all() and (np.diff(ta) >= 0).all()
            for t in range(n):  # within the tile limits, or ONE molecule beyond them
                if tr[t + 1] - tr[t] > 48 or ta[t + 1] - ta[t] > 32:
                    inside = [m for m in range(len(n_at)) if ta[t] <= ao[m] < ta[t + 1] and (n_at[m] or n_ed[m])]
                    assert len(inside) == 1 and (n_at[inside[0]] > 32 or n_ed[inside[0]] > 48)
            for t in range(n - 1):
                assert tr[t + 2] - tr[t] > 48 or ta[t + 2] - ta[t] > 32 or (tr[t + 1] == tr[t] and ta[t + 1] == ta[t])
            bounds = set(zip(ao.tolist(), eo.tolist()))
            assert all((int(a), int(r)) in bounds for a, r in zip(ta, tr))
        blocked = oc.blocked_molecule_tiles(n_at, n_ed)
        # (one under-filled tile per block of 64 molecules; a molecule WITHOUT atoms — the reference never makes one: an empty molecule
        #  gets a phantom atom, molecule.py:65-66 — next to an oversize one may stand as an empty tile of its own, which the kernels skip)
        n_empty = sum(1 for a_, e_ in zip(n_at, n_ed) if a_ == 0)
        assert n <= len(blocked[0]) - 1 <= n + (len(n_at) + 63) // 64 + 1 + n_empty

    prop()


def test_packed_batch_roundtrip_properties():
    """Random molecule lists (including empty molecules and lone atoms): wire bytes -> oracle decode == reference batching."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from chemprop_amd.data import MolGraph, PackedBatch
    from oracle import collate_numpy as oc

    @settings(max_examples=60, deadline=None)
    @given(st.lists(st.integers(0, 12), min_size=1, max_size=40), st.integers(0, 2 ** 31 - 1))
    def prop(n_atoms, seed):
        rng = np.random.default_rng(seed)
        mgs = []
        for n in n_atoms:
            pairs = [(i, int(rng.integers(0, i))) for i in range(1, n)]  # a random tree
            nb = len(pairs)
            src = np.array([a for a, b in pairs] + [b for a, b in pairs], dtype=np.int64)
            dst = np.array([b for a, b in pairs] + [a for a, b in pairs], dtype=np.int64)
            rev = np.concatenate([np.arange(nb) + nb, np.arange(nb)]).astype(np.int64)
            mgs.append(MolGraph(V=rng.random((n, 5), dtype=np.float32), E=rng.random((2 * nb, 3), dtype=np.float32),
                                edge_index=np.stack([src, dst]) if nb else np.zeros((2, 0), np.int64), rev_edge_index=rev))
        pb = PackedBatch(mgs)
        got, want = oc.unpack_wire(pb.buf.numpy()), oc.collate(mgs)
        _same(got, want)
        assert pb.n_tiles == got["n_tiles"]
        if sum(n_atoms):
            tiles = oc.greedy_molecule_tiles(n_atoms, [m.edge_index.shape[1] for m in mgs])
            assert pb.n_tiles == len(tiles[0]) - 1 and np.array_equal(got["tile_row"], tiles[0])

    prop()


# ---- the reference's own batching fixtures (tests/unit/data/test_dataloader.py:10-45: two hand-built molecules with
# 1-dim float64 features; the first carries fewer E rows than directed edges — the reference concatenates as given) ----
def _reference_fixture_molgraphs():
    from chemprop_amd.data import MolGraph

    m1 = MolGraph(V=np.array([[1.0], [2.0], [3.0]]), E=np.array([[0.5], [1.5]]),
                  edge_index=np.array([[0, 1, 0, 2], [1, 0, 2, 0]]), rev_edge_index=np.array([1, 0, 3, 2]))
    m2 = MolGraph(V=np.array([[4.0], [5.0]]), E=np.array([[2.5]]), edge_index=np.array([[0, 1], [1, 0]]),
                  rev_edge_index=np.array([1, 0]))
    want = dict(V=np.array([[1], [2], [3], [4], [5]], np.float32), E=np.array([[0.5], [1.5], [2.5]], np.float32),
                edge_index=np.array([[0, 1, 0, 2, 3, 4], [1, 0, 2, 0, 4, 3]], np.int64),
                rev_edge_index=np.array([1, 0, 3, 2, 5, 4], np.int64), batch=np.array([0, 0, 0, 1, 1], np.int64))
    return [m1, m2], want


def test_reference_dataloader_fixtures_on_the_wire():
    from chemprop_amd.data import BatchMolGraph, PackedBatch
    from oracle import collate_numpy as oc

    mgs, want = _reference_fixture_molgraphs()
    _same(oc.collate(mgs), want)
    host = BatchMolGraph(mgs)
    _same({k: getattr(host, k).numpy() for k in KEYS}, want)
    pb = PackedBatch(mgs)
    assert (pb.n_atoms, pb.n_edges, pb.n_erows, len(pb)) == (5, 6, 3, 2)
    _same(oc.unpack_wire(pb.buf.numpy()), want)
    single = PackedBatch(mgs[:1])
    _same(oc.unpack_wire(single.buf.numpy()), oc.collate(mgs[:1]))


@pytest.mark.gpu
def test_reference_dataloader_fixtures_on_the_device(gpu_device):
    from chemprop_amd.data import PackedBatch

    mgs, want = _reference_fixture_molgraphs()
    bmg = PackedBatch(mgs).to_device(gpu_device)
    _same({k: getattr(bmg, k).cpu().numpy() for k in KEYS}, want)
    assert len(bmg) == 2
