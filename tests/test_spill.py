"""A molecule larger than the tile of the whole-forward tile kernels (> 48 directed edges or > 32 atoms).

The reference has no size limit (``chemprop/data/collate.py:48-56``, ``nn/message_passing/base.py:196-212``).  The
planners hand such a molecule over as a tile of its own and the tile kernels run their generic fp32 path on it
(``csrc/dmpnn_spill_impl.hpp``) — whenever it turns up, in particular AFTER the batches a module validates
synchronously (round-1 VERDICT "What's weak" 1 / ADVICE high: it used to turn the whole batch into NaN).
"""
import numpy as np
import pytest
import torch

from conftest import TOL, parity_err


def _bare(bmg):
    """The same batch as bare tensors (the reference's own BatchMolGraph carries no host-side size knowledge)."""
    from chemprop_amd.data import BatchMolGraph

    return BatchMolGraph.from_tensors(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, bmg.batch, len(bmg))


def _mixed(n_small, big_kinds, seed, where=None):
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph

    mgs = synth.random_molgraphs(n_small, "qm9", seed=seed)
    for i, (kind, s) in enumerate(big_kinds):
        pos = (where[i] if where else (7 * (i + 1)) % max(n_small, 1))
        mgs.insert(pos, synth.random_molgraphs(1, kind, seed=s)[0])
    return mgs, BatchMolGraph(mgs)


# ---------------------------------------------------------------- CPU: host knowledge and the oracle tables
def test_host_side_size_knowledge():
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph, molecules_oversize

    assert BatchMolGraph(synth.random_molgraphs(30, "qm9", seed=0)).oversize is False
    mgs, b = _mixed(30, [("synth40", 1)], seed=0)
    assert b.oversize is True and _bare(b).oversize is None
    import copy

    assert copy.copy(b).oversize is True
    assert molecules_oversize([32], [48]) is False and molecules_oversize([33], [10]) is True and molecules_oversize([], []) is False


def test_oracle_tables_give_an_oversize_molecule_its_own_tile():
    from oracle import collate_numpy as oc
    from oracle import dmpnn_numpy as onp

    n_at, n_ed = [5, 40, 6, 6, 50, 50, 3], [8, 86, 10, 10, 110, 104, 4]
    for f in (oc.greedy_molecule_tiles, oc.blocked_molecule_tiles):
        tr, ta = f(n_at, n_ed)
        assert ta.tolist() == [0, 5, 45, 57, 107, 157, 160] and tr.tolist() == [0, 8, 94, 114, 224, 328, 332]
    # the connectivity form: a 40-atom chain between two small pieces
    def chain(n, a0):
        s = np.arange(n - 1) + a0
        return np.stack([np.concatenate([s, s + 1]), np.concatenate([s + 1, s])])
    ei = np.concatenate([chain(4, 0), chain(40, 4), chain(3, 44)], axis=1)
    order = np.argsort(ei[1], kind="stable")
    row_ptr, perm = onp.build_csr(ei[1], 47)
    mrow, matom, n = onp.piece_tiles(ei[0], ei[1], row_ptr, 20)
    assert n == 3 and matom[:4].tolist() == [0, 4, 44, 47] and mrow[:4].tolist() == [0, 6, 84, 88]


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n_small,depth,act,bias", [(200, 3, "relu", False), (40, 4, "tanh", True), (3000, 3, "leakyrelu", False)])
def test_late_oversize_molecule_inference(n_small, depth, act, bias, gpu_device):
    """Four QM9-shaped batches (two of them validated synchronously), then a batch with a 40-atom and a ZINC-sized molecule:
    finite, and within 1e-5 of the oracle — on the tile route, not by luck of the validation window."""
    from chemprop_amd import engine, synth
    from chemprop_amd.nn import BondMessagePassing
    from oracle import dmpnn_torch as ot

    torch.manual_seed(5)
    mp = BondMessagePassing(depth=depth, activation=act, bias=bias).eval()
    w = ot.MPWeights.from_module(mp)
    slope = 0.1 if act == "leakyrelu" else 0.0
    late_mgs, late = _mixed(n_small, [("synth40", 3), ("zinc", 4), ("synth40", 11)], seed=9)
    with torch.no_grad():
        ref = ot.forward_bmg(late, w, depth=depth, activation=act).numpy()
    mp = mp.to(gpu_device)
    with torch.no_grad():
        for i in range(4):
            b = _bare(synth.random_batch(n_small, "qm9", seed=20 + i))
            b.to(gpu_device)
            assert torch.isfinite(mp(b)).all()
        assert not getattr(mp, "_dmpnn_no_mega", False)
        late.to(gpu_device)
        late = _bare(late)
        plan = engine.GraphPlan.from_bmg(late, light="tiles")
        hdr = plan.header()
        n_over = sum(1 for m in late_mgs if len(m.V) > 32 or m.edge_index.shape[1] > 48)
        assert n_over >= 2 and plan.tiles_only and hdr[0] & 15 == 0 and hdr[8] == n_over, hdr      # oversize tiles, no error flag
        out = mp(late)
        st = mp.__dict__.get("_dmpnn_replay")
        assert st is not None, "the batch must have taken the tile route (replay state present)"
        assert torch.isfinite(out).all()
        assert parity_err(out.cpu().numpy(), ref) <= TOL
        out2 = mp(late)                                                          # (the replayed argument block)
        assert torch.equal(out, out2)


@pytest.mark.gpu
def test_oversize_molecule_fp32_build_and_full_plan(gpu_device, monkeypatch):
    """The exact-fp32-MFMA build of the tile kernel and the full (CSR-row) plan take the same generic path."""
    from chemprop_amd import engine
    from chemprop_amd.nn import BondMessagePassing
    from oracle import dmpnn_torch as ot

    torch.manual_seed(6)
    mp = BondMessagePassing(d_h=64, depth=3).eval()
    mgs, b = _mixed(50, [("synth40", 2)], seed=1)
    with torch.no_grad():
        ref = ot.forward_bmg(b, ot.MPWeights.from_module(mp), depth=3).numpy()
    mp = mp.to(gpu_device)
    b.to(gpu_device)
    p = {k: v for k, v in mp.state_dict().items()}
    for mfma in ("f32", "split16"):
        plan = engine.GraphPlan.from_bmg(_bare(b))
        assert plan.header()[8] == 1
        with torch.no_grad():
            out, st = engine.forward(plan, b.V, b.E, p["W_i.weight"], p["W_h.weight"], p["W_o.weight"], p["W_o.bias"], depth=3,
                                     route="mega", mfma=mfma)
        assert st.route in ("mega", "mega16")
        assert parity_err(out.cpu().numpy(), ref) <= TOL, mfma
        # kept tensors (training forward): same output, and the kept rows of the oversize piece are written
        with torch.no_grad():
            out_k, st_k = engine.forward(plan, b.V, b.E, p["W_i.weight"], p["W_h.weight"], p["W_o.weight"], p["W_o.bias"], depth=3,
                                         route="mega", mfma=mfma, keep=True)
        # (round 4: the f16 tile kernel keeps M^(t) as split rows; the oversize piece's fp32 rows are converted at the end of its tile)
        assert parity_err(out_k.cpu().numpy(), ref) <= TOL and torch.isfinite(st_k.H0).all() and torch.isfinite(engine.kept_messages(st_k)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("act,n_small", [("relu", 60), ("elu", 60), ("tanh", 1200)])
def test_training_step_with_a_late_oversize_molecule(act, n_small, gpu_device):
    """(n_small = 1200: a batch beyond the single-workgroup plan — the full plan with the molecule tiles of the batch
    vector, dmpnn_prepare_with_batch, the oversize molecules as tiles of their own.)"""
    """Gradients through forward + backward tile kernels when the batch holds molecules larger than the tile
    (both generic paths), against autograd of the reference's op sequence."""
    from chemprop_amd import synth
    from chemprop_amd.nn import BondMessagePassing
    from oracle import dmpnn_torch as ot

    torch.manual_seed(7)
    cpu = BondMessagePassing(d_h=128, depth=3, activation=act, bias=True)
    mgs, late = _mixed(n_small, [("synth40", 5), ("zinc", 6)], seed=2)
    G = torch.randn(late.V.shape[0], 128, generator=torch.Generator().manual_seed(1))
    w = ot.MPWeights(cpu.W_i.weight, cpu.W_h.weight, cpu.W_o.weight, cpu.W_o.bias, cpu.W_i.bias, cpu.W_h.bias)
    ref = ot.forward_bmg(late, w, depth=3, activation=act)
    (ref * G).sum().backward()
    gref = {n: p.grad.clone() for n, p in cpu.named_parameters()}
    mp = BondMessagePassing(d_h=128, depth=3, activation=act, bias=True)
    mp.load_state_dict(cpu.state_dict())
    mp = mp.to(gpu_device).train()
    for i in range(3):  # the validated window passes on ordinary batches
        b = _bare(synth.random_batch(60, "qm9", seed=30 + i))
        b.to(gpu_device)
        mp(b).sum().backward()
    mp.zero_grad()
    late.to(gpu_device)
    out = mp(_bare(late))
    st = out.grad_fn.st
    assert st.route == "mega16" and st.plan.header()[8] == 2 and st.plan.any_size == (n_small > 500)
    assert parity_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= TOL
    (out * G.to(gpu_device)).sum().backward()
    for n, p in mp.named_parameters():
        assert parity_err(p.grad.cpu().numpy(), gref[n].numpy()) <= 2e-5, n


@pytest.mark.gpu
def test_loader_table_with_an_oversize_entry(gpu_device):
    """A tile table that came with the batch may hold an oversize tile too (dmpnn_pack_tiles gives such a molecule its own)."""
    import ctypes as C

    from chemprop_amd import _lib, engine
    from chemprop_amd.nn import BondMessagePassing
    from oracle import dmpnn_torch as ot

    mgs, b = _mixed(300, [("synth40", 8)], seed=3)
    n_at = np.array([len(m.V) for m in mgs]); n_ed = np.array([m.edge_index.shape[1] for m in mgs])
    ao = np.concatenate([[0], np.cumsum(n_at)]).astype(np.int32); eo = np.concatenate([[0], np.cumsum(n_ed)]).astype(np.int32)
    lib = _lib.load()
    cap = int(lib.dmpnn_max_tiles(int(ao[-1]), int(eo[-1]))) + 1
    tr, ta = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    n = int(lib.dmpnn_pack_tiles(ao.ctypes.data, eo.ctypes.data, len(mgs), tr.ctypes.data, ta.ctypes.data, cap))
    assert n > 0
    torch.manual_seed(3)
    mp = BondMessagePassing().eval()
    with torch.no_grad():
        ref = ot.forward_bmg(b, ot.MPWeights.from_module(mp), depth=3).numpy()
    b.to(gpu_device)
    tiles = (torch.from_numpy(tr[:n + 1]).to(gpu_device), torch.from_numpy(ta[:n + 1]).to(gpu_device), n)
    plan = engine.GraphPlan(b.edge_index, b.rev_edge_index, int(b.V.shape[0]), light="tiles", batch=b.batch, tiles=tiles)
    hdr = plan.header()
    assert plan.loader_tiles == n and hdr[0] & 15 == 0 and hdr[8] == 1
    p = {k: v.to(gpu_device) for k, v in mp.state_dict().items()}
    with torch.no_grad():
        out, st = engine.forward(plan, b.V, b.E, p["W_i.weight"], p["W_h.weight"], p["W_o.weight"], p["W_o.bias"], depth=3)
    assert st.route == "mega16" and parity_err(out.cpu().numpy(), ref) <= TOL


@pytest.mark.gpu
def test_host_known_oversize_batch_takes_the_per_step_route(gpu_device):
    from chemprop_amd.nn import BondMessagePassing
    from oracle import dmpnn_torch as ot

    torch.manual_seed(4)
    mp = BondMessagePassing().eval()
    mgs, b = _mixed(100, [("synth40", 1)], seed=4)
    with torch.no_grad():
        ref = ot.forward_bmg(b, ot.MPWeights.from_module(mp), depth=3).numpy()
    mp = mp.to(gpu_device)
    b.to(gpu_device)
    assert b.oversize is True
    with torch.no_grad():
        for _ in range(4):
            out = mp(b)
            assert mp.__dict__.get("_dmpnn_replay") is None          # never the tile route for this batch
            assert parity_err(out.cpu().numpy(), ref) <= TOL


@pytest.mark.gpu
def test_module_that_keeps_meeting_oversize_molecules_moves_to_the_per_step_routes(gpu_device):
    """Speed heuristic (never correctness): the asynchronous look at the plan header switches the tile route off."""
    from chemprop_amd import synth
    from chemprop_amd.nn import BondMessagePassing
    from oracle import dmpnn_torch as ot

    torch.manual_seed(5)
    mp = BondMessagePassing(d_h=64).eval()
    mgs, late = _mixed(64, [("synth40", 1)], seed=5)
    with torch.no_grad():
        ref = ot.forward_bmg(late, ot.MPWeights.from_module(mp), depth=3).numpy()
    mp = mp.to(gpu_device)
    with torch.no_grad():
        for i in range(3):
            b = _bare(synth.random_batch(64, "qm9", seed=40 + i))
            b.to(gpu_device)
            mp(b)
        late.to(gpu_device)
        late = _bare(late)
        for i in range(40):
            out = mp(late)
            torch.cuda.synchronize()
            if i % 13 == 0:
                assert parity_err(out.cpu().numpy(), ref) <= TOL, i
    assert getattr(mp, "_dmpnn_no_mega", False)
    with torch.no_grad():
        assert parity_err(mp(late).cpu().numpy(), ref) <= TOL
