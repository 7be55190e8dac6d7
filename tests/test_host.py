"""Host logic: the BatchMolGraph / BondMessagePassing mirrors follow the reference contract."""
import numpy as np
import pytest
import torch

from chemprop_amd import synth
from chemprop_amd.data import BatchMolGraph, MolGraph
from chemprop_amd.nn import BondMessagePassing, classify_activation
from oracle import dmpnn_numpy as onp
from oracle import ref_shim


def test_batching_semantics():
    """collate.py:37-62: offsets, dtypes, batch vector, len()."""
    mgs = synth.random_molgraphs(5, "qm9", seed=0)
    b = BatchMolGraph(mgs)
    assert len(b) == 5
    assert b.V.dtype == torch.float32 and b.E.dtype == torch.float32
    assert b.edge_index.dtype == torch.int64 and b.rev_edge_index.dtype == torch.int64 and b.batch.dtype == torch.int64
    nV = sum(len(m.V) for m in mgs)
    nE = sum(m.edge_index.shape[1] for m in mgs)
    assert b.V.shape == (nV, 72) and b.E.shape == (nE, 14) and b.edge_index.shape == (2, nE)
    assert torch.all(b.batch[1:] >= b.batch[:-1]) and int(b.batch[-1]) == 4
    # edges never cross molecules and rev is an involution running dst -> src
    assert torch.equal(b.batch[b.edge_index[0]], b.batch[b.edge_index[1]])
    assert onp.graph_is_symmetric(b.edge_index[0].numpy(), b.edge_index[1].numpy(), b.rev_edge_index.numpy())
    assert b.to("cpu") is None  # in-place move returning None, like collate.py:68-73


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not present")
def test_batching_equals_reference_collate():
    _, BMG, _ = ref_shim.load_reference()
    for layout in ("interleaved", "block", "shuffled"):
        mgs = synth.random_molgraphs(7, "zinc", seed=3, layout=layout)
        mine, ref = BatchMolGraph(mgs), BMG(mgs)
        for f in ("V", "E", "edge_index", "rev_edge_index", "batch"):
            assert torch.equal(getattr(mine, f), getattr(ref, f)), f
        assert len(mine) == len(ref)


@pytest.mark.parametrize("layout", ["interleaved", "block", "shuffled"])
def test_synth_graphs_are_valid_molecules(layout):
    for kind in ("qm9", "zinc", "synth40", "cgr"):
        for mg in synth.random_molgraphs(20, kind, seed=1, layout=layout):
            src, dst = mg.edge_index
            assert onp.graph_is_symmetric(src, dst, mg.rev_edge_index)
            assert np.bincount(dst, minlength=len(mg.V)).max() <= 4
            assert np.array_equal(mg.E, mg.E[mg.rev_edge_index])  # both directions share bond features
            assert mg.V.shape[1] == (106 if kind == "cgr" else 72)


def test_qm9_shape_statistics():
    b = synth.random_batch(512, "qm9", seed=0)
    assert 7.5 < b.V.shape[0] / 512 < 10.5
    assert 1.9 < b.E.shape[0] / b.V.shape[0] < 2.3


def test_module_mirror_contract():
    """state_dict keys / shapes / hparams of base.py:75-92,238-251."""
    mp = BondMessagePassing()
    sd = mp.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        "W_i.weight": (300, 86), "W_h.weight": (300, 300), "W_o.weight": (300, 372), "W_o.bias": (300,)}
    assert mp.output_dim == 300 and mp.W_i.in_features == 86
    assert mp.hparams["cls"] is BondMessagePassing and mp.hparams["d_h"] == 300
    clone = mp.hparams["cls"](**{k: v for k, v in mp.hparams.items() if k != "cls"})
    assert clone.state_dict().keys() == sd.keys()
    mp2 = BondMessagePassing(d_v=10, d_e=6, d_h=24, bias=True, d_vd=5, activation="prelu")
    assert set(mp2.state_dict()) == {"W_i.weight", "W_i.bias", "W_h.weight", "W_h.bias", "W_o.weight", "W_o.bias",
                                     "W_d.weight", "W_d.bias", "tau.weight"}
    assert mp2.output_dim == 29


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not present")
def test_same_seed_same_initial_weights_as_reference():
    BMP, _, _ = ref_shim.load_reference()
    for kw in (dict(), dict(d_h=64, bias=True, d_vd=3)):
        torch.manual_seed(11)
        ref = BMP(**kw)
        torch.manual_seed(11)
        mine = BondMessagePassing(**kw)
        for k, v in ref.state_dict().items():
            assert torch.equal(v, mine.state_dict()[k]), k
        mine.load_state_dict(ref.state_dict())  # and the reference's files load unchanged


def test_activation_classification():
    nn = torch.nn
    assert classify_activation(nn.ReLU())[0] == "relu"
    assert classify_activation(nn.LeakyReLU(0.1))[:2] == ("leakyrelu", pytest.approx(0.1))
    assert classify_activation(nn.PReLU())[0] == "prelu"
    assert classify_activation(nn.PReLU(4))[0] == "custom"
    assert classify_activation(nn.ELU())[0] == "elu" and classify_activation(nn.ELU(0.5))[0] == "custom"
    assert classify_activation(nn.Tanh())[0] == "tanh" and classify_activation(nn.Softplus())[0] == "custom"


def test_no_cpu_fallback():
    mp = BondMessagePassing(d_h=16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mp(synth.random_batch(2, "qm9", seed=0))


def test_product_does_not_import_oracle():
    """The shipped package must never reach into oracle/ (parity claims depend on it)."""
    import glob
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "chemprop_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert "import oracle" not in src and "from oracle" not in src, path


def test_accelerate_swaps_the_real_reference_classes():
    """`integration.accelerate` on a model made of the REAL chemprop classes (imported through the shim,
    build container only): the block and the aggregation become HIP subclasses in place, parameters,
    hparams (`cls` stays the reference class) and state_dict keys untouched."""
    import torch

    from oracle import ref_shim

    if not ref_shim.reference_available():
        pytest.skip("/root/reference absent (GPU box)")
    ref_shim.install()
    from chemprop.nn import BondMessagePassing as RefMP
    from chemprop.nn import MABAtomMessagePassing as RefMABAtom
    from chemprop.nn import MABBondMessagePassing as RefMABBond
    from chemprop.nn.agg import NormAggregation as RefNorm
    from chemprop.nn.ffn import MLP as RefMLP

    from chemprop_amd import integration

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.message_passing = RefMP(d_h=16)
            self.agg = RefNorm(norm=3.0)
            self.ffn = RefMLP.build(16, 2, hidden_dim=8)
            self.mab = torch.nn.ModuleList([RefMABBond(d_h=8, return_vertex_embeddings=False), RefMABAtom(d_h=8, d_ed=2)])

    m = Model()
    keys, w = list(m.state_dict().keys()), m.message_passing.W_h.weight
    assert integration.accelerate(m) == 5
    assert isinstance(m.ffn, RefMLP) and type(m.ffn) is not RefMLP and m.ffn.output_dim == 2
    for blk, Ref in zip(m.mab, (RefMABBond, RefMABAtom)):
        assert isinstance(blk, Ref) and type(blk) is not Ref and blk.hparams["cls"] is Ref
    assert m.mab[0].atom_messages is False and m.mab[1].atom_messages is True and m.mab[0].W_vo is None
    assert m.mab[1].output_dims == (8, 10)
    assert isinstance(m.message_passing, RefMP) and type(m.message_passing) is not RefMP
    assert isinstance(m.agg, RefNorm) and type(m.agg) is not RefNorm
    assert m.message_passing.hparams["cls"] is RefMP and m.agg.hparams["cls"] is RefNorm and m.agg.norm == 3.0
    assert list(m.state_dict().keys()) == keys and m.message_passing.W_h.weight is w
    with pytest.raises(RuntimeError):  # CPU tensors: the engine has no fallback
        m.agg(torch.zeros(3, 16), torch.zeros(3, dtype=torch.int64))


def test_committed_bench_line_follows_the_contract():
    """profiles/r01_bench.json is a line `bench.py` printed on MI355X: the keys the driver / judge read are all there and
    consistent with each other (roofline.frac = achieved / peak, value = units / time)."""
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.loads(open(os.path.join(root, "profiles", "r01_bench.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference")
    edges = d["config"]["directed_edges_per_gpu"]
    assert abs(d["value"] - d["n_gpus"] * edges * 2 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]  # depth 3: 2 updates / edge


def test_plan_policy_decision_table(monkeypatch):
    """Which plan an inference forward builds (chemprop_amd/nn.py): full plan while the module's first batches are being
    validated, then the light / tile plan; the tile plan needs small molecules and either a batch the single-workgroup
    plan takes or a table / batch vector for any size.  Host logic only."""
    import torch

    from chemprop_amd import engine
    from chemprop_amd.nn import BondMessagePassing, _light_plan_ok, _tile_plan_ok

    for k in ("DMPNN_VALIDATE", "DMPNN_GENERAL", "DMPNN_MEGA", "DMPNN_MFMA"):
        monkeypatch.delenv(k, raising=False)
    mp = BondMessagePassing().eval()
    with torch.no_grad():
        assert not _light_plan_ok(mp)                       # first batches: validated on the full plan
        object.__setattr__(mp, "_dmpnn_batches_checked", 2)
        assert _light_plan_ok(mp)
        monkeypatch.setenv("DMPNN_GENERAL", "1")
        assert not _light_plan_ok(mp)
        monkeypatch.delenv("DMPNN_GENERAL")
        assert not _light_plan_ok(BondMessagePassing(undirected=True).eval())
        assert not _light_plan_ok(BondMessagePassing(d_h=512).eval())           # wider than the fused routes
        fresh = BondMessagePassing().eval()
        monkeypatch.setenv("DMPNN_VALIDATE", "never")
        assert _light_plan_ok(fresh)                        # nothing to wait for
        monkeypatch.delenv("DMPNN_VALIDATE")
    assert not _light_plan_ok(mp)                           # grad enabled, parameters require grad: training builds the full plan
    # the tile plan: QM9-like batches
    assert engine.small_plan_fits(4636, 9120) and not engine.small_plan_fits(37000, 73000)
    assert _tile_plan_ok(mp, 4636, 9120, 512)
    assert not _tile_plan_ok(mp, 4636, 9120, 0)             # molecule count unknown
    assert not _tile_plan_ok(mp, 12000, 25000, 512)         # ~49 directed edges per molecule: tiles would not fit them
    assert not _tile_plan_ok(mp, 37000, 73000, 4096)        # beyond the single-workgroup plan without a table / batch vector
    assert _tile_plan_ok(mp, 37000, 73000, 4096, True)      # ... with one
    for k, v in (("DMPNN_MEGA", "0"), ("DMPNN_MFMA", "f32"), ("DMPNN_VALIDATE", "always")):
        monkeypatch.setenv(k, v)
        assert not _tile_plan_ok(mp, 4636, 9120, 512), k
        monkeypatch.delenv(k)
    object.__setattr__(mp, "_dmpnn_no_mega", True)          # the module has seen an oversize molecule
    assert not _tile_plan_ok(mp, 4636, 9120, 512)


def test_forward_route_is_one_shape_rule():
    """dmpnn_forward_route (include/dmpnn.h): the default policy as a pure function of shapes, enumerated (no GPU)."""
    import ctypes as C

    from chemprop_amd import _lib

    lib = _lib.load()
    R = {n: i for i, n in enumerate(_lib.ROUTES)}

    def args(nV, nE, d_h=300, d_v=72, d_e=14, flags=0):
        a = _lib.FwdArgs()
        a.n_atoms, a.n_edges, a.d_v, a.d_e, a.d_h, a.depth, a.flags = nV, nE, d_v, d_e, d_h, 3, flags
        a.ldv, a.lde, a.ldh, a.ldout = d_v, d_e, (d_h + 3) // 4 * 4, d_h
        for f in ("V", "E", "W_i", "W_h", "H0", "Ms", "Mv"):
            setattr(a, f, 4096)
        return a

    route = lambda a, keep=0, cap=2, plan=0, arith=0: lib.dmpnn_forward_route(C.byref(a), keep, cap, plan, arith)
    qm9, big = args(4636, 9120), args(37000, 73000, flags=_lib.F_LOADER_TILES)
    # QM9-sized molecules: the whole-forward tile kernel, inference and training, at any batch size with molecule tiles
    assert route(qm9) == route(qm9, keep=1) == route(qm9, plan=2) == R["mega16"] and route(qm9, arith=1) == R["mega"]
    assert route(big) == route(big, keep=1) == R["mega16"]
    # molecules beyond the tile (cap 1): per-step fused route on the f16 pipe for inference — and, round 4, for TRAINING at size with
    # a ReLU-class activation (its lean forward + the backward step kernels); general16 for training with anything else
    assert route(big, cap=1) == route(big, cap=1, plan=1) == R["fused16"]
    assert route(big, keep=1, cap=1) == R["fused16"]
    for mod in (dict(act=_lib.ACT["tanh"]), dict(act=_lib.ACT["prelu"]), dict(depth=1), dict(dropout_p=0.5), dict(W_d=4096, d_vd=8, V_d=4096, ldvd=8)):
        t = args(37000, 73000, flags=_lib.F_LOADER_TILES)
        for k, v in mod.items():
            setattr(t, k, v)
        assert route(t, keep=1, cap=1) == R["general16"], mod
    assert route(big, keep=1, cap=1, plan=1) == -1 and route(big, keep=1, cap=0) == R["general16"]
    assert route(qm9, cap=1) == R["fused16"] and route(qm9, keep=1, cap=1) == R["fused"] and route(qm9, cap=1, arith=1) == R["fused"]
    small = args(300, 600)
    assert route(small, cap=1) == R["fused"] and route(small, cap=0) == R["general"]          # below the crossover: fp32-MFMA kernels
    # wide hidden layers: no fp32 fused route; fused16 up to 640 columns, general16 beyond and for training
    assert route(args(12000, 25000, d_h=512)) == R["fused16"] and route(args(12000, 25000, d_h=512), keep=1) == R["general16"]
    assert route(args(12000, 25000, d_h=1024)) == R["general16"] and route(args(300, 600, d_h=1024), arith=1) == R["general"]
    # undirected, odd widths: the general route
    assert route(args(4636, 9120, flags=_lib.F_UNDIRECTED)) == R["general"] and route(args(37000, 73000, flags=_lib.F_UNDIRECTED)) == R["general16"]
    assert route(args(4636, 9120, d_e=13)) == R["general"]
    # plans that cannot serve: a tile plan without the tile kernel, a light plan for training / the general route
    assert route(qm9, cap=1, plan=2) == -1 and route(qm9, arith=1, plan=2) == -1
    assert route(qm9, keep=1, plan=2) == R["mega16"]          # training on a tile plan (DMPNN_F_TILE_PLAN) ...
    vd = args(4636, 9120); vd.W_d = 4096; vd.d_vd = 8; vd.V_d = 4096; vd.ldvd = 8
    assert route(vd, keep=1, plan=2) == -1 and route(vd, plan=2) == R["mega16"]   # ... but not with W_d
    assert route(qm9, keep=1, plan=1) == -1 and route(args(4636, 9120, d_e=13), plan=1) == -1
    assert lib.dmpnn_forward_route(None, 0, 2, 0, 0) == -1 and route(qm9, cap=-1) == -1


def test_dropout_hash_restatement_is_the_library_s_and_behaves_like_bernoulli():
    """oracle/dropout_hash.py == dmpnn_dropout_keep (the host twin of the device hash), element for element; keep fraction 1 - p;
    sites, seeds and neighbouring elements are uncorrelated (what nn.Dropout's Bernoulli mask guarantees, base.py:85)."""
    import numpy as np

    from chemprop_amd import _lib
    from oracle import dropout_hash as dh

    lib = _lib.load()
    rng = np.random.default_rng(0)
    for seed, site, p in ((1, 0, 0.25), (0x1234_5678_9ABC_DEF0, 3, 0.5), (2 ** 62 - 1, 1, 0.1), (77, 2, 0.9)):
        rows = rng.integers(0, 2 ** 31 - 1, size=40)
        cols = rng.integers(0, 1024, size=24)
        want = dh.drop_hash(seed, site, rows, cols) >= np.uint32(dh.threshold(p))
        got = np.array([[lib.dmpnn_dropout_keep(seed, site, int(r), int(c), p) for c in cols] for r in rows], dtype=bool)
        assert np.array_equal(want, got), (seed, site, p)
    assert lib.dmpnn_dropout_keep(5, 0, 1, 1, 0.0) == 1            # p = 0: everything is kept
    for p in (0.1, 0.25, 0.5):
        m0 = dh.keep_mask(42, 0, 4096, 300, p)
        m1 = dh.keep_mask(42, 1, 4096, 300, p)          # another site
        m2 = dh.keep_mask(43, 0, 4096, 300, p)          # another seed
        n = m0.size
        tol = 5 * np.sqrt(p * (1 - p) / n)
        for m in (m0, m1, m2):
            assert abs(m.mean() - (1 - p)) <= tol
        for a, b in ((m0, m1), (m0, m2), (m0[:, :-1], m0[:, 1:]), (m0[:-1], m0[1:])):
            assert abs((a & b).mean() - (1 - p) ** 2) <= 6 / np.sqrt(a.size) + 1e-3      # independent: P(both kept) = (1 - p)^2
        assert abs(m0.mean(axis=0).std() - np.sqrt(p * (1 - p) / 4096)) <= 0.3 * np.sqrt(p * (1 - p) / 4096)   # columns look alike
    assert np.array_equal(dh.keep_mask(9, 0, 50, 64, 0.3), dh.keep_mask(9, 0, 50, 64, 0.3))      # a function of its arguments


def test_training_plan_kind_is_a_host_rule(monkeypatch):
    """``nn._training_plan_kind``: which plan a TRAINING forward asks for — the tile table (``DMPNN_F_TILE_PLAN``) after the validated
    first batches for blocks and batches bound for the tile kernels, the full CSR plan otherwise.  Pure host logic (shapes, dtypes,
    module attributes): enumerated here without a GPU."""
    import torch

    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph
    from chemprop_amd.nn import BondMessagePassing, _training_plan_kind, _VALIDATE_FIRST_N

    qm9 = synth.random_batch(64, "qm9", seed=1)
    big = synth.random_batch(8, "synth40", seed=1)            # molecules beyond the tile: the host's batching code knows (oversize True)
    mp = BondMessagePassing().train()
    assert _training_plan_kind(mp, qm9) is False               # the first batches of a module are validated on full plans
    object.__setattr__(mp, "_dmpnn_batches_checked", _VALIDATE_FIRST_N)
    assert _training_plan_kind(mp, qm9) == "tiles"
    assert _training_plan_kind(mp, big) is False
    bare = BatchMolGraph.from_tensors(qm9.V, qm9.E, qm9.edge_index, qm9.rev_edge_index, qm9.batch, len(qm9))
    assert _training_plan_kind(mp, bare) == "tiles"            # (oversize unknown: the kernels' generic path covers what turns up)
    monkeypatch.setenv("DMPNN_TRAIN_PLAN", "full")
    assert _training_plan_kind(mp, qm9) is False
    monkeypatch.delenv("DMPNN_TRAIN_PLAN")
    monkeypatch.setenv("DMPNN_VALIDATE", "always")             # (the per-batch verdict is read from a full plan)
    assert _training_plan_kind(mp, qm9) is False
    monkeypatch.delenv("DMPNN_VALIDATE")
    # blocks the tile kernels' backward does not take on a tile plan
    for kw in (dict(undirected=True), dict(d_vd=4), dict(activation="prelu"), dict(activation=torch.nn.Softplus()), dict(d_h=302),
               dict(d_h=512), dict(dropout=0.2, activation="tanh")):
        m2 = BondMessagePassing(**kw).train()
        object.__setattr__(m2, "_dmpnn_batches_checked", _VALIDATE_FIRST_N)
        assert _training_plan_kind(m2, qm9) is False, kw
    m3 = BondMessagePassing(dropout=0.2).train()               # ReLU + nn.Dropout: dropout inside the tile kernels
    object.__setattr__(m3, "_dmpnn_batches_checked", _VALIDATE_FIRST_N)
    assert _training_plan_kind(m3, qm9) == "tiles"
    for p in (mp.W_i.weight, mp.W_h.weight):                    # nothing to differentiate in the edge part: the tile backward would refuse
        p.requires_grad_(False)
    assert _training_plan_kind(mp, qm9) is False
    mp.W_h.weight.requires_grad_(True)
    assert _training_plan_kind(mp, qm9) == "tiles"
    object.__setattr__(mp, "_dmpnn_no_mega", True)             # a module that keeps meeting oversize molecules
    assert _training_plan_kind(mp, qm9) is False


def test_keep_bits_bytes_is_a_shape_rule():
    """``dmpnn_forward_keep_bits_bytes`` (include/dmpnn.h): depth x tile bound x 2 KB for a training forward of the tile kernel on a tile
    plan with a ReLU-class activation and no dropout / W_d — 0 for everything else (the caller then keeps fp32 rows)."""
    import ctypes as C

    from chemprop_amd import _lib

    lib = _lib.load()
    need = _lib.F_TILE_PLAN | _lib.F_KEEP | _lib.F_MEGA | _lib.F_SPLIT16 | _lib.F_FUSED

    def args(flags=need, act="relu", p=0.0, depth=3, nV=4636, nE=9120, wd=0):
        a = _lib.FwdArgs()
        a.n_atoms, a.n_edges, a.d_v, a.d_e, a.d_h, a.depth, a.flags = nV, nE, 72, 14, 300, depth, flags
        a.act, a.dropout_p, a.W_d = _lib.ACT[act], p, wd
        return a

    off = (C.c_int64 * _lib.PLAN_NOFFSETS)()
    assert lib.dmpnn_plan_layout(4636, 9120, off) == 0
    tiles = int(lib.dmpnn_max_tiles(4636, 9120)) if hasattr(lib, "dmpnn_max_tiles") else None
    n = int(lib.dmpnn_forward_keep_bits_bytes(C.byref(args())))
    assert n > 0 and n % (3 * 2048) == 0                         # depth slots of 2 KB per tile of the launch bound
    if tiles:
        assert n >= 3 * 2048 * (9120 // 48)                        # ... which covers every tile the batch can have
    assert int(lib.dmpnn_forward_keep_bits_bytes(C.byref(args(depth=5)))) == n // 3 * 5
    for a in (args(act="leakyrelu"), args(act="none") if "none" in _lib.ACT else args()):
        assert int(lib.dmpnn_forward_keep_bits_bytes(C.byref(a))) == n
    for a in (args(flags=need & ~_lib.F_TILE_PLAN), args(flags=need & ~_lib.F_KEEP), args(act="tanh"), args(act="elu"), args(p=0.1),
              args(wd=4096), args(nE=0), args(nV=0)):
        assert int(lib.dmpnn_forward_keep_bits_bytes(C.byref(a))) == 0
    assert int(lib.dmpnn_forward_keep_bits_bytes(None)) == 0


def test_split_row_format_round_trip_and_the_keep_rows_rule(monkeypatch):
    """The split-row format of csrc/dmpnn_step16_impl.hpp (chunks of [hi 32 halfs | lo 32 halfs] + a 16-byte tail with the row's scale)
    restated in numpy against `engine.split_rows_to_float`, and the host's size rule for keeping the tile kernel's messages so."""
    from chemprop_amd import _lib, engine

    rng = np.random.default_rng(0)
    n_rows, d_h = 7, 300
    srf = int(_lib.load().dmpnn_split_row_floats(d_h))
    assert srf * 4 == ((d_h + 63) // 64 * 64) * 4 + 16
    x = (rng.standard_normal((n_rows, d_h)) * np.array([1e-3, 1.0, 37.0, 1e3, 0.0, 2.5, 1e-6])[:, None]).astype(np.float32)
    raw = np.zeros((n_rows, srf * 2), np.float16)
    tails = np.zeros((n_rows, 4), np.float32)
    for r in range(n_rows):
        mx = float(np.abs(x[r]).max())
        s = 1.0 if not (mx > 0) else float(2.0 ** (14 - np.frexp(mx)[1]))     # scale_for: the row maximum at [2^13, 2^14)
        y = x[r].astype(np.float64) * s
        hi = y.astype(np.float16)
        lo = (y - hi.astype(np.float64)).astype(np.float16)
        for c in range((d_h + 31) // 32):
            n = min(32, d_h - 32 * c)
            raw[r, 64 * c: 64 * c + n] = hi[32 * c: 32 * c + n]
            raw[r, 64 * c + 32: 64 * c + 32 + n] = lo[32 * c: 32 * c + n]
        tails[r] = (s, 0.0 if mx > 0 else 1.0, 0.0, 0.0)
    rows = torch.from_numpy(raw.view(np.float32).copy())
    rows[:, srf - 4:] = torch.from_numpy(tails)
    back = engine.split_rows_to_float(rows, d_h).numpy()
    # hi + lo carries 22 significant bits relative to the row's largest element: fp32-class
    for r in range(n_rows):
        tol = 2.0 ** -21 * float(np.abs(x[r]).max())
        assert np.abs(back[r] - x[r]).max() <= tol, r
    # the rule (the library's: dmpnn_train_route.keep_rows): from DMPNN_KEEP_ROWS_MIN message rows on, unless forced
    monkeypatch.delenv("DMPNN_KEEP_ROWS", raising=False)
    kr = lambda n_edges, depth=3: bool(engine.train_route(n_edges // 2, n_edges, 72, 14, 300, depth, "relu", n_edges // 20).keep_rows)
    half = engine.KEEP_ROWS_MIN // 2
    assert not kr(half - 1) and kr(half) and kr((engine.KEEP_ROWS_MIN + 4) // 5, depth=6) and not kr(10 ** 6, depth=1)
    monkeypatch.setenv("DMPNN_KEEP_ROWS", "1")
    assert kr(2)
    monkeypatch.setenv("DMPNN_KEEP_ROWS", "0")
    assert not kr(10 ** 8)


def test_train_route_is_one_rule_in_the_library(monkeypatch):
    """``dmpnn_train_route`` (include/dmpnn.h, round-4 VERDICT weak #10): the training-plan rule that lived in three places of the host
    code, enumerated against its restatement — which plan K0 builds, the route on it, the form of the kept tensors (no GPU)."""
    import ctypes as C

    from chemprop_amd import _lib, engine

    for k in ("DMPNN_KEEP_ROWS", "DMPNN_MEGA", "DMPNN_MFMA"):
        monkeypatch.delenv(k, raising=False)
    lib = _lib.load()
    R = _lib.ROUTES

    def want_tiles(nV, nE, d_v, d_e, d_h, act, n_mols, undirected, has_vd, p, have_batch, have_table, oversize, cap):
        if undirected or has_vd or act not in ("none", "relu", "leakyrelu", "tanh", "elu") or cap < 2 or nE <= 0 or nV <= 0 or oversize is True:
            return False
        if p > 0 and act not in ("relu", "leakyrelu"):
            return False
        if not (d_h % 4 == 0 and d_h <= 320 and d_v % 2 == 0 and d_e % 2 == 0):
            return False
        small = engine.small_plan_fits(nV, nE)
        large = have_batch and bool(lib.dmpnn_tile_plan_any_size(nV, nE))
        return (small or have_table or large) and n_mols > 0 and nE <= 30 * n_mols

    seen = set()
    for (nV, nE, n_mols) in ((4636, 9120, 512), (37000, 72800, 4096), (20500, 43800, 512), (166000, 355702, 4096), (9, 16, 1), (60, 130, 2)):
        for d_h in (300, 320, 512, 302):
            for act in ("relu", "tanh", "prelu"):
                for (undirected, has_vd, p) in ((False, False, 0.0), (True, False, 0.0), (False, True, 0.0), (False, False, 0.2)):
                    for (have_batch, have_table) in ((False, False), (True, False), (False, True)):
                        for oversize in (None, False, True):
                            for cap in (2, 1):
                                info = engine.train_route(nV, nE, 72, 14, d_h, 3, act, n_mols, undirected=undirected, has_vd=has_vd, dropout_p=p,
                                                          have_batch=have_batch, have_table=have_table, oversize=oversize, max_level=cap)
                                w = want_tiles(nV, nE, 72, 14, d_h, act, n_mols, undirected, has_vd, p, have_batch, have_table, oversize, cap)
                                key = (nV, d_h, act, undirected, has_vd, p, have_batch, have_table, oversize, cap)
                                if w:   # (the tile plan serves the tile kernel or nothing)
                                    assert info.plan_kind == 2 and R[info.route] == "mega16", key
                                else:
                                    assert info.plan_kind == 0 and info.route >= 0, key
                                assert info.keep_bits == int((info.plan_kind == 2 and act == "relu" and p == 0.0) or bool(info.lean)), key
                                if info.lean:
                                    assert R[info.route] == "fused16" and act == "relu" and not has_vd and p == 0.0 and d_h <= 320 and nE >= 20000, key
                                seen.add((info.plan_kind, R[info.route], info.keep_rows, info.keep_bits, info.lean))
    # the combinations BASELINE's configs live on all occur: tile plan + sign bits (qm9-512), + split rows (qm9-4096), the lean
    # per-step fused route (synth40-4096), the per-step general route on the f16 pipe (h 512 training)
    assert (2, "mega16", 0, 1, 0) in seen and (2, "mega16", 1, 1, 0) in seen and (0, "fused16", 1, 1, 1) in seen and any(r == "general16" for _, r, *_ in seen)
    bad = _lib.TrainRouteInfo()
    assert lib.dmpnn_train_route(None, 1, 0, -1, 2, 0, -1, C.byref(bad)) != 0


def test_mab_tile_training_rule_is_host_logic():
    """`mab._tile_train_ok`: which training forwards of the mol-atom-bond blocks go to the tile kernels (everything else keeps the
    per-step chain) — decided on the host from the module and the batch alone."""
    from chemprop_amd import mab

    bmg = synth.random_batch(4, "qm9", seed=0)
    ok = lambda m, V_d=None, b=bmg: mab._tile_train_ok(m.train(), b, V_d)
    assert ok(mab.MABBondMessagePassing(d_h=32)) and ok(mab.MABAtomMessagePassing(d_h=32))
    assert ok(mab.MABBondMessagePassing(d_h=32, d_vd=3), torch.zeros(bmg.V.shape[0], 3))           # W_vd: inside the bond block's chain
    assert not ok(mab.MABAtomMessagePassing(d_h=32, d_vd=3), torch.zeros(bmg.V.shape[0], 3))        # ... not with atom messages
    assert not ok(mab.MABBondMessagePassing(d_h=32, return_vertex_embeddings=False))                 # the tile kernel's finalize IS the vertex read-out
    assert ok(mab.MABBondMessagePassing(d_h=32, return_edge_embeddings=False, depth=1))
    assert not ok(mab.MABBondMessagePassing(d_h=32, depth=1))                                        # the edge read-out reads a kept H^(depth-1)
    assert not ok(mab.MABBondMessagePassing(d_h=32, undirected=True))
    assert not ok(mab.MABBondMessagePassing(d_h=32, dropout=0.1))
    assert not ok(mab.MABBondMessagePassing(d_h=32, activation="prelu")) and not ok(mab.MABBondMessagePassing(d_h=32, activation=torch.nn.Softplus()))
    with torch.no_grad():
        assert not ok(mab.MABBondMessagePassing(d_h=32))
    frozen = mab.MABBondMessagePassing(d_h=32)
    for p in frozen.parameters():
        p.requires_grad_(False)
    assert not ok(frozen)
    odd = BatchMolGraph.from_tensors(bmg.V, bmg.E[:, :13].contiguous(), bmg.edge_index, bmg.rev_edge_index, bmg.batch, len(bmg))
    assert not ok(mab.MABAtomMessagePassing(d_e=13, d_h=32), b=odd) and ok(mab.MABBondMessagePassing(d_e=13, d_h=32), b=odd)   # (the engine then refuses the odd width: RouteUnavailable -> the chain)
