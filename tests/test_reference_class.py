"""The REAL reference classes on a device (round-1 VERDICT rows a10 / a11, "Next round" item 2).

``HipBondMessagePassing`` is a subclass of the reference's own ``chemprop.nn.BondMessagePassing`` that overrides
``forward`` only (``chemprop_amd/integration.py``).  The reference is imported through ``oracle/ref_shim.py`` from
``/root/reference`` (build container) or from ``oracle/_ref/`` — the git-ignored staging copy ``oracle/stage_ref.py``
makes at build time, which travels to the GPU box.  Every comparison is against ``Ref.forward`` executed on the CPU.
"""
import copy

import numpy as np
import pytest
import torch

from conftest import TOL, parity_err
from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="no reference tree (/root/reference or oracle/_ref)")


def _ref():
    BMP, BMG, MG = ref_shim.load_reference()
    Multi, GT, ST, MPNN, cnn = ref_shim.load_reference_extras()
    return dict(BMP=BMP, BMG=BMG, MG=MG, Multi=Multi, GT=GT, ST=ST, MPNN=MPNN, nn=cnn)


def _bmg(R, n, kind, seed):
    from chemprop_amd import synth

    return R["BMG"](synth.random_molgraphs(n, kind, seed=seed))


def test_staged_reference_imports_and_runs_on_cpu():
    R = _ref()
    torch.manual_seed(0)
    mp = R["BMP"]().eval()
    with torch.no_grad():
        out = mp(_bmg(R, 4, "qm9", 0))
    assert out.shape[1] == 300 and torch.isfinite(out).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kw,kind,n", [(dict(), "qm9", 64), (dict(d_h=128, depth=4, bias=True, activation="elu"), "zinc", 20),
                                       (dict(d_v=106, d_e=28), "cgr", 64)])
def test_real_subclass_forward_on_device(kw, kind, n, gpu_device):
    """a11 / b: the subclass of the REAL class, built by its own constructor, on a reference ``BatchMolGraph`` moved with
    the reference's own ``.to()``: equals ``Ref.forward`` on the CPU; identity constraints of cli/predict.py:256-263 hold."""
    from chemprop_amd import integration

    R = _ref()
    Hip = integration.hip_bond_message_passing_class()
    torch.manual_seed(1)
    ref_mp = R["BMP"](**kw).eval()
    torch.manual_seed(1)
    mp = Hip(**kw).eval()                      # same seed, same constructor -> same initial weights
    for (k, a), (k2, b) in zip(ref_mp.state_dict().items(), mp.state_dict().items()):
        assert k == k2 and torch.equal(a, b)
    assert isinstance(mp, R["BMP"]) and mp.hparams["cls"] is R["BMP"] and mp.W_i.in_features == ref_mp.W_i.in_features
    assert mp.output_dim == ref_mp.output_dim
    bmg = _bmg(R, n, kind, 2)
    with torch.no_grad():
        ref = ref_mp(bmg).numpy()
    mp = mp.to(gpu_device)
    V0 = bmg.V.clone()
    assert bmg.to(gpu_device) is None          # the reference's in-place move (collate.py:68-73)
    with torch.no_grad():
        for i in range(4):                      # validated batches, then the steady (replayed) path
            out = mp(bmg)
            assert parity_err(out.cpu().numpy(), ref) <= TOL, i
    assert torch.equal(bmg.V.cpu(), V0)        # the block must not mutate its input (test_regression_mol.py:217-226)
    # state dict / hparams round trip into the stock class
    sd = {k: v.cpu() for k, v in mp.state_dict().items()}
    stock = mp.hparams["cls"](**kw)
    stock.load_state_dict(sd)
    with torch.no_grad():
        bmg_c = _bmg(R, n, kind, 2)
        assert parity_err(stock.eval()(bmg_c).numpy(), ref) == 0.0


@pytest.mark.gpu
def test_real_subclass_training_gradients(gpu_device):
    from chemprop_amd import integration

    R = _ref()
    Hip = integration.hip_bond_message_passing_class()
    torch.manual_seed(3)
    ref_mp = R["BMP"](d_h=96, bias=True)
    mp = Hip(d_h=96, bias=True)
    mp.load_state_dict(ref_mp.state_dict())
    bmg = _bmg(R, 48, "qm9", 5)
    G = torch.randn(bmg.V.shape[0], 96, generator=torch.Generator().manual_seed(2))
    (ref_mp(bmg) * G).sum().backward()
    mp = mp.to(gpu_device).train()
    bmg.to(gpu_device)
    (mp(bmg) * G.to(gpu_device)).sum().backward()
    for (k, p), (_, q) in zip(mp.named_parameters(), ref_mp.named_parameters()):
        assert parity_err(p.grad.cpu().numpy(), q.grad.numpy()) <= 2e-5, k
    # a frozen encoder (cli/train.py:1826-1828) produces no gradients and still runs
    mp.zero_grad()
    mp.requires_grad_(False)
    assert torch.isfinite(mp(bmg)).all()


@pytest.mark.gpu
def test_graph_transform_on_device(gpu_device):
    """a10: a real ``GraphTransform(ScaleTransform, ScaleTransform)`` (nn/transforms.py:37-42,65-74): identity in training
    mode, ``(X - mean) / scale`` on a shallow copy in eval mode — and the caller's batch is left untouched."""
    from chemprop_amd import integration

    R = _ref()
    Hip = integration.hip_bond_message_passing_class()
    rng = np.random.default_rng(0)
    mk = lambda: R["GT"](R["ST"](rng.standard_normal(72) * 0.1, 0.5 + rng.random(72)), R["ST"](rng.standard_normal(14) * 0.1, 0.5 + rng.random(14)))
    rng = np.random.default_rng(0)
    gt_ref = mk()
    rng = np.random.default_rng(0)
    gt_hip = mk()
    vdt = lambda: R["ST"](np.linspace(-1, 1, 5), np.linspace(0.5, 2, 5))
    torch.manual_seed(4)
    ref_mp = R["BMP"](d_h=64, d_vd=5, graph_transform=gt_ref, V_d_transform=vdt())
    torch.manual_seed(4)
    mp = Hip(d_h=64, d_vd=5, graph_transform=gt_hip, V_d_transform=vdt())
    bmg = _bmg(R, 40, "qm9", 7)
    V_d = torch.randn(bmg.V.shape[0], 5, generator=torch.Generator().manual_seed(3))
    refs = {}
    for mode in ("eval", "train"):
        getattr(ref_mp, mode)()
        with torch.no_grad():
            refs[mode] = ref_mp(bmg, V_d).numpy()
    assert np.abs(refs["eval"] - refs["train"]).max() > 1e-3        # the transform does something
    mp = mp.to(gpu_device)
    V0, E0 = bmg.V.clone(), bmg.E.clone()
    bmg.to(gpu_device)
    for mode in ("eval", "train", "eval"):
        getattr(mp, mode)()
        with torch.no_grad():
            for i in range(3):
                assert parity_err(mp(bmg, V_d.to(gpu_device)).cpu().numpy(), refs[mode]) <= TOL, (mode, i)
    assert torch.equal(bmg.V.cpu(), V0) and torch.equal(bmg.E.cpu(), E0)


@pytest.mark.gpu
@pytest.mark.parametrize("shared", [False, True])
def test_multicomponent_message_passing_on_device(shared, gpu_device):
    """a11: ``MulticomponentMessagePassing.forward`` (multi.py:65-84) over two blocks — reaction + solvent style: a CGR
    component and a molecule component (tests/data/regression/rxn+mol) — after ``accelerate`` swapped the blocks."""
    from chemprop_amd import integration

    R = _ref()
    torch.manual_seed(6)
    if shared:
        blocks = [R["BMP"](d_h=64)]
        dims = [dict(), dict()]
    else:
        blocks = [R["BMP"](d_v=106, d_e=28, d_h=96), R["BMP"](d_h=64, depth=2)]
    multi = R["Multi"](blocks, n_components=2, shared=shared).eval()
    kinds = ("qm9", "qm9") if shared else ("cgr", "qm9")
    bmgs = [_bmg(R, 24, k, 8 + i) for i, k in enumerate(kinds)]
    with torch.no_grad():
        refs = [h.numpy() for h in multi(bmgs)]
    hip = copy.deepcopy(multi)
    n = integration.accelerate(hip)
    assert n == (1 if shared else 2) and hip.output_dim == multi.output_dim
    assert all(isinstance(b, R["BMP"]) and type(b) is not R["BMP"] for b in hip.blocks)
    hip = hip.to(gpu_device)
    for b in bmgs:
        b.to(gpu_device)
    with torch.no_grad():
        for i in range(3):
            outs = hip(bmgs)
            assert len(outs) == 2
            for o, r in zip(outs, refs):
                assert parity_err(o.cpu().numpy(), r) <= TOL, i


@pytest.mark.gpu
def test_mpnn_fingerprint_with_accelerated_blocks(gpu_device):
    """``MPNN.fingerprint`` (models/model.py:126-134): block -> aggregation -> batch norm, every piece the reference's own
    object, the block and the aggregation swapped in place by ``accelerate``."""
    from chemprop_amd import integration

    R = _ref()
    cnn = R["nn"]
    torch.manual_seed(9)
    model = R["MPNN"](R["BMP"](), cnn.NormAggregation(), cnn.RegressionFFN(), batch_norm=True).eval()
    bmg = _bmg(R, 100, "qm9", 11)
    with torch.no_grad():
        ref_fp = model.fingerprint(bmg).numpy()
        ref_y = model(bmg).numpy()
    keys = list(model.state_dict().keys())
    assert integration.accelerate(model) >= 2
    assert list(model.state_dict().keys()) == keys
    model = model.to(gpu_device)
    bmg.to(gpu_device)
    with torch.no_grad():
        for i in range(3):
            assert parity_err(model.fingerprint(bmg).cpu().numpy(), ref_fp) <= TOL, i
            assert parity_err(model(bmg).cpu().numpy(), ref_y) <= TOL, i


@pytest.mark.gpu
def test_deepcopy_and_device_round_trip_drop_the_engine_caches(gpu_device):
    """ADVICE (medium): the replayed argument block / pre-split weights must never outlive the tensors they point into."""
    from chemprop_amd import integration, nn as hnn

    R = _ref()
    Hip = integration.hip_bond_message_passing_class()
    torch.manual_seed(12)
    mp = Hip(d_h=64).eval().to(gpu_device)
    bmg = _bmg(R, 30, "qm9", 13)
    ref_mp = R["BMP"](d_h=64).eval()
    ref_mp.load_state_dict({k: v.cpu() for k, v in mp.state_dict().items()})
    with torch.no_grad():
        ref = ref_mp(bmg).numpy()
    bmg.to(gpu_device)
    with torch.no_grad():
        for _ in range(4):
            mp(bmg)
        assert mp.__dict__.get("_dmpnn_replay") is not None
        twin = copy.deepcopy(mp)
        assert twin.__dict__.get("_dmpnn_replay") is None and twin.__dict__.get("_dmpnn_wcache") is None
        with torch.no_grad():
            twin.W_h.weight.mul_(2.0)          # the copy's weights change: the copy's output must follow them, not the original's
        out_twin = twin(bmg)
        assert parity_err(mp(bmg).cpu().numpy(), ref) <= TOL
        ref_mp.W_h.weight.data.mul_(2.0)
        bc = _bmg(R, 30, "qm9", 13)
        assert parity_err(out_twin.cpu().numpy(), ref_mp(bc).numpy()) <= TOL
        ref_mp.W_h.weight.data.div_(2.0)
        mp = mp.cpu().to(gpu_device)            # device round trip: new storage behind the same Parameter objects
        assert mp.__dict__.get("_dmpnn_replay") is None
        assert parity_err(mp(bmg).cpu().numpy(), ref) <= TOL
        # a write through .data bumps no version: the documented hook
        for _ in range(3):
            mp(bmg)
        mp.W_o.bias.data.add_(1.0)
        hnn.invalidate(mp)
        ref_mp.W_o.bias.data.add_(1.0)
        assert parity_err(mp(bmg).cpu().numpy(), ref_mp(bc).numpy()) <= TOL


# ------------------------------------------------------------------------------------------------
# round 4: the fused training step behind the reference's OWN MPNN (models/model.py:148-161,208-231)
# ------------------------------------------------------------------------------------------------
def _golden_model_cases():
    import glob
    import os

    from conftest import GOLDEN_DIR

    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "model", "*.npz")))


def _build_hip_mpnn(R, cfg, stock=False):
    """tests/golden/make_golden_model.py: build(), with HipMPNN in place of the reference's MPNN (``stock``: the reference's own)."""
    from chemprop_amd import integration

    cnn = R["nn"]
    HipMPNN = R["MPNN"] if stock else integration.hip_mpnn_class()[1]
    agg = dict(norm=cnn.NormAggregation, mean=cnn.MeanAggregation, sum=cnn.SumAggregation)[cfg["agg"]]()
    mp = R["BMP"](**cfg["mp"])
    crit = None
    tw, kind = cfg.get("task_weights"), cfg.get("criterion", "mse")
    if kind != "mse" or tw is not None:
        crit = {"mse": cnn.MSE, "mae": cnn.MAE, "bounded-mse": cnn.BoundedMSE}[kind](task_weights=tw if tw is not None else 1.0)
    pred = cnn.RegressionFFN(input_dim=mp.output_dim, criterion=crit, **cfg["ffn"])
    return HipMPNN(mp, agg, pred, batch_norm=cfg["bn"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", _golden_model_cases(), ids=lambda p: p.split("/")[-1][:-4])
def test_hip_mpnn_training_step_is_the_reference_s(path, gpu_device):
    """``HipMPNN(chemprop.models.MPNN)``: the reference's LightningModule, built by its own constructor from the reference's own
    sub-modules, driven through ITS ``training_step(batch, batch_idx)`` with a reference ``TrainingBatch`` tuple on the device.
    Losses of both steps and the parameters / batch-norm buffers after two Adam steps against the goldens frozen from the executed
    reference's ``training_step`` + ``torch.optim.Adam``; the step ran as ONE ``dmpnn_train_step`` call (route ``fused:*``);
    ``isinstance`` / hparams / state-dict identity with the stock class hold; the trained state loads into the stock class."""
    import json

    from chemprop_amd import integration

    R = _ref()
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    cfg = meta["cfg"]
    t = lambda k: torch.from_numpy(np.array(z[k]))
    torch.manual_seed(meta["seed"])
    model = _build_hip_mpnn(R, cfg)
    assert isinstance(model, R["MPNN"]) and model.automatic_optimization is False
    assert type(model.message_passing) is integration.hip_bond_message_passing_class()
    w0 = {k[3:]: t(k) for k in z.files if k.startswith("w0.")}
    missing = model.load_state_dict(w0, strict=False)
    assert all(k.startswith("metrics.") for k in missing.missing_keys) and not missing.unexpected_keys
    model.init_lr = meta["lr"]          # (a bare loop: no trainer attached, the step takes init_lr — the goldens' constant rate)
    model = model.to(gpu_device).train()
    bmg = R["BMG"](__import__("chemprop_amd").synth.random_molgraphs(meta["n_mols"], cfg["kind"], seed=meta["seed"]))
    assert torch.equal(bmg.V, t("V")) and torch.equal(bmg.edge_index, t("edge_index"))
    bmg.to(gpu_device)
    mv = lambda k: t(k).to(gpu_device)
    batch = (bmg, None, None, mv("targets"), mv("weights"), mv("lt_mask"), mv("gt_mask"))
    for s in range(meta["steps"]):
        loss = model.training_step(batch, s)
        torch.cuda.synchronize()
        ref = float(z[f"loss{s}"])
        assert abs(float(loss) - ref) <= (1e-5 if s == 0 else 5e-4) * max(1.0, abs(ref)), (s, float(loss), ref)
        assert model.__dict__["_hip"]["route"].startswith("fused:"), model.__dict__["_hip"]
    for k, v in model.state_dict().items():
        if k.startswith("metrics."):
            continue
        want = z["w2." + k]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(want)
        elif "task_weights" in k:
            assert np.array_equal(v.cpu().numpy().reshape(-1), want.reshape(-1))
        else:
            # (after TWO Adam steps: the update g / (|g| + eps) amplifies 1e-7 differences of near-zero gradients — the functional bar
            #  of test_model.py::test_fused_step_matches_goldens)
            assert parity_err(v.cpu().numpy(), want) <= 2e-4, k
    # the trained state moves into the STOCK class (same keys), which predicts like the golden's trained reference
    stock = _build_hip_mpnn(R, cfg, stock=True)
    assert type(stock.message_passing) is R["BMP"]
    stock.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, strict=False)
    stock.eval()
    with torch.no_grad():
        pe = stock(R["BMG"](__import__("chemprop_amd").synth.random_molgraphs(meta["n_mols"], cfg["kind"], seed=meta["seed"])))
    assert parity_err(pe.numpy(), z["preds_eval"]) <= 2e-3


@pytest.mark.gpu
def test_hip_mpnn_falls_back_to_the_module_path_and_follows_the_schedule(gpu_device):
    """What the fused step refuses (here: ``V_d`` descriptors with a ``W_d`` branch) trains through the reference's own
    ``training_step`` arithmetic on the HIP kernels, same flat Adam; the learning rate of every step is the reference's Noam-like
    schedule's (``schedulers.py``), read from the optimizer Lightning would hold; the flat Adam's moments travel in the checkpoint."""
    from chemprop_amd import integration, synth

    R = _ref()
    cnn = R["nn"]
    HipMPNN = integration.hip_mpnn_class()[1]
    torch.manual_seed(3)
    model = HipMPNN(R["BMP"](d_h=64, d_vd=4), cnn.MeanAggregation(), cnn.RegressionFFN(input_dim=68, hidden_dim=32), batch_norm=True)
    model = model.to(gpu_device).train()
    # what Lightning's trainer would provide in manual optimization: the reference's own configure_optimizers() objects
    from chemprop.schedulers import build_NoamLike_LRSched

    topt = torch.optim.Adam(model.parameters(), model.init_lr)
    sched = build_NoamLike_LRSched(topt, 2, 4, 1e-4, 1e-3, 1e-4)
    model.optimizers = lambda: topt
    model.lr_schedulers = lambda: sched
    bmg = R["BMG"](synth.random_molgraphs(32, "qm9", seed=5))
    bmg.to(gpu_device)
    y = torch.randn(32, 1, device=gpu_device)
    w = torch.ones(32, 1, device=gpu_device)
    no = torch.zeros(32, 1, dtype=torch.bool, device=gpu_device)
    V_d = torch.randn(int(bmg.V.shape[0]), 4, device=gpu_device)
    lrs, seen = [], []
    for s in range(4):
        st = model._hip_state()
        before = st["opt"].steps
        lrs.append(topt.param_groups[0]["lr"])
        loss = model.training_step((bmg, V_d, None, y, w, no, no), s)
        seen.append(st["route"])
        assert st["opt"].steps == before + 1 and torch.isfinite(loss)
    # (the block has a W_d branch and the batch carries V_d: FusedTrainer refuses the model, every step is the module path — on the
    #  flat Adam)
    assert all(r == "module" for r in seen), seen
    assert "V_d" in (model._hip_state()["why"] or "")
    want = [1e-4, 1e-4 + (1e-3 - 1e-4) / 2, 1e-3, 1e-3 * (1e-4 / 1e-3) ** (1 / 4)]
    assert np.allclose(lrs, want, rtol=1e-6), (lrs, want)
    ck = {}
    model.on_save_checkpoint(ck)
    assert int(ck["hip_flat_adam"]["step"]) == 4
    model.on_load_checkpoint(ck)
    assert model._hip_state()["opt"].steps == 4
