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

from conftest import TOL, adam_comparable, parity_err, parity_err_where
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
        crit = {"mse": cnn.MSE, "mae": cnn.MAE, "bounded-mse": cnn.BoundedMSE, "bce": cnn.BCELoss, "ce": cnn.CrossEntropyLoss,
                "mve": cnn.MVELoss, "evidential": cnn.EvidentialLoss, "quantile": cnn.QuantileLoss}[kind](task_weights=tw if tw is not None else 1.0)
    FFN = {"classification": cnn.BinaryClassificationFFN, "multiclass": cnn.MulticlassClassificationFFN, "mve": cnn.MveFFN,
           "evidential": cnn.EvidentialFFN, "quantile": cnn.QuantileFFN}.get(cfg.get("predictor"), cnn.RegressionFFN)
    pred = FFN(input_dim=mp.output_dim, criterion=crit, **cfg["ffn"])
    return HipMPNN(mp, agg, pred, batch_norm=cfg["bn"])


def _fit_one_epoch(R, model, batches, ckpt_path=None, **trainer_kw):
    """``Trainer.fit`` of the Lightning stand-in (``oracle/lightning_shim.py``: automatic optimization, the closure inside
    ``optimizer.step``) for one epoch over ``batches``; returns the trainer and the per-step losses / routes."""
    import lightning.pytorch as pl

    tr = pl.Trainer(max_epochs=1, **trainer_kw)
    losses, routes = [], []
    end = model.on_train_batch_end

    def on_end(out, b, i):
        losses.append(float(out["loss"]))
        routes.append((model.__dict__.get("_hip") or {}).get("route"))
        return end(out, b, i)

    model.on_train_batch_end = on_end
    tr.fit(model, batches, None, ckpt_path=ckpt_path)
    model.on_train_batch_end = end
    torch.cuda.synchronize()
    return tr, losses, routes


@pytest.mark.gpu
@pytest.mark.parametrize("path", _golden_model_cases(), ids=lambda p: p.split("/")[-1][:-4])
def test_hip_mpnn_training_step_is_the_reference_s(path, gpu_device, tmp_path):
    """``HipMPNN(chemprop.models.MPNN)``: the reference's LightningModule, built by its own constructor from the reference's own
    sub-modules, driven by ``Trainer.fit`` (the Lightning stand-in: AUTOMATIC optimization, as ``chemprop train`` runs it) over
    reference ``TrainingBatch`` tuples on the device, at the goldens' constant learning rate (``init_lr = max_lr = final_lr``: the
    reference's own schedule is then flat).  Losses of both steps and the parameters / batch-norm buffers after two Adam steps against
    the goldens frozen from the executed reference's ``training_step`` + ``torch.optim.Adam``; every step ran as ONE
    ``dmpnn_train_step`` call (route ``fused:*``) and counted as one optimizer step of the trainer; ``isinstance`` / hparams /
    state-dict identity with the stock class hold; the trained state loads into the stock class.  Then STEP 2 BY ITSELF (round-4
    VERDICT weak #3): resumed — through ``Trainer.fit(ckpt_path=...)``, i.e. ``HipAdam.load_state_dict`` on ``torch.optim.Adam``'s own
    format — from the golden's post-step-1 parameters, buffers and moments, it reproduces the golden's second loss at 1e-5."""
    import json

    from chemprop_amd import integration

    R = _ref()
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    cfg = meta["cfg"]
    t = lambda k: torch.from_numpy(np.array(z[k]))

    def fresh(prefix):
        torch.manual_seed(meta["seed"])
        m = _build_hip_mpnn(R, cfg)
        w = {k[len(prefix):]: t(k) for k in z.files if k.startswith(prefix)}
        missing = m.load_state_dict(w, strict=False)
        assert all(k.startswith("metrics.") for k in missing.missing_keys) and not missing.unexpected_keys
        m.init_lr = m.max_lr = m.final_lr = meta["lr"]
        return m.to(gpu_device).train()

    model = fresh("w0.")
    assert isinstance(model, R["MPNN"]) and model.automatic_optimization is True
    assert type(model.message_passing) is integration.hip_bond_message_passing_class()
    bmg = R["BMG"](__import__("chemprop_amd").synth.random_molgraphs(meta["n_mols"], cfg["kind"], seed=meta["seed"]))
    assert torch.equal(bmg.V, t("V")) and torch.equal(bmg.edge_index, t("edge_index"))
    bmg.to(gpu_device)
    mv = lambda k: t(k).to(gpu_device)
    batch = (bmg, None, None, mv("targets"), mv("weights"), mv("lt_mask"), mv("gt_mask"))
    tr, losses, routes = _fit_one_epoch(R, model, [batch] * meta["steps"])
    assert tr.global_step == meta["steps"] and all(r.startswith("fused:") for r in routes), routes
    for s in range(meta["steps"]):
        ref = float(z[f"loss{s}"])
        assert abs(losses[s] - ref) <= (1e-5 if s == 0 else 5e-4) * max(1.0, abs(ref)), (s, losses[s], ref)
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
            assert parity_err_where(v.cpu().numpy(), want, adam_comparable(z, k, meta["steps"])) <= 2e-4, k
    # the trained state moves into the STOCK class (same keys), which predicts like the golden's trained reference
    stock = _build_hip_mpnn(R, cfg, stock=True)
    assert type(stock.message_passing) is R["BMP"]
    stock.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, strict=False)
    stock.eval()
    with torch.no_grad():
        pe = stock(R["BMG"](__import__("chemprop_amd").synth.random_molgraphs(meta["n_mols"], cfg["kind"], seed=meta["seed"])))
    assert parity_err(pe.numpy(), z["preds_eval"]) <= 2e-3

    # ---- step 2 by itself, from the golden's own state after step 1 (a checkpoint in the stock trainer's format) ----
    model2 = fresh("w1.")
    names = [k for k, _ in model2.named_parameters()]
    ost = {"state": {i: {"step": torch.tensor(1.0), "exp_avg": t("m1." + k), "exp_avg_sq": t("v1." + k)} for i, k in enumerate(names)},
           "param_groups": [dict(lr=meta["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, params=list(range(len(names))))]}
    ck = {"epoch": -1, "global_step": 1, "state_dict": {k: v.detach().cpu() for k, v in model2.state_dict().items()},
          "optimizer_states": [ost], "lr_schedulers": [], "loops": {"auto_step": 1, "manual_step": 0, "epochs_done": 0}, "hyper_parameters": {}}
    torch.save(ck, tmp_path / "after_step1.ckpt")
    seen = {}
    # (automatic optimization zeroes the gradients AFTER training_step, inside the same closure: look at them in the hook before that)
    model2.on_before_zero_grad = lambda opt: seen.update(flat=model2.__dict__["_hip"]["sync"].flat.detach().clone())
    tr2, losses2, routes2 = _fit_one_epoch(R, model2, [batch], ckpt_path=str(tmp_path / "after_step1.ckpt"))
    ref = float(z["loss1"])
    assert routes2[0].startswith("fused:") and tr2.global_step == 2 and model2.__dict__["_hip"]["opt"].steps == 2
    assert abs(losses2[0] - ref) <= 1e-5 * max(1.0, abs(ref)), (losses2[0], ref)
    sync = model2.__dict__["_hip"]["sync"]
    for p, o, k in zip(sync.params, sync.offsets, [k for k, p in model2.named_parameters() if p.requires_grad]):
        got = seen["flat"][o:o + p.numel()].view_as(p).cpu().numpy()
        assert parity_err(got, z["g1." + k]) <= 2e-5, k


@pytest.mark.gpu
def test_hip_mpnn_without_a_trainer_is_the_reference_s_training_step(gpu_device):
    """No ``Trainer`` attached: ``training_step`` is the reference's — a loss WITH a graph, nothing updated, gradients by autograd
    through the HIP kernels of the swapped blocks (what a bare loop with its own optimizer expects)."""
    from chemprop_amd import integration, synth

    R = _ref()
    cnn = R["nn"]
    HipMPNN = integration.hip_mpnn_class()[1]
    torch.manual_seed(3)
    mk = lambda cls: cls(R["BMP"](d_h=64), cnn.MeanAggregation(), cnn.RegressionFFN(input_dim=64, hidden_dim=32), batch_norm=True)
    model = mk(HipMPNN)
    twin = mk(R["MPNN"])
    twin.load_state_dict(model.state_dict(), strict=False)
    model = model.to(gpu_device).train()
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    gen = torch.Generator().manual_seed(1)
    y, w = torch.randn(32, 1, generator=gen), torch.ones(32, 1)
    no = torch.zeros(32, 1, dtype=torch.bool)
    bmg = R["BMG"](synth.random_molgraphs(32, "qm9", seed=5))
    bmg.to(gpu_device)
    loss = model.training_step((bmg, None, None, y.to(gpu_device), w.to(gpu_device), no.to(gpu_device), no.to(gpu_device)), 0)
    assert loss.requires_grad and model.__dict__["_hip"] is None
    loss.backward()
    ref = twin.train().training_step((R["BMG"](synth.random_molgraphs(32, "qm9", seed=5)), None, None, y, w, no, no), 0)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    for (k, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        assert torch.equal(p.detach(), before[k])
        assert parity_err(p.grad.cpu().numpy(), q.grad.numpy()) <= 2e-5, k
