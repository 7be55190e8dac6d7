"""f4 — the whole training step (``models/model.py:148-161`` + ``torch.optim.Adam``): the restatement against goldens frozen from
the executed reference (CPU), the fused ``dmpnn_train_step`` against goldens, restatement, module path and — where the staged
reference travelled — the reference's own ``MPNN.training_step`` executed live (GPU)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, adam_comparable, parity_err, parity_err_where
from oracle import model_torch as om
from oracle import ref_shim

CASES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "model", "*.npz")))


class G:
    def __init__(self, path):
        z = np.load(path)
        self.a = {k: z[k] for k in z.files}
        self.meta = json.loads(bytes(self.a.pop("meta")).decode())
        self.cfg = self.meta["cfg"]
        self.name = self.meta["name"]

    def t(self, k):
        return torch.from_numpy(np.array(self.a[k]))

    def state(self, prefix):
        return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in self.a.items() if k.startswith(prefix)}

    def bmg(self, device="cpu"):
        from chemprop_amd.data import BatchMolGraph

        b = BatchMolGraph.from_tensors(self.t("V"), self.t("E"), self.t("edge_index"), self.t("rev_edge_index"), self.t("batch"), self.meta["n_mols"])
        if device != "cpu":
            b.to(device)
        return b

    def batch_args(self, device="cpu"):
        bounded = self.cfg.get("criterion", "mse").startswith("bounded")
        mv = lambda x: x.to(device)
        return (mv(self.t("targets")), mv(self.t("weights")), mv(self.t("lt_mask")) if bounded else None, mv(self.t("gt_mask")) if bounded else None)


@pytest.fixture(params=CASES, ids=[os.path.basename(c)[:-4] for c in CASES])
def g(request):
    return G(request.param)


def test_goldens_exist():
    assert len(CASES) >= 9


def test_restatement_matches_executed_reference(g):
    """oracle/model_torch.py == the reference's own training_step + Adam: losses, every gradient of both steps, the parameters
    and batch-norm buffers after the second step."""
    targets, weights, lt, gt = g.batch_args()
    m, losses, grads = om.train_steps(g.state("w0."), g.cfg, g.bmg(), targets, weights, g.t("lt_mask"), g.t("gt_mask"), g.meta["lr"], g.meta["steps"])
    for s in range(g.meta["steps"]):
        assert abs(losses[s] - float(g.a[f"loss{s}"])) <= 1e-6 * max(1.0, abs(losses[s])), (s, losses[s], float(g.a[f"loss{s}"]))
        for k, v in grads[s].items():
            assert parity_err(v.numpy(), g.a[f"g{s}.{k}"]) <= 1e-6, (s, k)
    for k, v in m.state().items():
        ref = g.a["w2." + k]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref)
        elif k.endswith("task_weights"):
            assert np.array_equal(v.numpy().reshape(-1), ref.reshape(-1))
        else:
            assert parity_err_where(v.numpy(), ref, adam_comparable(g.a, k, g.meta["steps"])) <= 2e-6, k


def test_module_criterion_is_the_reference_formula():
    """chemprop_amd.model.masked_loss (the module path's criterion, torch ops) == the restated criterion, all variants."""
    from chemprop_amd.model import masked_loss

    gen = torch.Generator().manual_seed(0)
    P, T = torch.randn(30, 4, generator=gen), torch.randn(30, 4, generator=gen)
    T[torch.rand(30, 4, generator=gen) < 0.2] = float("nan")
    w, tw = torch.rand(30, 1, generator=gen) + 0.5, torch.tensor([1.0, 0.3, 2.0, 1.5])
    lt, gt = torch.rand(30, 4, generator=gen) < 0.3, torch.rand(30, 4, generator=gen) < 0.3
    for kind, bounded in (("mse", False), ("mae", False), ("mse", True), ("mae", True)):
        a = masked_loss(P, T, w, tw, lt if bounded else None, gt if bounded else None, kind)
        b = om.criterion(P, T, w, tw, lt, gt, ("bounded-" if bounded else "") + kind)
        assert abs(float(a) - float(b)) <= 1e-7 * max(1.0, abs(float(b)))


def build_mirror(cfg):
    """The mirror model of a golden's configuration, constructed in the reference's order (same RNG stream)."""
    from chemprop_amd import agg as cagg
    from chemprop_amd.model import (BCE, CE, MAE, MPNN, MSE, MVE, BinaryClassificationFFN, Evidential, EvidentialFFN, MulticlassClassificationFFN, MveFFN,
                                    Quantile, QuantileFFN, RegressionFFN)
    from chemprop_amd.nn import BondMessagePassing

    mp = BondMessagePassing(**cfg["mp"])
    agg = dict(norm=cagg.NormAggregation, mean=cagg.MeanAggregation, sum=cagg.SumAggregation)[cfg["agg"]]()
    kind = cfg.get("criterion", "mse")
    t = cfg["ffn"]["n_tasks"]
    crit = None
    if kind != "mse" or cfg.get("task_weights") is not None:   # (an explicit criterion: task_weights as given, 1.0 -> shape [1, 1], broadcast)
        crit = ({"ce": CE, "bce": BCE, "mve": MVE, "evidential": Evidential, "quantile": Quantile}.get(kind) or (MAE if kind.endswith("mae") else MSE))(cfg.get("task_weights") or 1.0)
    FFN = {"classification": BinaryClassificationFFN, "multiclass": MulticlassClassificationFFN, "mve": MveFFN,
           "evidential": EvidentialFFN, "quantile": QuantileFFN}.get(cfg.get("predictor"), RegressionFFN)
    pred = FFN(input_dim=mp.output_dim, criterion=crit, **cfg["ffn"])
    return MPNN(mp, agg, pred, batch_norm=cfg["bn"])


def test_mirror_state_dict_is_the_reference_s(g):
    """Same keys and shapes as the reference's MPNN state dict (the bookkeeping ``metrics.*`` buffers aside), same initial
    weights for the same seed: a checkpoint moves between the two."""
    torch.manual_seed(g.meta["seed"])
    model = build_mirror(g.cfg)
    ref = g.state("w0.")
    mine = model.state_dict()
    assert set(mine.keys()) == set(ref.keys()), set(mine.keys()) ^ set(ref.keys())
    for k, v in ref.items():
        assert tuple(mine[k].shape) == tuple(v.shape), k
        if "task_weights" not in k and "num_batches" not in k:
            assert torch.equal(mine[k], v), k
    model.load_state_dict(ref)


# ------------------------------------------------------------------------------------------------
# GPU: the fused step
# ------------------------------------------------------------------------------------------------
def _adam_reference(params0, grads_per_step, lr):
    """torch.optim.Adam on the CPU driven by GIVEN gradients (checks the fused update's arithmetic on its own)."""
    ps = [p.clone().requires_grad_(True) for p in params0]
    opt = torch.optim.Adam(ps, lr)
    for gs in grads_per_step:
        for p, gr in zip(ps, gs):
            p.grad = gr.clone()
        opt.step()
    return [p.detach() for p in ps]


@pytest.mark.gpu
@pytest.mark.parametrize("keep_rows", ["auto", "1"], ids=["block-products", "split-row-products"])
def test_fused_step_matches_goldens(g, keep_rows, gpu_device, monkeypatch):
    from chemprop_amd.model import FusedTrainer

    # ("1": the messages kept as split rows and EVERY weight gradient — the predictor's riding first layer included — on the split-row
    #  product k_wgrad16r, which batches of this size do not take by themselves)
    monkeypatch.setenv("DMPNN_KEEP_ROWS", keep_rows)

    model = build_mirror(g.cfg)
    model.load_state_dict(g.state("w0."))
    model = model.to(gpu_device).train()
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    p0 = [p.detach().cpu().clone() for _, p in model.named_parameters() if p.requires_grad]
    tr = FusedTrainer(model, lr=g.meta["lr"])
    bmg = g.bmg(gpu_device)
    targets, weights, lt, gt = g.batch_args(gpu_device)
    bounded = g.cfg.get("criterion", "mse").startswith("bounded")
    my_grads = []
    for s in range(g.meta["steps"]):
        out = tr.step(bmg, targets, weights, lt if bounded else None, gt if bounded else None)
        torch.cuda.synchronize()
        loss, n_fin = float(out[0]), float(out[1])
        ref = float(g.a[f"loss{s}"])
        # step 0: same parameters as the reference -> the fp32 bar; step 1: parameters after ONE Adam step, whose update
        # g / (|g| + eps) amplifies 1e-7 differences of near-zero gradients -> a functional bar
        assert abs(loss - ref) <= (1e-5 if s == 0 else 5e-4) * max(1.0, abs(ref)), (s, loss, ref)
        assert n_fin == float(np.isfinite(g.a["targets"]).sum())
        grads = [tr.sync.views[i].detach().cpu().clone() for i in range(len(tr.sync.params))]
        my_grads.append(grads)
        if s == 0:
            by_name = dict(zip(names, grads))
            for k, v in by_name.items():
                err = parity_err(v.numpy(), g.a[f"g0.{k}"])
                assert err <= 2e-5, f"{g.name} d{k}: {err:.3e}"
    # the update itself: Adam on the CPU fed with the gradients the kernels produced
    want = _adam_reference(p0, my_grads, g.meta["lr"])
    for k, p, w in zip(names, [p for _, p in model.named_parameters() if p.requires_grad], want):
        assert parity_err(p.detach().cpu().numpy(), w.numpy()) <= 2e-6, k
    if g.cfg["bn"]:
        for k in ("running_mean", "running_var"):
            assert parity_err(getattr(model.bn, k).cpu().numpy(), g.a[f"w2.bn.{k}"]) <= 2e-5, k
        assert int(model.bn.num_batches_tracked) == g.meta["steps"]
    # and the trained model predicts like the reference's trained model (eval: running statistics, the inference kernels)
    model.eval()
    with torch.no_grad():
        pe = model(bmg)
    assert parity_err(pe.cpu().numpy(), g.a["preds_eval"]) <= 2e-3


def _load_step1_state(g, model, adam):
    """Parameters, batch-norm buffers and Adam's moments after the golden's FIRST step (``w1.*`` / ``m1.*`` / ``v1.*``): where its second
    step starts."""
    model.load_state_dict({k: v for k, v in g.state("w1.").items()}, strict=False)
    st = {}
    for i, (k, p) in enumerate((k, p) for k, p in model.named_parameters() if p.requires_grad):
        st[i] = {"step": torch.tensor(1.0), "exp_avg": g.t("m1." + k), "exp_avg_sq": g.t("v1." + k)}
    adam.load_torch_state(st)
    assert adam.steps == 1


def test_restatement_step_two_from_the_goldens_own_step_one_state(g):
    """Step 2 by itself (round-4 VERDICT weak #3): the restatement started from the golden's post-step-1 parameters reproduces the
    golden's second loss and second gradients at the fp32 bar."""
    targets, weights, lt, gt = g.batch_args()
    st1 = {k: v for k, v in g.state("w1.").items()}
    m, losses, grads = om.train_steps(st1, g.cfg, g.bmg(), targets, weights, g.t("lt_mask"), g.t("gt_mask"), g.meta["lr"], 1)
    ref = float(g.a["loss1"])
    assert abs(losses[0] - ref) <= 1e-6 * max(1.0, abs(ref))
    for k, v in grads[0].items():
        assert parity_err(v.numpy(), g.a[f"g1.{k}"]) <= 1e-6, k


@pytest.mark.gpu
def test_fused_step_two_alone_matches_goldens_at_the_fp32_bar(g, gpu_device):
    """Step 2 pinned BY ITSELF: the fused step starts from the golden's own post-step-1 parameters, batch-norm buffers and Adam
    moments (``FlatAdam.load_torch_state``) and must reproduce the golden's second loss at 1e-5 and its second gradients at 2e-5
    (parameters after the update: the functional 2e-4) — a step-2 bug can no longer hide behind the 5e-4 / 2e-4 end-to-end bars of
    ``test_fused_step_matches_goldens`` (which exist because ONE Adam step amplifies fp32-level differences of near-zero gradients)."""
    from chemprop_amd.model import FusedTrainer

    model = build_mirror(g.cfg)
    model.load_state_dict(g.state("w0."))
    model = model.to(gpu_device).train()
    tr = FusedTrainer(model, lr=g.meta["lr"])
    _load_step1_state(g, model, tr.opt)
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    bmg = g.bmg(gpu_device)
    targets, weights, lt, gt = g.batch_args(gpu_device)
    bounded = g.cfg.get("criterion", "mse").startswith("bounded")
    out = tr.step(bmg, targets, weights, lt if bounded else None, gt if bounded else None)
    torch.cuda.synchronize()
    ref = float(g.a["loss1"])
    assert abs(float(out[0]) - ref) <= 1e-5 * max(1.0, abs(ref)), (float(out[0]), ref)
    assert tr.opt.steps == 2
    for i, k in enumerate(names):
        err = parity_err(tr.sync.views[i].detach().cpu().numpy(), g.a[f"g1.{k}"])
        assert err <= 2e-5, f"{g.name} step-2 d{k}: {err:.3e}"
    for k, v in model.state_dict().items():
        want = g.a["w2." + k]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(want)
        elif "task_weights" in k:
            continue
        else:
            # (ONE Adam update from identical moments; an entry whose gradient is below Adam's eps still turns an absolute 1e-8 into
            #  a visible fraction of lr — the bars that pin step 2 are the loss and the gradients above)
            assert parity_err_where(v.cpu().numpy(), want, adam_comparable(g.a, k, g.meta["steps"])) <= 2e-4, k


@pytest.mark.gpu
@pytest.mark.parametrize("n_mols,kind,bn,agg,tasks,act", [(512, "qm9", True, "norm", 1, "elu"), (512, "qm9", False, "mean", 12, "tanh"),
                                                          # (40-atom molecules: the per-step routes; a smooth activation — at this size ONE
                                                          #  ReLU mask flip between two fp32-class arithmetics moves a gradient row by 1e-3,
                                                          #  DESIGN.md section 5 — the engine and the module path agree to 5e-8 either way)
                                                          (128, "synth40", True, "sum", 2, "tanh"),
                                                          # (beyond the one-launch kernels of a short batch — k_out_all <= 1024 molecules,
                                                          #  k_layer_bwd <= 2048: the row-split weight gradients and the MFMA contractions)
                                                          (2304, "qm9", True, "mean", 2, "elu")])
def test_fused_step_at_size_vs_restatement_and_module_path(n_mols, kind, bn, agg, tasks, act, gpu_device):
    """The CLI's default widths (d_h 300, hidden 300) at BASELINE's batch size: loss and every gradient of the fused step against
    (a) the restatement on the CPU, (b) autograd through the module path (the same kernels driven from Python), and — where the
    staged reference is present — (c) the reference's own ``MPNN.training_step`` executed live."""
    from chemprop_amd import synth
    from chemprop_amd.model import FusedTrainer

    # (smooth activations in block and predictor: with ReLU the comparison is only as good as the masks — at 512 molecules one
    #  hidden unit of one molecule within 1e-7 of the kink, flipped by the 1e-7 differences between two fp32 batch-norm
    #  implementations, moves gradient entries by 2e-5 (measured); the default ReLU model is the golden cases' and the next test's)
    cfg = dict(mp=dict(activation=act), agg=agg, bn=bn, ffn=dict(n_tasks=tasks, activation=act))
    torch.manual_seed(17)
    model = build_mirror(cfg)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(23)
    targets = torch.randn(n_mols, tasks, generator=gen)
    if tasks > 1:
        targets[torch.rand(n_mols, tasks, generator=gen) < 0.15] = float("nan")
    weights = 0.5 + torch.rand(n_mols, 1, generator=gen)
    cpu_bmg = synth.random_batch(n_mols, kind, seed=31)
    ref_model = om.Model(state, cfg)
    ref_loss = ref_model.loss(cpu_bmg, targets, weights, None, None)
    ref_loss.backward()

    model = model.to(gpu_device).train()
    bmg = synth.random_batch(n_mols, kind, seed=31)
    bmg.to(gpu_device)
    tg, wg = targets.to(gpu_device), weights.to(gpu_device)
    # (b) module path first (it does not touch the parameters)
    loss_mod = model.loss(bmg, tg, wg)
    loss_mod.backward()
    mod_grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
    if bn:  # the module path's forward updated the running statistics once: restore, the fused step must do the same update itself
        model.bn.load_state_dict({k[3:]: v for k, v in state.items() if k.startswith("bn.")})
    model.zero_grad(set_to_none=True)
    tr = FusedTrainer(model, lr=1e-4)
    out = tr.step(bmg, tg, wg)
    torch.cuda.synchronize()
    assert abs(float(out[0]) - float(ref_loss)) <= 1e-5 * max(1.0, abs(float(ref_loss)))
    assert abs(float(loss_mod) - float(ref_loss)) <= 1e-5 * max(1.0, abs(float(ref_loss)))
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    for i, k in enumerate(names):
        gf = tr.sync.views[i].detach().cpu().numpy()
        e_ref = parity_err(gf, ref_model.p[k].grad.numpy())
        e_mod = parity_err(gf, mod_grads[k].numpy())
        bad_mod = np.argwhere(~np.isfinite(mod_grads[k].numpy()))
        assert e_ref <= 2e-5 and e_mod <= 2e-5, (f"{k}: vs restatement {e_ref:.2e}, vs module path {e_mod:.2e}; non-finite module-path entries: "
                                                 f"{len(bad_mod)} of {mod_grads[k].numel()}, first {bad_mod[:12].tolist()}, last {bad_mod[-4:].tolist()}")
    if bn:
        want = om.Model(state, cfg)
        want.loss(cpu_bmg, targets, weights, None, None)
        assert parity_err(model.bn.running_var.cpu().numpy(), want.buf["bn.running_var"].numpy()) <= 1e-5
    if ref_shim.reference_available() and tasks == 1:
        BMP, BMG, _ = ref_shim.load_reference()
        _, _, _, RefMPNN, cnn = ref_shim.load_reference_extras()
        ref = RefMPNN(BMP(activation=act), dict(norm=cnn.NormAggregation, mean=cnn.MeanAggregation, sum=cnn.SumAggregation)[agg](), cnn.RegressionFFN(n_tasks=tasks, activation=act), batch_norm=bn)
        ref.load_state_dict(state, strict=False)
        ref.train()
        rb = BMG(synth.random_molgraphs(n_mols, kind, seed=31))
        l = ref.training_step((rb, None, None, targets, weights, torch.zeros_like(targets, dtype=torch.bool), torch.zeros_like(targets, dtype=torch.bool)), 0)
        l.backward()
        assert abs(float(out[0]) - float(l)) <= 1e-5 * max(1.0, abs(float(l)))
        rg = dict(ref.named_parameters())
        for i, k in enumerate(names):
            assert parity_err(tr.sync.views[i].detach().cpu().numpy(), rg[k].grad.numpy()) <= 2e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("n_mols", [96, 512])
def test_fused_step_on_the_tile_plan_equals_the_full_plan(n_mols, gpu_device):
    """After the validated first batches the trainer plans with the tile table alone (DMPNN_F_TILE_PLAN: K0 11 us instead of 28,
    kept tensors in the caller's edge order).  Same losses and parameters as a trainer held on the full CSR plan, to summation
    order (the weight-gradient products walk the rows in another order)."""
    from chemprop_amd import _lib, agg as cagg, synth
    from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
    from chemprop_amd.nn import BondMessagePassing

    batches = [synth.random_batch(n_mols, "qm9", seed=300 + i) for i in range(5)]
    for b in batches:
        b.to(gpu_device)
    ys = [torch.randn(n_mols, 1, generator=torch.Generator().manual_seed(i)).to(gpu_device) for i in range(5)]

    def run(tile_plan):
        torch.manual_seed(5)
        m = MPNN(BondMessagePassing(d_h=300, activation="elu"), cagg.NormAggregation(), RegressionFFN(n_tasks=1, input_dim=300, activation="elu"),
                 batch_norm=True).to(gpu_device).train()
        tr = FusedTrainer(m, lr=1e-3, tile_plan=tile_plan)
        losses, kinds = [], []
        for i in range(5):
            losses.append(tr.step(batches[i], ys[i]))
            kinds.append(bool(tr._last_plan_tiles))
        torch.cuda.synchronize()
        return torch.stack(losses).cpu()[:, 0], tr.opt.flat.detach().cpu().clone(), kinds

    l_t, p_t, k_t = run(True)
    l_f, p_f, k_f = run(False)
    assert k_f == [False] * 5 and k_t[:2] == [False, False] and all(k_t[2:]), (k_t, k_f)
    assert torch.allclose(l_t, l_f, rtol=2e-5, atol=1e-6), (l_t, l_f)
    assert parity_err(p_t.numpy(), p_f.numpy()) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n_mols", [64, 512])
def test_fused_step_with_block_dropout_equals_module_path_on_the_same_masks(n_mols, gpu_device):
    """CLI ``--dropout`` on the block (base.py:139,182): the fused step draws ONE seed from torch's CPU generator like the module
    path's fused dropout (autograd.py), so after the same ``torch.manual_seed`` both see the same hash masks — loss and every
    gradient agree (the module path itself is checked against the executed reference given those masks: tests/test_dropout_gpu.py);
    and the masks are live (the loss differs from the p = 0 step, the seed changes from step to step)."""
    from chemprop_amd import agg as cagg, synth
    from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
    from chemprop_amd.nn import BondMessagePassing

    bmg = synth.random_batch(n_mols, "qm9", seed=77)
    bmg.to(gpu_device)
    y = torch.randn(n_mols, 1, generator=torch.Generator().manual_seed(1)).to(gpu_device)

    def make(p):
        torch.manual_seed(9)
        return MPNN(BondMessagePassing(d_h=300, dropout=p), cagg.MeanAggregation(), RegressionFFN(n_tasks=1, input_dim=300), batch_norm=False).to(gpu_device).train()

    m_mod = make(0.25)
    torch.manual_seed(123)
    loss_mod = m_mod.loss(bmg, y)
    loss_mod.backward()
    m_fus = make(0.25)
    tr = FusedTrainer(m_fus, lr=1e-4)
    torch.manual_seed(123)
    out = tr.step(bmg, y)
    torch.cuda.synchronize()
    seed0 = tr.last_dropout_seed
    assert abs(float(out[0]) - float(loss_mod)) <= 1e-5 * max(1.0, abs(float(loss_mod)))
    names = [k for k, p in m_fus.named_parameters() if p.requires_grad]
    mod = dict(m_mod.named_parameters())
    for i, k in enumerate(names):
        e = parity_err(tr.sync.views[i].detach().cpu().numpy(), mod[k].grad.detach().cpu().numpy())
        assert e <= 2e-5, f"{k}: {e:.2e}"
    # live masks: another seed, another loss; p = 0 gives yet another
    out2 = tr.step(bmg, y)
    torch.cuda.synchronize()
    assert tr.last_dropout_seed != seed0
    m0 = make(0.0)
    l0 = FusedTrainer(m0, lr=1e-4).step(bmg, y)
    torch.cuda.synchronize()
    assert abs(float(l0[0]) - float(out[0])) > 1e-4 * max(1.0, abs(float(l0[0])))
    # eval: dropout is the identity (nn.Dropout in eval mode), the trainer refuses nothing and the module predicts deterministically
    m_fus.eval()
    with torch.no_grad():
        a, b = m_fus(bmg), m_fus(bmg)
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_default_relu_model_at_size_fused_equals_module_path(gpu_device):
    """The CLI's default model (ReLU everywhere, d_h 300, norm aggregation, batch norm) at 512 molecules: loss to 1e-5 and gradients
    against the restatement to the kink-aware bar 2e-4 (a flipped ReLU mask of ONE predictor unit moves entries by ~2e-5, see above),
    and the fused step against the module path, which drive the same block kernels, to the same bar."""
    from chemprop_amd import synth
    from chemprop_amd.model import FusedTrainer

    cfg = dict(mp=dict(), agg="norm", bn=True, ffn=dict(n_tasks=1))
    torch.manual_seed(17)
    model = build_mirror(cfg)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(23)
    targets, weights = torch.randn(512, 1, generator=gen), 0.5 + torch.rand(512, 1, generator=gen)
    ref_model = om.Model(state, cfg)
    ref_loss = ref_model.loss(synth.random_batch(512, "qm9", seed=31), targets, weights, None, None)
    ref_loss.backward()
    model = model.to(gpu_device).train()
    bmg = synth.random_batch(512, "qm9", seed=31)
    bmg.to(gpu_device)
    tr = FusedTrainer(model, lr=1e-4)
    out = tr.step(bmg, targets.to(gpu_device), weights.to(gpu_device))
    torch.cuda.synchronize()
    assert abs(float(out[0]) - float(ref_loss.detach())) <= 1e-5 * max(1.0, abs(float(ref_loss.detach())))
    assert tr.last_route == "mega16"
    for i, k in enumerate([k for k, p in model.named_parameters() if p.requires_grad]):
        e = parity_err(tr.sync.views[i].detach().cpu().numpy(), ref_model.p[k].grad.numpy())
        assert e <= 2e-4, f"{k}: {e:.2e}"


@pytest.mark.gpu
def test_fused_trainer_learns_and_refuses_what_it_does_not_implement(gpu_device):
    from chemprop_amd import agg as cagg, synth
    from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
    from chemprop_amd.nn import BondMessagePassing

    torch.manual_seed(0)
    model = MPNN(BondMessagePassing(d_h=64), cagg.MeanAggregation(), RegressionFFN(n_tasks=1, input_dim=64, hidden_dim=64), batch_norm=True).to(gpu_device).train()
    bmg = synth.random_batch(64, "qm9", seed=2)
    bmg.to(gpu_device)
    y = torch.randn(64, 1, device=gpu_device)
    tr = FusedTrainer(model, lr=3e-3)
    losses = [float(tr.step(bmg, y)[0]) for _ in range(60)]
    assert np.mean(losses[-5:]) < 0.5 * np.mean(losses[:5]), (losses[:5], losses[-5:])
    with pytest.raises(NotImplementedError):   # (dropout inside the block: ReLU-class activations only — the sign of the kept tensor carries the mask)
        FusedTrainer(MPNN(BondMessagePassing(d_h=64, dropout=0.1, activation="tanh"), cagg.MeanAggregation(), RegressionFFN(input_dim=64)).to(gpu_device))
    with pytest.raises(NotImplementedError):   # (dropout in the predictor: the module path)
        FusedTrainer(MPNN(BondMessagePassing(d_h=64), cagg.MeanAggregation(), RegressionFFN(input_dim=64, dropout=0.1)).to(gpu_device))
    with pytest.raises(NotImplementedError):
        FusedTrainer(MPNN(BondMessagePassing(d_h=64), cagg.AttentiveAggregation(output_size=64), RegressionFFN(input_dim=64)).to(gpu_device))
    with pytest.raises(ValueError):
        tr.step(bmg, torch.randn(63, 1, device=gpu_device))


# ------------------------------------------------------------------------------------------------
# round 4: what the round-3 advisor found
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("d_h", [400, 600])
def test_fused_trainer_beyond_the_tile_kernels_widths(d_h, gpu_device):
    """d_h > 320 (common chemprop widths; hpopt searches 300-2400, hpopt.py:73) rules the tile kernels out: the trainer must keep
    asking for the FULL plan after the validated window (round 3 asked for a tile plan from the third step on and raised).  Four
    steps against the module path (MPNN.loss + autograd + FlatAdam on a twin), tanh so that no mask can flip."""
    from chemprop_amd import agg as cagg, distributed as ddp, synth
    from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
    from chemprop_amd.nn import BondMessagePassing
    from chemprop_amd.optim import FlatAdam

    def make():
        torch.manual_seed(5)
        return MPNN(BondMessagePassing(d_h=d_h, activation="tanh"), cagg.MeanAggregation(),
                    RegressionFFN(n_tasks=2, input_dim=d_h, hidden_dim=64, activation="tanh"), batch_norm=True).to(gpu_device).train()

    batches = [synth.random_batch(48, "qm9", seed=90 + i) for i in range(4)]
    for b in batches:
        b.to(gpu_device)
    ys = [torch.randn(48, 2, generator=torch.Generator().manual_seed(i)).to(gpu_device) for i in range(4)]
    a, b = make(), make()
    tr = FusedTrainer(a, lr=1e-3)
    sync = ddp.GradSync(list(b.parameters()), modules=[b])
    opt = FlatAdam(sync, lr=1e-3)
    for i in range(4):
        la = tr.step(batches[i], ys[i])
        assert tr.last_route not in ("mega16", "mega"), tr.last_route
        lb = b.loss(batches[i], ys[i])
        lb.backward()
        sync.allreduce()
        opt.step()
        sync.zero_grad()
        assert abs(float(la[0]) - float(lb)) <= 2e-5 * max(1.0, abs(float(lb))), (i, float(la[0]), float(lb))
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        # (after FOUR Adam steps: the update g / (|g| + eps) amplifies 1e-7 differences of near-zero gradients — the functional bar)
        assert parity_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) <= 3e-4, k


@pytest.mark.gpu
def test_dropout_with_frozen_edge_weights_keeps_its_scale(gpu_device):
    """W_i / W_h frozen, W_o trainable, dropout > 0 (a frozen encoder fine-tuned at its read-out, train.py:1826-1828): the backward
    tile kernel — the only holder of the in-kernel dropout's 1 / (1 - p) — does not run, so the forward must not take the in-kernel
    dropout either (round 3 returned gW_o without the factor).  Parity GIVEN the masks against the restated forward."""
    from chemprop_amd import synth
    from chemprop_amd.nn import BondMessagePassing
    from test_dropout_gpu import RecordingDropout, ReplayDropout, _restated_forward

    p = 0.3
    cpu_bmg = synth.random_batch(64, "qm9", seed=8)
    torch.manual_seed(2)
    mp = BondMessagePassing(d_h=64, dropout=p)
    state = {k: v.clone() for k, v in mp.state_dict().items()}
    for lin in (mp.W_i, mp.W_h):
        lin.weight.requires_grad_(False)
    mp = mp.to(gpu_device).train()
    mp.dropout = RecordingDropout(p).train()
    bmg = synth.random_batch(64, "qm9", seed=8)
    bmg.to(gpu_device)
    G = torch.randn(cpu_bmg.V.shape[0], 64, generator=torch.Generator().manual_seed(1))
    for _ in range(3):   # (past the validated window: the steady path must behave the same)
        mp.zero_grad()
        mp.dropout.masks.clear()
        out = mp(bmg)
        (out * G.to(gpu_device)).sum().backward()
    masks = [m.cpu() for m in mp.dropout.masks]
    assert len(masks) == mp.depth            # the block's own dropout module ran at every site: not the in-kernel hash
    ref = BondMessagePassing(d_h=64, dropout=p)
    ref.load_state_dict(state)
    ref.train()
    ref_out = _restated_forward(cpu_bmg, ref, ReplayDropout(p, masks))
    (ref_out * G).sum().backward()
    assert parity_err(out.detach().cpu().numpy(), ref_out.detach().numpy()) <= 1e-5
    assert mp.W_i.weight.grad is None and mp.W_h.weight.grad is None
    for k in ("weight", "bias"):
        assert parity_err(getattr(mp.W_o, k).grad.cpu().numpy(), getattr(ref.W_o, k).grad.numpy()) <= 2e-5, k


@pytest.mark.gpu
def test_backward_refuses_in_kernel_dropout_without_the_tile_kernel(gpu_device, monkeypatch):
    """The C boundary itself: a forward that ran with dropout inside the kernels followed by a backward that cannot take the tile
    kernel (no gradient of W_i / W_h wanted) is an argument error, not a silently unscaled gradient.  (With the messages kept as split
    rows — the default since round 4 — the backward pass ALWAYS takes the tile kernel: checked below against the run that refuses.)"""
    from chemprop_amd import _lib, engine, synth
    from chemprop_amd.nn import BondMessagePassing

    monkeypatch.setenv("DMPNN_KEEP_ROWS", "0")

    bmg = synth.random_batch(32, "qm9", seed=3)
    bmg.to(gpu_device)
    mp = BondMessagePassing(d_h=64).to(gpu_device)
    plan = engine.GraphPlan.from_bmg(bmg)
    out, st = engine.forward(plan, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, keep=True, dropout=(0.25, 77))
    assert st.route == "mega16"
    g = torch.ones_like(out)
    with pytest.raises(_lib.DmpnnError, match="dropout inside the kernels"):
        engine.backward(st, g, dict(W_o=True, b_o=True))
    grads = engine.backward(st, g, dict(W_i=True, W_h=True, W_o=True, b_o=True))   # (with the edge gradients wanted: fine)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in grads.values() if v is not None)
    monkeypatch.setenv("DMPNN_KEEP_ROWS", "1")
    out2, st2 = engine.forward(plan, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, keep=True, dropout=(0.25, 77))
    assert st2.args.msplit and torch.equal(out2, out)
    g2 = engine.backward(st2, g, dict(W_o=True, b_o=True))   # split rows kept: the tile kernel whatever is wanted — the scaled gradient
    for k in ("W_o", "b_o"):
        assert parity_err(g2[k].cpu().numpy(), grads[k].cpu().numpy()) <= 2e-6, k
    with pytest.raises(engine.RouteUnavailable):   # PReLU is not a dropout activation of the tile kernels
        engine.forward(plan, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, keep=True, dropout=(0.25, 77),
                       act="prelu", slope_t=torch.full((1,), 0.25, device=gpu_device))


@pytest.mark.gpu
def test_fused_trainer_validates_what_it_hands_over_as_raw_pointers(gpu_device):
    """A wrong dtype / shape must be an error at the boundary, not an out-of-bounds device read; a refused step must not advance
    Adam's bias correction; eval mode and a batch of one molecule under batch norm are refused like torch refuses them."""
    from chemprop_amd import agg as cagg, synth
    from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
    from chemprop_amd.nn import BondMessagePassing

    torch.manual_seed(0)
    m = MPNN(BondMessagePassing(d_h=64), cagg.MeanAggregation(), RegressionFFN(n_tasks=2, input_dim=64, hidden_dim=32), batch_norm=True).to(gpu_device).train()
    tr = FusedTrainer(m, lr=1e-3)
    bmg = synth.random_batch(16, "qm9", seed=1)
    bmg.to(gpu_device)
    y = torch.randn(16, 2, device=gpu_device)
    good = bmg.batch
    for bad in (good.int(), good[:-1], good.cpu()):
        bmg.batch = bad
        with pytest.raises(ValueError, match="bmg.batch"):
            tr.step(bmg, y)
    bmg.batch = good
    with pytest.raises(ValueError, match="weights"):
        tr.step(bmg, y, weights=torch.ones(15, device=gpu_device))
    with pytest.raises(ValueError, match="lt_mask"):
        tr.step(bmg, y, lt_mask=torch.zeros(16, 1, dtype=torch.bool, device=gpu_device))
    assert tr.opt.steps == 0
    tr.step(bmg, y)
    assert tr.opt.steps == 1
    m.eval()
    with pytest.raises(RuntimeError, match="eval mode"):
        tr.step(bmg, y)
    m.train()
    one = synth.random_batch(1, "qm9", seed=2)
    one.to(gpu_device)
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        tr.step(one, torch.randn(1, 2, device=gpu_device))
    assert tr.opt.steps == 1


@pytest.mark.gpu
@pytest.mark.parametrize("bn,agg,tasks,kind,act", [(True, "norm", 1, "mse", "relu"), (False, "mean", 3, "mae", "tanh"), (True, "sum", 2, "bounded-mse", "elu")])
def test_module_path_head_is_one_autograd_node(bn, agg, tasks, kind, act, gpu_device):
    """``MPNN.loss`` (the module path of a training step): everything behind the block — aggregation, batch norm, predictor, criterion
    and their backward — is ONE autograd node on the head kernels (``model.head_loss``).  Loss, every gradient and the batch-norm
    buffers against the same model run through the torch modules (``fingerprint`` / ``predictor`` / ``masked_loss``: the reference's
    op sequence), missing targets, sample weights and bounds included; a loss scaled before ``backward`` scales every gradient."""
    from chemprop_amd import synth
    from chemprop_amd.model import head_loss, masked_loss

    cfg = dict(mp=dict(d_h=64, activation=act), agg=agg, bn=bn, ffn=dict(n_tasks=tasks, hidden_dim=48, n_layers=2, activation=act), criterion=kind)
    torch.manual_seed(3)
    a = build_mirror(cfg).to(gpu_device).train()
    b = build_mirror(cfg).to(gpu_device).train()
    b.load_state_dict(a.state_dict())
    bmg = synth.random_batch(40, "qm9", seed=6)
    bmg.to(gpu_device)
    gen = torch.Generator().manual_seed(1)
    y = torch.randn(40, tasks, generator=gen)
    if tasks > 1:
        y[torch.rand(40, tasks, generator=gen) < 0.2] = float("nan")
    w = (0.5 + torch.rand(40, 1, generator=gen)).to(gpu_device)
    bounded = kind.startswith("bounded")
    lt = (torch.rand(40, tasks, generator=gen) < 0.3).to(gpu_device) if bounded else None
    gt = (torch.rand(40, tasks, generator=gen) < 0.3).to(gpu_device) if bounded else None
    y = y.to(gpu_device)
    la = a.loss(bmg, y, w, lt, gt)
    assert type(la.grad_fn).__name__ == "_HeadLossBackward", type(la.grad_fn).__name__
    (2.5 * la).backward()
    # the torch modules, op by op (what MPNN.loss ran before round 4)
    preds = b.predictor.train_step(b.fingerprint(bmg))
    c = b.criterion
    lb = masked_loss(preds, y, w, getattr(c, "task_weights", None), lt, gt, getattr(c, "kind", "mse"))
    (2.5 * lb).backward()
    torch.cuda.synchronize()
    assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb))), (float(la), float(lb))
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert (pa.grad is None) == (pb.grad is None), k
        if pa.grad is not None:
            assert parity_err(pa.grad.cpu().numpy(), pb.grad.cpu().numpy()) <= 2e-5, k
    if bn:
        for k in ("running_mean", "running_var"):
            assert parity_err(getattr(a.bn, k).cpu().numpy(), getattr(b.bn, k).cpu().numpy()) <= 1e-6, k
        assert int(a.bn.num_batches_tracked) == int(b.bn.num_batches_tracked) == 1
    # what the head kernels do not implement falls back to the torch modules: extra descriptors X_d, a frozen-in-eval model
    a.eval()
    assert type(a.loss(bmg, y, w, lt, gt).grad_fn).__name__ != "_HeadLossBackward"


@pytest.mark.gpu
@pytest.mark.parametrize("n_mols,d_h,hidden,tasks,bn,agg,kind,act,switches", [
    (512, 300, 300, 1, True, "norm", "mse", "relu", {}),          # the headline model's head: 32 row blocks x 5 column slices, 2 quads per column workgroup
    (1000, 300, 200, 4, False, "mean", "bce", "leakyrelu", {}),   # 2 molecules per thread in the column kernels, N != K, a last row block of 8, the aggregation in front
    (77, 64, 128, 3, True, "sum", "bounded-mse", "elu", {}),      # 2 operand chunks of 64 columns, 4 quads per column workgroup, bounds, missing targets
    (16, 128, 36, 2, True, "mean", "mae", "tanh", {}),            # one row block, N not a multiple of 16
    (512, 300, 300, 2, True, "mean", "mse", "relu", {"DMPNN_HEAD_AGG": "split", "DMPNN_HEAD_QPW": "4"}),   # the forms the size rules do not pick here
    (1000, 300, 300, 1, True, "norm", "mse", "relu", {"DMPNN_HEAD_AGG": "fused", "DMPNN_HEAD_QPW": "4"}),  # ... 4 molecules per thread
])
def test_head_in_row_and_column_kernels_equals_the_nine_launch_chain(n_mols, d_h, hidden, tasks, bn, agg, kind, act, switches, gpu_device, monkeypatch):
    """Round 5: aggregation + batch norm (+ the hidden layer's weight split) as one column kernel, predictor + criterion + their
    backward as two launches over (row block) x (column slice) on the f16 pipe (3-product split), batch norm backward + the broadcast to
    the atoms (+ the sums over row blocks) as one column kernel (``csrc/dmpnn_head.hip``: ``k_agg_bn_fwd`` / ``k_head_rows<., 1 | 2>`` /
    ``k_bn_agg_bwd``) against the chain of rounds 3-4 (``DMPNN_HEAD=chain``: one exact-fp32 contraction per layer): loss, every
    gradient (the block's included — ``dl/dH_v`` goes through it), batch-norm buffers."""
    from chemprop_amd import synth

    cfg = dict(mp=dict(d_h=d_h, activation="relu"), agg=agg, bn=bn, ffn=dict(n_tasks=tasks, hidden_dim=hidden, n_layers=1, activation=act),
               criterion=kind)
    if kind == "bce":
        cfg["predictor"] = "classification"
    torch.manual_seed(5)
    a = build_mirror(cfg).to(gpu_device).train()
    b = build_mirror(cfg).to(gpu_device).train()
    b.load_state_dict(a.state_dict())
    bmg = synth.random_batch(n_mols, "qm9", seed=9)
    bmg.to(gpu_device)
    gen = torch.Generator().manual_seed(2)
    y = torch.rand(n_mols, tasks, generator=gen).round() if kind == "bce" else torch.randn(n_mols, tasks, generator=gen)
    if tasks > 1:
        y[torch.rand(n_mols, tasks, generator=gen) < 0.2] = float("nan")
    w = (0.5 + torch.rand(n_mols, 1, generator=gen)).to(gpu_device)
    bounded = kind.startswith("bounded")
    lt = (torch.rand(n_mols, tasks, generator=gen) < 0.3).to(gpu_device) if bounded else None
    gt = (torch.rand(n_mols, tasks, generator=gen) < 0.3).to(gpu_device) if bounded else None
    y = y.to(gpu_device)
    monkeypatch.setenv("DMPNN_HEAD", "rows")   # (a shape that falls back to the chain is an error under this value)
    for k, v in switches.items():
        monkeypatch.setenv(k, v)
    la = a.loss(bmg, y, w, lt, gt)
    assert type(la.grad_fn).__name__ == "_HeadLossBackward"
    la.backward()
    monkeypatch.setenv("DMPNN_HEAD", "chain")
    lb = b.loss(bmg, y, w, lt, gt)
    lb.backward()
    torch.cuda.synchronize()
    assert abs(float(la) - float(lb)) <= 2e-6 * max(1.0, abs(float(lb))), (float(la), float(lb))
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert (pa.grad is None) == (pb.grad is None), k
        if pa.grad is not None:
            assert torch.isfinite(pa.grad).all(), k
            assert parity_err(pa.grad.cpu().numpy(), pb.grad.cpu().numpy()) <= 1e-5, k
    if bn:
        for k in ("running_mean", "running_var"):
            assert parity_err(getattr(a.bn, k).cpu().numpy(), getattr(b.bn, k).cpu().numpy()) <= 1e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize("n_mols,agg,bn", [(512, "norm", True), (77, "mean", True), (200, "sum", False)])
def test_fused_step_takes_the_aggregate_from_the_forward_tile_kernel_bit_for_bit(n_mols, agg, bn, gpu_device, monkeypatch):
    """Round 5: inside ``dmpnn_train_step`` the forward tile kernel leaves ``agg(H_v)`` of every molecule of a regular tile with the
    tile (rows added in increasing atom order from the tile's LDS copy — ``k_mol_reduce``'s arithmetic) and marks it in the bounds
    table; the head's column kernel sums only what is not marked.  Against the same step with the ride switched off
    (``DMPNN_HEAD_AGG=fused``): loss and every gradient bit for bit."""
    from chemprop_amd import synth
    from chemprop_amd.model import FusedTrainer

    cfg = dict(mp=dict(d_h=300, activation="relu"), agg=agg, bn=bn, ffn=dict(n_tasks=2, hidden_dim=300, n_layers=1, activation="relu"))
    bmg = synth.random_batch(n_mols, "qm9", seed=21)
    bmg.to(gpu_device)
    y = torch.randn(n_mols, 2, generator=torch.Generator().manual_seed(4)).to(gpu_device)
    out = {}
    for form in ("tile", "fused"):
        monkeypatch.setenv("DMPNN_HEAD_AGG", form)
        torch.manual_seed(11)
        m = build_mirror(cfg).to(gpu_device).train()
        tr = FusedTrainer(m, lr=1e-3)
        l = tr.step(bmg, y)
        torch.cuda.synchronize()
        assert tr.last_route == "mega16"
        out[form] = (l.detach().cpu().clone(), [v.detach().cpu().clone() for v in tr.sync.views])
    assert torch.equal(out["tile"][0], out["fused"][0]), (out["tile"][0], out["fused"][0])
    assert torch.isfinite(out["tile"][0]).all()
    for i, (a, b) in enumerate(zip(out["tile"][1], out["fused"][1])):
        assert torch.equal(a, b), i


@pytest.mark.gpu
@pytest.mark.parametrize("tasks,act", [(1, "relu"), (3, "elu")])
def test_no_kernel_of_a_training_step_reads_lds_it_never_wrote(tasks, act, gpu_device):
    """Round 6: a cold box showed a NaN that no warm box reproduced — ``k_head_rows`` read three never-written columns of an LDS tile
    against zero weights (0 x the previous kernel's finite leftovers on a warm box, 0 x NaN on a cold one).  With the whole LDS of every
    CU filled with NaN after EVERY launch (``dmpnn_debug_lds_poison``) the fused step, the module path (a module's first, validated
    batches and the tile-plan ones) and the inference forward must reproduce the clean run BIT FOR BIT."""
    from chemprop_amd import _lib, synth
    from chemprop_amd.model import FusedTrainer

    lib = _lib.load()
    cfg = dict(mp=dict(activation=act), agg="mean", bn=True, ffn=dict(n_tasks=tasks, activation=act))
    n_mols = 300
    bmg = synth.random_batch(n_mols, "qm9", seed=5)
    bmg.to(gpu_device)
    gen = torch.Generator().manual_seed(3)
    tg = torch.randn(n_mols, tasks, generator=gen).to(gpu_device)
    wg = (0.5 + torch.rand(n_mols, 1, generator=gen)).to(gpu_device)

    def run():
        torch.manual_seed(11)
        model = build_mirror(cfg).to(gpu_device).train()
        state = {k: v.clone() for k, v in model.state_dict().items()}
        outs = []
        for _ in range(4):                       # batches 0, 1 on the full plan, 2, 3 on the tile plan
            model.load_state_dict(state)
            model.zero_grad(set_to_none=True)
            loss = model.loss(bmg, tg, wg)
            loss.backward()
            outs.append([loss.detach().clone()] + [p.grad.detach().clone() for p in model.parameters()])
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        tr = FusedTrainer(model, lr=0.0)
        for _ in range(3):
            out = tr.step(bmg, tg, wg)
            outs.append([out[0].detach().clone()] + [v.detach().clone() for v in tr.sync.views])
        model.eval()
        with torch.no_grad():
            outs.append([model(bmg).clone()])
        torch.cuda.synchronize()
        return outs

    clean = run()
    lib.dmpnn_debug_lds_poison(1)
    try:
        poisoned = run()
    finally:
        lib.dmpnn_debug_lds_poison(0)
    for i, (a, b) in enumerate(zip(clean, poisoned)):
        for j, (x, y) in enumerate(zip(a, b)):
            assert torch.isfinite(y).all(), f"run {i}, tensor {j}: non-finite under LDS poison"
            assert torch.equal(x, y), f"run {i}, tensor {j}: differs under LDS poison"
