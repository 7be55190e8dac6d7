"""Active dropout (``base.py:135-141`` ``self.dropout(H_t)``, ``:182,188`` in ``finalize``; CLI ``--dropout``,
``cli/train.py:225-230``) through the engine in ``.train()`` with p > 0 (round-2 VERDICT, parity gap a).

The reference draws its masks from torch's CPU generator, the engine's block from the device generator (its
``nn.Dropout`` module runs between the kernels, ``chemprop_amd/autograd.py``): the two streams cannot agree, so parity is
checked the way a stochastic op can be —

* GIVEN the masks the engine emitted, its output and its parameter gradients equal the reference's own forward / autograd
  with those masks replayed (the executed reference where a reference tree is present, else the restated op sequence);
* the masks themselves are Bernoulli(1 - p) scaled by 1 / (1 - p): keep fraction, independence across the call sites of one
  forward, a fresh draw every forward;
* ``E[out]`` over many draws of the LAST dropout equals the p = 0 output (depth 1: the only dropout is finalize's).
"""
import numpy as np
import pytest
import torch
from torch import nn

from conftest import TOL, parity_err
from oracle import dmpnn_torch as ot
from oracle import ref_shim

pytestmark = pytest.mark.gpu


class RecordingDropout(nn.Dropout):
    """An ``nn.Dropout`` (same ``p``, same inverted scaling, draws from the device generator like ``F.dropout``) that keeps the
    masks it emitted."""

    def __init__(self, p):
        super().__init__(p)
        self.masks = []

    def forward(self, x):
        if not self.training or self.p == 0:
            return x
        m = (torch.rand_like(x) >= self.p).to(x.dtype) / (1.0 - self.p)
        self.masks.append(m.detach())
        return x * m


class ReplayDropout(nn.Dropout):
    """Replays recorded masks in call order (the reference's call order: update x (depth - 1), finalize, W_d branch)."""

    def __init__(self, p, masks):
        super().__init__(p)
        self.masks, self.i = list(masks), 0

    def forward(self, x):
        m = self.masks[self.i]
        self.i += 1
        assert m.shape == x.shape
        return x * m


def _reference_block(kw, state):
    """The executed reference class when a reference tree is present (build container, or oracle/_ref on the GPU box)."""
    if not ref_shim.reference_available():
        return None
    BMP, _, _ = ref_shim.load_reference()
    ref = BMP(**kw)
    ref.load_state_dict(state)
    return ref


def _restated_forward(bmg, mp, drop, V_d=None):
    """base.py:196-212 with dropout active, restated on oracle/dmpnn_torch.py's pieces (used where no reference tree exists)."""
    w = ot.MPWeights(mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, mp.W_i.bias, mp.W_h.bias)
    tau = mp.tau
    src, dst, rev = bmg.edge_index[0], bmg.edge_index[1], bmg.rev_edge_index
    nV = bmg.V.shape[0]
    H0 = ot.initialize(bmg.V, bmg.E, src, w)
    H = tau(H0)
    for _ in range(1, mp.depth):
        if mp.undirected:
            H = (H + H[rev]) / 2
        M = ot.message(H, src, dst, rev, nV)
        H = drop(tau(H0 + torch.nn.functional.linear(M, w.W_h, w.b_h)))          # base.py:135-141
    Mv = ot.segment_sum_dst(H, dst, nV)
    Hv = drop(tau(torch.nn.functional.linear(torch.cat((bmg.V, Mv), 1), w.W_o, w.b_o)))   # base.py:180-183
    if V_d is not None and mp.W_d is not None:
        Hv = drop(torch.nn.functional.linear(torch.cat((Hv, V_d), 1), mp.W_d.weight, mp.W_d.bias))  # base.py:185-188
    return Hv


@pytest.mark.parametrize("n_mols,kind,kw,p", [
    (64, "qm9", dict(), 0.25),
    (48, "qm9", dict(d_h=96, depth=4, bias=True, activation="tanh"), 0.4),
    (32, "zinc", dict(d_h=128, depth=3, activation="elu", undirected=True), 0.1),
    (40, "qm9", dict(d_h=64, depth=3, d_vd=5), 0.3),                      # the W_d branch's own dropout (base.py:188)
    (512, "qm9", dict(), 0.2),                                            # the headline shape
])
def test_train_mode_dropout_given_the_emitted_masks(n_mols, kind, kw, p, gpu_device):
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph
    from chemprop_amd.nn import BondMessagePassing

    cpu_bmg = synth.random_batch(n_mols, kind, seed=21)
    torch.manual_seed(9)
    mp = BondMessagePassing(dropout=p, **kw)
    state = {k: v.clone() for k, v in mp.state_dict().items()}
    d_vd = kw.get("d_vd")
    V_d = torch.randn(cpu_bmg.V.shape[0], d_vd, generator=torch.Generator().manual_seed(2)) if d_vd else None
    G = torch.randn(cpu_bmg.V.shape[0], mp.output_dim, generator=torch.Generator().manual_seed(5))

    mp = mp.to(gpu_device).train()
    rec = RecordingDropout(p).train()
    mp.dropout = rec                                   # (same attribute the reference's update / finalize call)
    bmg = synth.random_batch(n_mols, kind, seed=21)
    bmg.to(gpu_device)
    out = mp(bmg, V_d.to(gpu_device) if V_d is not None else None)
    (out * G.to(gpu_device)).sum().backward()
    n_sites = (mp.depth - 1) + 1 + (1 if d_vd else 0)
    assert len(rec.masks) == n_sites, (len(rec.masks), n_sites)
    masks = [m.cpu() for m in rec.masks]

    # ---- mask statistics: Bernoulli(1 - p) / (1 - p), independent between the call sites ----
    for m in masks:
        pos = m[m > 0]
        assert torch.allclose(pos, torch.full_like(pos, 1.0 / (1.0 - p)))       # inverted dropout: kept entries are scaled by 1 / (1 - p)
        keep = float((m > 0).double().mean())
        tol = 5.0 * np.sqrt(p * (1 - p) / m.numel()) + 1e-4
        assert abs(keep - (1 - p)) <= tol, (keep, 1 - p, tol)
    if len(masks) >= 2 and masks[0].shape == masks[1].shape:
        both = float(((masks[0] > 0) & (masks[1] > 0)).double().mean())
        assert abs(both - (1 - p) ** 2) <= 6.0 * np.sqrt(1.0 / masks[0].numel()) + 1e-3   # (two sites do not share a mask)

    # ---- GIVEN the masks: the reference's forward and autograd ----
    ref = _reference_block(dict(dropout=p, **kw), state)
    if ref is not None:
        ref.train()
        ref.dropout = ReplayDropout(p, masks)
        _, BMG, _ = ref_shim.load_reference()
        ref_bmg = BMG(synth.random_molgraphs(n_mols, kind, seed=21))
        for k in ("V", "E", "edge_index", "rev_edge_index"):
            assert torch.equal(getattr(ref_bmg, k), getattr(cpu_bmg, k))
        ref_out = ref(ref_bmg, V_d)
        named_ref = dict(ref.named_parameters())
    else:
        ref = BondMessagePassing(dropout=p, **kw)
        ref.load_state_dict(state)
        ref.train()
        ref_out = _restated_forward(cpu_bmg, ref, ReplayDropout(p, masks), V_d)
        named_ref = dict(ref.named_parameters())
    (ref_out * G).sum().backward()
    assert parity_err(out.detach().cpu().numpy(), ref_out.detach().numpy()) <= TOL
    for k, prm in mp.named_parameters():
        if prm.grad is None:
            assert named_ref[k].grad is None or float(named_ref[k].grad.abs().max()) == 0.0
            continue
        err = parity_err(prm.grad.cpu().numpy(), named_ref[k].grad.numpy())
        assert err <= 2e-5, f"{k}: {err:.3e}"


def test_real_dropout_module_statistics_and_expectation(gpu_device):
    """The stock ``nn.Dropout`` of the block (no recording): a fresh mask every forward; with depth 1 the only dropout is
    finalize's (base.py:182), so ``E[out] = out(p = 0)`` exactly and an entry is zeroed with probability p."""
    from chemprop_amd import synth
    from chemprop_amd.nn import BondMessagePassing

    p, n_draws = 0.35, 400
    bmg = synth.random_batch(24, "qm9", seed=3)
    bmg.to(gpu_device)
    torch.manual_seed(2)
    mp = BondMessagePassing(d_h=64, depth=1, dropout=p, activation="tanh").to(gpu_device)
    mp.eval()
    with torch.no_grad():
        base = mp(bmg)                        # eval: dropout is the identity (and the fused routes run)
    mp.train()
    acc = torch.zeros_like(base)
    zeros = torch.zeros_like(base)
    with torch.no_grad():
        first = mp(bmg)
        second = mp(bmg)
        assert not torch.equal(first, second)                          # a fresh draw every forward
        for _ in range(n_draws):
            o = mp(bmg)
            kept = o != 0
            # a kept entry is the p = 0 value times 1 / (1 - p)
            assert float((o[kept] - base[kept] / (1 - p)).abs().max()) <= 1e-5 * max(1.0, float(base.abs().max()))
            acc += o
            zeros += (~kept).float()
    mean = acc / n_draws
    # E[out] = base: the standard error of an entry's mean is |base| sqrt(p / ((1 - p) n))
    se = base.abs() * np.sqrt(p / ((1 - p) * n_draws))
    assert bool(((mean - base).abs() <= 6.0 * se + 1e-6).all())
    nz = base != 0
    frac = float(zeros[nz].sum() / (n_draws * int(nz.sum())))
    assert abs(frac - p) <= 5.0 * np.sqrt(p * (1 - p) / (n_draws * int(nz.sum()))) + 1e-4, frac


def test_dropout_training_gradients_flow_and_eval_is_deterministic(gpu_device):
    """A short optimisation with active dropout decreases a fit loss; switching to eval gives the deterministic forward."""
    from chemprop_amd import synth
    from chemprop_amd.nn import BondMessagePassing

    bmg = synth.random_batch(32, "qm9", seed=4)
    bmg.to(gpu_device)
    torch.manual_seed(0)
    mp = BondMessagePassing(d_h=64, dropout=0.2).to(gpu_device).train()
    target = torch.randn(int(bmg.V.shape[0]), 64, device=gpu_device) * 0.1
    opt = torch.optim.Adam(mp.parameters(), lr=3e-3)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        loss = ((mp(bmg) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), (losses[:5], losses[-5:])
    mp.eval()
    with torch.no_grad():
        assert torch.equal(mp(bmg), mp(bmg))


# ------------------------------------------------------------------------------------------------
# round 3: dropout INSIDE the tile kernels (dmpnn_fwd_args.dropout_p) — the mask is a counter-based hash, restated in oracle/
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_mols,kw,p", [(64, dict(), 0.25), (96, dict(d_h=128, depth=4, activation="leakyrelu", bias=True), 0.4),
                                         (512, dict(), 0.1), (2048, dict(), 0.3)])
def test_fused_dropout_on_the_tile_kernels_given_its_masks(n_mols, kw, p, gpu_device):
    """``.train()`` with p > 0 on the tile route: ONE forward launch and ONE backward tile launch carry the dropout.  The masks are
    the oracle's restatement of the kernels' hash for the seed the forward drew; the executed reference with those masks replayed
    gives the output and every gradient."""
    from chemprop_amd import synth
    from chemprop_amd.nn import BondMessagePassing
    from oracle import dropout_hash as dh

    cpu_bmg = synth.random_batch(n_mols, "qm9", seed=33)
    torch.manual_seed(4)
    mp = BondMessagePassing(dropout=p, **kw)
    state = {k: v.clone() for k, v in mp.state_dict().items()}
    G = torch.randn(cpu_bmg.V.shape[0], mp.output_dim, generator=torch.Generator().manual_seed(6))
    mp = mp.to(gpu_device).train()
    bmg = synth.random_batch(n_mols, "qm9", seed=33)
    bmg.to(gpu_device)
    for _ in range(2):   # (the module's first batches are validated; the steady path must do the same)
        mp.zero_grad()
        torch.manual_seed(1234)
        out = mp(bmg)
        st = out.grad_fn.st
        assert st.route == "mega16", st.route
        seed, p_used = int(st.args.dropout_seed), float(st.args.dropout_p)
        assert abs(p_used - p) < 1e-7 and seed > 0
        (out * G.to(gpu_device)).sum().backward()
    torch.manual_seed(1234)
    again = mp(bmg)
    torch.manual_seed(1234)
    again2 = mp(bmg)
    assert torch.equal(again.detach(), again2.detach())                  # torch.manual_seed fixes the masks
    # (the two validated batches ran on the CSR plan, the steady ones on the tile table: the same masks — keyed on the caller's edge
    #  ids — and the same function to summation order)
    assert parity_err(again.detach().cpu().numpy(), out.detach().cpu().numpy()) <= 2e-6
    assert float(((again.detach() == 0) != (out.detach() == 0)).float().mean()) < 1e-4
    assert not torch.equal(mp(bmg).detach(), again.detach())             # ... and the next forward draws new ones
    assert again.grad_fn.st.plan.tiles_only and not st.plan.tiles_only   # (the validated batches ran on the CSR plan, the steady one on the tile table:
    d_h, nE, nV = mp.W_h.weight.shape[0], cpu_bmg.E.shape[0], cpu_bmg.V.shape[0]   #  the hash is keyed on the caller's edge id — the same masks on both)
    scale = 1.0 / (1.0 - p)
    masks = [torch.from_numpy(dh.keep_mask(seed, t, nE, d_h, p).astype(np.float32) * np.float32(scale)) for t in range(mp.depth - 1)]
    masks.append(torch.from_numpy(dh.keep_mask(seed, mp.depth - 1, nV, d_h, p).astype(np.float32) * np.float32(scale)))
    # what the kernel zeroed is what the hash says (the output's zero pattern at kept-and-active entries aside)
    fin = masks[-1] > 0
    assert bool((out.detach().cpu()[~fin] == 0).all())
    keep_frac = float(fin.float().mean())
    assert abs(keep_frac - (1 - p)) <= 5 * np.sqrt(p * (1 - p) / fin.numel()) + 1e-4
    # ---- GIVEN the dropout masks AND the activation masks the engine's forward used (the method of test_relu_gradients_at_size) ----
    # A kinked activation makes a gradient only as reproducible as its masks: at 2 048 molecules (37 M activations) a pre-activation
    # within 1e-8 of the kink flips between two fp32-class arithmetics and moves a gradient entry by ~3e-3 of the largest one
    # (scripts/dbg_drop2048b.py).  So: (a) the reference forward with the dropout masks replayed gives the TRUE activation masks; the
    # engine's — the signs of its kept tensors at the entries dropout kept — may differ from them in a handful of entries, each ON
    # the kink; (b) with the engine's activation masks replayed as well, the backward pass is linear algebra and every gradient holds
    # the fp32 bar at every size.
    slope = {"relu": 0.0, "leakyrelu": 0.1}[str(kw.get("activation", "relu"))]
    rows = not st.plan.tiles_only          # kept edge tensors of a CSR plan are in row order (row i = edge perm[i])
    to_edges = (lambda X: X[st.plan.inv32.long()]) if rows else (lambda X: X)
    eng_pos = [(to_edges(st.H0[:, :d_h]) > 0).cpu()]                              # site 0: tau(H_0), no dropout (base.py:200)
    eng_pos += [(to_edges(st.Hs[t][:, :d_h]) > 0).cpu() for t in range(mp.depth - 1)]   # post-dropout H^(t): > 0 iff active AND kept
    eng_pos.append((out.detach() > 0).cpu())
    kept = [torch.ones_like(eng_pos[0])] + [m > 0 for m in masks]                 # where the activation mask matters at all

    class RecordingTau(nn.Module):
        """tau of the reference, recording every pre-activation it sees (call order: H_0, the updates, finalize)."""
        def __init__(self, inner):
            super().__init__()
            self.inner, self.pre = inner, []

        def forward(self, z):
            self.pre.append(z.detach().clone())
            return self.inner(z)

    class ReplayTau(nn.Module):
        """A ReLU-class activation with its 0 / 1 decisions fixed: z * (m + (1 - m) * slope)."""
        def __init__(self, pos, slope):
            super().__init__()
            self.f, self.i = [m.float() + (1.0 - m.float()) * slope for m in pos], 0

        def forward(self, z):
            f = self.f[self.i]
            self.i += 1
            assert f.shape == z.shape
            return z * f

    def run_reference(tau_of):
        ref = _reference_block(dict(dropout=p, **kw), state)
        if ref is not None:
            ref.train()
            ref.dropout = ReplayDropout(p, masks)
            ref.tau = tau_of(ref.tau)
            _, BMG, _ = ref_shim.load_reference()
            ref_out = ref(BMG(synth.random_molgraphs(n_mols, "qm9", seed=33)))
        else:
            ref = BondMessagePassing(dropout=p, **kw)
            ref.load_state_dict(state)
            ref.train()
            ref.tau = tau_of(ref.tau)
            ref_out = _restated_forward(cpu_bmg, ref, ReplayDropout(p, masks))
        return ref, ref_out

    ref, ref_out = run_reference(RecordingTau)
    assert parity_err(out.detach().cpu().numpy(), ref_out.detach().numpy()) <= TOL
    pre = ref.tau.pre
    assert len(pre) == mp.depth + 1
    flips = 0
    for z, e_pos, k in zip(pre, eng_pos, kept):
        diff = ((z > 0) != e_pos) & k
        flips += int(diff.sum())
        if diff.any():   # a differing decision sits ON the kink: |z| within fp32 rounding of the values that were summed
            assert float(z[diff].abs().max()) <= 1e-5 * max(1.0, float(z.abs().max())), "an activation mask differs away from the kink"
    assert flips <= 8, f"{flips} activation-mask disagreements"
    # the engine's decision where dropout kept the entry, the reference's own where it did not (the entry is multiplied by 0 there)
    cond = [torch.where(k, e_pos, z > 0) for z, e_pos, k in zip(pre, eng_pos, kept)]
    ref, ref_out = run_reference(lambda inner: ReplayTau(cond, slope))
    named_ref = dict(ref.named_parameters())
    (ref_out * G).sum().backward()
    assert parity_err(out.detach().cpu().numpy(), ref_out.detach().numpy()) <= TOL
    errs = {k: parity_err(prm.grad.cpu().numpy(), named_ref[k].grad.numpy()) for k, prm in mp.named_parameters()}
    print(f"dropout-{n_mols}: activation-mask disagreements {flips}; gradient errors given the masks {errs}")
    assert max(errs.values()) <= 2e-5, errs


def test_fused_dropout_falls_back_where_the_tile_kernel_does_not_apply(gpu_device):
    """Molecules beyond the tile, a smooth activation, a V_d branch: the block's own nn.Dropout between the row kernels, as before."""
    from chemprop_amd import synth
    from chemprop_amd.nn import BondMessagePassing

    for kind, kw in (("zinc", dict(d_h=64)), ("qm9", dict(d_h=64, activation="tanh"))):
        bmg = synth.random_batch(24, kind, seed=1)
        bmg.to(gpu_device)
        mp = BondMessagePassing(dropout=0.2, **kw).to(gpu_device).train()
        for _ in range(3):
            out = mp(bmg)
            assert not hasattr(out.grad_fn, "st") or out.grad_fn.st.args.dropout_p == 0.0
            out.sum().backward()
            assert all(torch.isfinite(p.grad).all() for p in mp.parameters())
