"""f2 — the mol-atom-bond blocks MABBondMessagePassing / MABAtomMessagePassing
(chemprop/nn/message_passing/mol_atom_bond.py:16-388): the oracle restatement and the HIP-kernel mirror against
goldens frozen from the executed reference (tests/golden/make_golden_mab.py)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, TOL, parity_err

MAB = sorted(glob.glob(os.path.join(GOLDEN_DIR, "mab", "*.npz")))


class Case:
    def __init__(self, path):
        z = np.load(path)
        self.arr = {k: z[k] for k in z.files}
        self.meta = json.loads(bytes(self.arr.pop("meta")).decode())
        self.cfg = self.meta["cfg"]
        self.atom = self.meta["kind"] == "atom"

    def __getitem__(self, k):
        return self.arr[k]

    def __contains__(self, k):
        return k in self.arr

    def state_dict(self):
        return {k[2:]: torch.from_numpy(np.array(v)) for k, v in self.arr.items() if k.startswith("w.")}

    def cls(self):
        from chemprop_amd.mab import MABAtomMessagePassing, MABBondMessagePassing

        return MABAtomMessagePassing if self.atom else MABBondMessagePassing

    def module(self, device="cpu"):
        mp = self.cls()(**self.cfg)
        mp.load_state_dict(self.state_dict())
        return mp.eval().to(device)

    def bmg(self, device="cpu"):
        from chemprop_amd.data import BatchMolGraph

        t = lambda k: torch.from_numpy(self.arr[k])
        b = BatchMolGraph.from_tensors(t("V"), t("E"), t("edge_index"), t("rev_edge_index"), t("batch"), self.meta["n_mols"])
        if device != "cpu":
            b.to(device)
        return b

    def descriptors(self, device="cpu"):
        return tuple(torch.from_numpy(self.arr[k]).to(device) if k in self.arr else None for k in ("V_d", "E_d"))


@pytest.fixture(params=MAB, ids=[os.path.basename(p)[:-4] for p in MAB])
def mab_case(request):
    return Case(request.param)


def test_goldens_exist():
    assert len(MAB) >= 11


def test_oracle_matches_golden(mab_case):
    from oracle import dmpnn_torch as ot

    c = mab_case
    t = lambda k: torch.from_numpy(c[k])
    V_d, E_d = c.descriptors()
    H_v, H_e = ot.mab_forward(t("V"), t("E"), t("edge_index"), t("rev_edge_index"), ot.MABWeights.from_state_dict(c.state_dict()),
                              atom_messages=c.atom, depth=c.cfg.get("depth", 3), activation=c.cfg.get("activation", "relu"),
                              undirected=c.cfg.get("undirected", False), V_d=V_d, E_d=E_d)
    for tag, H in (("H_v", H_v), ("H_e", H_e)):
        assert (H is None) == (tag not in c)
        if H is not None:
            assert parity_err(H.numpy(), c[tag]) <= 1e-6, tag


def test_mirror_reproduces_the_reference_rng_stream(mab_case):
    """Same constructor order as mol_atom_bond.py:318-335,371-388: identical initial weights, keys and output_dims."""
    torch.manual_seed(mab_case.meta["seed"])
    mp = mab_case.cls()(**mab_case.cfg)
    sd = mab_case.state_dict()
    assert list(mp.state_dict().keys()) == list(sd.keys())
    for k, v in mp.state_dict().items():
        assert torch.equal(v, sd[k]), k
    dv, de = mp.output_dims
    assert dv == (mab_case["H_v"].shape[1] if "H_v" in mab_case else None)
    assert de == (mab_case["H_e"].shape[1] if "H_e" in mab_case else None)


def test_fails_loudly_off_device(mab_case):
    """No CPU fallback: host tensors are refused before any arithmetic."""
    with pytest.raises(RuntimeError):
        mab_case.module()(mab_case.bmg(), *mab_case.descriptors())


@pytest.mark.gpu
def test_forward_and_gradients_vs_executed_reference(mab_case, gpu_device):
    c = mab_case
    mp, bmg = c.module(gpu_device), c.bmg(gpu_device)
    V_d, E_d = c.descriptors(gpu_device)
    H_v, H_e = mp(bmg, V_d, E_d)  # grad enabled, parameters require grad: the per-step kernels with autograd
    loss = 0.0
    for tag, H in (("v", H_v), ("e", H_e)):
        assert (H is None) == ("H_" + tag not in c)
        if H is not None:
            assert parity_err(H.detach().cpu().numpy(), c["H_" + tag]) <= TOL, tag
            loss = loss + (H * torch.from_numpy(c["G_" + tag]).to(gpu_device)).sum()
    loss.backward()
    for k, p in mp.named_parameters():
        ref = c["g." + k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        assert parity_err(got, ref) <= 2e-5, k
    with torch.no_grad():  # inference: the bond variant takes the whole-forward tile kernel + one fused read-out GEMM
        I_v, I_e = mp(bmg, V_d, E_d)
    for tag, H in (("v", I_v), ("e", I_e)):
        if H is not None:
            assert parity_err(H.cpu().numpy(), c["H_" + tag]) <= TOL, tag


@pytest.mark.gpu
def test_bad_descriptor_shapes(gpu_device):
    from chemprop_amd.mab import MABBondMessagePassing
    from chemprop_amd.nn import InvalidShapeError
    from chemprop_amd import synth

    bmg = synth.random_batch(4, "qm9", seed=3)
    bmg.to(gpu_device)
    mp = MABBondMessagePassing(d_h=16, d_vd=2, d_ed=3).eval().to(gpu_device)
    nV, nE = bmg.V.shape[0], bmg.E.shape[0]
    ok_v, ok_e = torch.zeros(nV, 2, device=gpu_device), torch.zeros(nE, 3, device=gpu_device)
    with torch.no_grad():
        H_v, H_e = mp(bmg, ok_v, ok_e)
        assert H_v.shape == (nV, 18) and H_e.shape == (nE, 19)
        with pytest.raises(InvalidShapeError):
            mp(bmg, torch.zeros(nV, 3, device=gpu_device), ok_e)
        with pytest.raises(InvalidShapeError):  # the reference's hint: E_d must have one row per DIRECTED edge
            mp(bmg, ok_v, torch.zeros(nE // 2, 3, device=gpu_device))


@pytest.mark.gpu
@pytest.mark.parametrize("atom", [False, True])
def test_full_size_vs_oracle(atom, gpu_device):
    from chemprop_amd import synth
    from chemprop_amd.mab import MABAtomMessagePassing, MABBondMessagePassing
    from oracle import dmpnn_torch as ot

    bmg = synth.random_batch(512, "qm9", seed=78)
    torch.manual_seed(6)
    mp = (MABAtomMessagePassing if atom else MABBondMessagePassing)().eval()
    with torch.no_grad():
        ref = ot.mab_forward(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, ot.MABWeights.from_state_dict(mp.state_dict()),
                             atom_messages=atom)
    mp = mp.to(gpu_device)
    bmg.to(gpu_device)
    with torch.no_grad():
        out = mp(bmg)
    for got, want in zip(out, ref):
        assert parity_err(got.cpu().numpy(), want.numpy()) <= TOL


@pytest.mark.gpu
def test_inference_with_an_oversize_molecule_is_routed_not_poisoned(gpu_device):
    """A 40-atom molecule does not fit a tile of the whole-forward kernel: the host-side batching code knows and the batch
    takes the per-step kernels; handed over as bare tensors the tile kernel's generic path computes it.  Never NaN."""
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph
    from chemprop_amd.mab import MABBondMessagePassing
    from oracle import dmpnn_torch as ot

    mgs = synth.random_molgraphs(20, "qm9", seed=4) + synth.random_molgraphs(2, "synth40", seed=5)
    bmg = BatchMolGraph(mgs)
    torch.manual_seed(8)
    mp = MABBondMessagePassing(d_h=64).eval()
    with torch.no_grad():
        ref = ot.mab_forward(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, ot.MABWeights.from_state_dict(mp.state_dict()),
                             atom_messages=False)
    mp = mp.to(gpu_device)
    bmg.to(gpu_device)
    with torch.no_grad():
        for i in range(3):
            out = mp(bmg)
            for got, want in zip(out, ref):
                assert torch.isfinite(got).all() and parity_err(got.cpu().numpy(), want.numpy()) <= TOL, i
    # ... and as bare tensors (no host-side size knowledge): the tile kernel's generic path carries the big molecules
    from chemprop_amd.data import BatchMolGraph as B

    bare = B.from_tensors(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, bmg.batch, len(bmg))
    mp2 = MABBondMessagePassing(d_h=64).eval().to(gpu_device)
    mp2.load_state_dict(mp.state_dict())
    with torch.no_grad():
        for i in range(3):
            for got, want in zip(mp2(bare), ref):
                assert torch.isfinite(got).all() and parity_err(got.cpu().numpy(), want.numpy()) <= TOL, i


@pytest.mark.gpu
@pytest.mark.parametrize("atom,n_mols,kw", [
    (False, 512, dict()), (True, 512, dict()),
    (False, 300, dict(activation="elu", bias=True, depth=4, d_h=128)), (True, 300, dict(activation="tanh", bias=True, depth=2, d_h=64)),
    (False, 200, dict(activation="tanh", d_vd=3, d_ed=2)),                  # descriptors: W_vd inside the block's chain, W_ed behind the read-out
    (True, 200, dict(activation="elu", return_edge_embeddings=False)),     # vertex read-out only: the atom block itself
], ids=["bond-512-relu", "atom-512-relu", "bond-300-elu-d4-h128", "atom-300-tanh-d2-h64", "bond-200-vd-ed", "atom-200-vertex-only"])
def test_training_on_the_tile_kernels(atom, n_mols, kw, gpu_device, monkeypatch):
    _mab_training_on_the_tile_kernels(atom, n_mols, kw, gpu_device, monkeypatch)


@pytest.mark.gpu
def test_training_on_the_tile_kernels_with_split_row_products(gpu_device, monkeypatch):
    """The same step with the bond block's messages kept as split rows and every product on k_wgrad16r (what batches from ~900
    molecules on take by themselves): the edge read-out's gradient enters the same backward tile kernel."""
    monkeypatch.setenv("DMPNN_KEEP_ROWS", "1")
    _mab_training_on_the_tile_kernels(False, 300, dict(activation="elu", bias=True, depth=4, d_h=128), gpu_device, monkeypatch)


def _mab_training_on_the_tile_kernels(atom, n_mols, kw, gpu_device, monkeypatch):
    """Round 4 (round-3 VERDICT item 8): a TRAINING step of the mol-atom-bond blocks on the tile kernels.  The block's forward is one
    launch (DMPNN_F_KEEP; DMPNN_F_ATOM for the atom variant); the kept H^(depth-1) is its second output, the edge read-out a row
    kernel under autograd, and that read-out's gradient enters the backward tile kernel beside the vertex one
    (dmpnn_bwd_args.g_edge).  Both read-outs and every gradient against the oracle's autograd (mol_atom_bond.py:266-282), on the
    tile plan (caller's edge order) and on the full plan (CSR-row order: the edge states are permuted on the way out and back)."""
    from chemprop_amd import synth
    from chemprop_amd.mab import MABAtomMessagePassing, MABBondMessagePassing
    from oracle import dmpnn_torch as ot

    monkeypatch.setenv("DMPNN_VALIDATE", "never")
    cls = MABAtomMessagePassing if atom else MABBondMessagePassing
    bmg = synth.random_batch(n_mols, "qm9", seed=14)
    torch.manual_seed(9)
    ref_mp = cls(**kw)
    nV, nE = bmg.V.shape[0], bmg.E.shape[0]
    gen = torch.Generator().manual_seed(3)
    V_d = torch.randn(nV, kw["d_vd"], generator=gen) if kw.get("d_vd") else None
    E_d = torch.randn(nE, kw["d_ed"], generator=gen) if kw.get("d_ed") else None
    w = ot.MABWeights.from_state_dict(dict(ref_mp.named_parameters()))
    ref = ot.mab_forward(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, w, atom_messages=atom, depth=ref_mp.depth,
                         activation=kw.get("activation", "relu"), V_d=V_d, E_d=E_d)
    Gs = [None if r is None else torch.randn(r.shape, generator=gen) for r in ref]
    sum((r * g).sum() for r, g in zip(ref, Gs) if r is not None).backward()
    bmg.to(gpu_device)
    dev = lambda t: None if t is None else t.to(gpu_device)
    res = {}
    for plan_kind in ("tiles", "full"):
        if kw.get("d_vd") and plan_kind == "tiles":
            continue   # (W_vd inside the block: the full plan only)
        monkeypatch.setenv("DMPNN_TRAIN_PLAN", plan_kind)
        mp = cls(**kw)
        mp.load_state_dict(ref_mp.state_dict())
        mp = mp.to(gpu_device).train()
        out = mp(bmg, dev(V_d), dev(E_d))
        assert mp.__dict__.get("_dmpnn_route") == ("mega16/atom" if atom else "mega16"), (plan_kind, mp.__dict__.get("_dmpnn_route"))
        for got, want in zip(out, ref):
            assert (got is None) == (want is None)
            if got is not None:
                assert parity_err(got.detach().cpu().numpy(), want.detach().numpy()) <= TOL, plan_kind
        sum((o * dev(g)).sum() for o, g in zip(out, Gs) if o is not None).backward()
        res[plan_kind] = {k: p.grad.cpu().numpy() for k, p in mp.named_parameters()}
        for k, q in ref_mp.named_parameters():
            err = parity_err(res[plan_kind][k], q.grad.numpy())
            assert err <= 2e-5, f"{plan_kind} {k}: {err:.3e}"
    if len(res) == 2:
        for k in res["full"]:
            assert parity_err(res["tiles"][k], res["full"][k]) <= 5e-6, k


@pytest.mark.gpu
def test_training_with_an_oversize_molecule_as_bare_tensors(gpu_device, monkeypatch):
    """A molecule beyond the tile inside a batch of bare tensors (no host-side size knowledge), trusted plan: the bond variant's tile
    kernels carry it through their generic path — forward, and backward WITH the edge read-out's gradient.  The atom variant has no
    generic path (such a molecule would be NaN there, forward and every gradient, and the NaN loss would reach the optimizer:
    round-4 ADVICE): the host COUNTS the molecule sizes of a foreign batch on the device (``nn.batch_oversize``) and trains such a
    batch on the per-step chain — right numbers, not loud ones."""
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph
    from chemprop_amd.mab import MABAtomMessagePassing, MABBondMessagePassing
    from oracle import dmpnn_torch as ot

    monkeypatch.setenv("DMPNN_VALIDATE", "never")
    mgs = synth.random_molgraphs(20, "qm9", seed=4)
    mgs.insert(7, synth.random_molgraphs(1, "synth40", seed=5)[0])
    b = BatchMolGraph(mgs)
    bare = BatchMolGraph.from_tensors(b.V, b.E, b.edge_index, b.rev_edge_index, b.batch, len(mgs))
    torch.manual_seed(8)
    ref_mp = MABBondMessagePassing(d_h=64, activation="tanh")
    w = ot.MABWeights.from_state_dict(dict(ref_mp.named_parameters()))
    ref = ot.mab_forward(bare.V, bare.E, bare.edge_index, bare.rev_edge_index, w, atom_messages=False, activation="tanh")
    gen = torch.Generator().manual_seed(1)
    Gs = [torch.randn(r.shape, generator=gen) for r in ref]
    sum((r * g).sum() for r, g in zip(ref, Gs)).backward()
    bare.to(gpu_device)
    mp = MABBondMessagePassing(d_h=64, activation="tanh")
    mp.load_state_dict(ref_mp.state_dict())
    mp = mp.to(gpu_device).train()
    out = mp(bare)
    assert mp.__dict__.get("_dmpnn_route") == "mega16"
    sum((o * g.to(gpu_device)).sum() for o, g in zip(out, Gs)).backward()
    for got, want in zip(out, ref):
        assert parity_err(got.detach().cpu().numpy(), want.detach().numpy()) <= TOL
    for (k, p), (_, q) in zip(mp.named_parameters(), ref_mp.named_parameters()):
        assert parity_err(p.grad.cpu().numpy(), q.grad.numpy()) <= 2e-5, k
    from chemprop_amd.nn import batch_oversize

    assert batch_oversize(bare, len(mgs)) is True and batch_oversize(b, len(mgs)) is True
    torch.manual_seed(9)
    ref_a = MABAtomMessagePassing(d_h=64, activation="tanh")
    wa = ot.MABWeights.from_state_dict(dict(ref_a.named_parameters()))
    cpu = BatchMolGraph.from_tensors(b.V, b.E, b.edge_index, b.rev_edge_index, b.batch, len(mgs))
    ref2 = ot.mab_forward(cpu.V, cpu.E, cpu.edge_index, cpu.rev_edge_index, wa, atom_messages=True, activation="tanh")
    sum((r * g).sum() for r, g in zip(ref2, Gs)).backward()
    mpa = MABAtomMessagePassing(d_h=64, activation="tanh")
    mpa.load_state_dict(ref_a.state_dict())
    mpa = mpa.to(gpu_device).train()
    out_a = mpa(bare)
    assert mpa.__dict__.get("_dmpnn_route") == "rows"
    sum((o * g.to(gpu_device)).sum() for o, g in zip(out_a, Gs)).backward()
    for got, want in zip(out_a, ref2):
        assert torch.isfinite(got).all() and parity_err(got.detach().cpu().numpy(), want.detach().numpy()) <= TOL
    for (k, p), (_, q) in zip(mpa.named_parameters(), ref_a.named_parameters()):
        assert parity_err(p.grad.cpu().numpy(), q.grad.numpy()) <= 2e-5, k
