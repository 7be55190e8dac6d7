"""f2 — AtomMessagePassing (chemprop/nn/message_passing/base.py:254-289, mixins.py:21-30): the oracle restatement and
the HIP-kernel mirror against goldens frozen from the executed reference (tests/golden/make_golden_atom.py)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, TOL, parity_err

ATOM = sorted(glob.glob(os.path.join(GOLDEN_DIR, "atom", "*.npz")))


class Case:
    def __init__(self, path):
        z = np.load(path)
        self.arr = {k: z[k] for k in z.files}
        self.meta = json.loads(bytes(self.arr.pop("meta")).decode())
        self.cfg = self.meta["cfg"]

    def __getitem__(self, k):
        return self.arr[k]

    def __contains__(self, k):
        return k in self.arr

    def state_dict(self):
        return {k[2:]: torch.from_numpy(np.array(v)) for k, v in self.arr.items() if k.startswith("w.")}

    def module(self, device="cpu"):
        from chemprop_amd.nn import AtomMessagePassing

        mp = AtomMessagePassing(**self.cfg)
        mp.load_state_dict(self.state_dict())
        return mp.eval().to(device)

    def bmg(self, device="cpu"):
        from chemprop_amd.data import BatchMolGraph

        t = lambda k: torch.from_numpy(self.arr[k])
        b = BatchMolGraph.from_tensors(t("V"), t("E"), t("edge_index"), t("rev_edge_index"), t("batch"), self.meta["n_mols"])
        if device != "cpu":
            b.to(device)
        return b


@pytest.fixture(params=ATOM, ids=[os.path.basename(p)[:-4] for p in ATOM])
def atom_case(request):
    return Case(request.param)


def test_goldens_exist():
    assert len(ATOM) >= 7


def test_oracle_matches_golden(atom_case):
    from oracle import dmpnn_torch as ot

    w = atom_case.state_dict()
    W = ot.MPWeights(W_i=w["W_i.weight"], W_h=w["W_h.weight"], W_o=w["W_o.weight"], b_o=w["W_o.bias"], b_i=w.get("W_i.bias"),
                     b_h=w.get("W_h.bias"), W_d=w.get("W_d.weight"), b_d=w.get("W_d.bias"))
    t = lambda k: torch.from_numpy(atom_case[k])
    cfg = atom_case.cfg
    out, inter = ot.atom_forward(t("V"), t("E"), t("edge_index"), t("rev_edge_index"), W, depth=cfg["depth"],
                                 activation=cfg["activation"], undirected=cfg["undirected"],
                                 V_d=t("V_d") if "V_d" in atom_case else None, return_intermediates=True)
    assert parity_err(out.numpy(), atom_case["out"]) <= 1e-6
    assert parity_err(inter["H0"].numpy(), atom_case["H0"]) <= 1e-6
    if "M1" in atom_case:
        assert parity_err(inter["M"][0].numpy(), atom_case["M1"]) <= 1e-6


def test_mirror_reproduces_the_reference_rng_stream(atom_case):
    """Same constructor order as base.py:278-289: identical initial weights under the same seed."""
    from chemprop_amd.nn import AtomMessagePassing

    torch.manual_seed(atom_case.meta["seed"])
    mp = AtomMessagePassing(**atom_case.cfg)
    sd = atom_case.state_dict()
    assert list(mp.state_dict().keys()) == list(sd.keys())
    for k, v in mp.state_dict().items():
        assert torch.equal(v, sd[k]), k


@pytest.mark.gpu
def test_forward_and_gradients_vs_executed_reference(atom_case, gpu_device):
    mp = atom_case.module(gpu_device)
    bmg = atom_case.bmg(gpu_device)
    V_d = torch.from_numpy(atom_case["V_d"]).to(gpu_device) if "V_d" in atom_case else None
    out = mp(bmg, V_d)
    assert parity_err(out.detach().cpu().numpy(), atom_case["out"]) <= TOL
    (out * torch.from_numpy(atom_case["G"]).to(gpu_device)).sum().backward()
    for k, p in mp.named_parameters():
        ref = atom_case["g." + k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        assert parity_err(got, ref) <= 2e-5, k
    with torch.no_grad():
        # inference takes the whole-forward tile kernel where it applies (round 3: DMPNN_F_ATOM) — another fp32-class arithmetic than
        # the per-step chain above — else the same chain; deterministic either way
        a, b = mp(bmg, V_d), mp(bmg, V_d)
        assert torch.equal(a, b)
        assert parity_err(a.cpu().numpy(), atom_case["out"]) <= TOL
        route = mp.__dict__.get("_dmpnn_route")
        cfg = atom_case.cfg
        shapes_ok = (not cfg["undirected"] and V_d is None and cfg["d_h"] % 4 == 0 and cfg["d_h"] <= 320 and atom_case["E"].shape[1] <= 16
                     and atom_case["E"].shape[1] % 2 == 0 and atom_case["V"].shape[1] % 2 == 0 and atom_case["E"].shape[0] > 0)
        # (shapes that allow the tile kernel take it unless a molecule of the golden exceeds the tile; the others never do)
        assert route in (("mega16/atom", "rows/atom") if shapes_ok else ("rows/atom",)), (route, cfg)


@pytest.mark.gpu
def test_full_size_vs_oracle(gpu_device):
    from chemprop_amd import synth
    from chemprop_amd.nn import AtomMessagePassing
    from oracle import dmpnn_torch as ot

    bmg = synth.random_batch(512, "qm9", seed=77)
    torch.manual_seed(5)
    mp = AtomMessagePassing().eval()
    with torch.no_grad():
        ref = ot.atom_forward(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, ot.MPWeights.from_module(mp))
    mp = mp.to(gpu_device)
    bmg.to(gpu_device)
    with torch.no_grad():
        out = mp(bmg)
    assert parity_err(out.cpu().numpy(), ref.numpy()) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("n_mols,kw", [(512, dict()), (4096, dict(activation="leakyrelu", bias=True)), (300, dict(d_h=128, depth=5, activation="elu")),
                                        (64, dict(d_h=64, depth=1))])
def test_atom_messages_on_the_tile_kernel(n_mols, kw, gpu_device):
    """Inference of AtomMessagePassing as ONE launch per tile of molecules (DMPNN_F_ATOM): the incidence without the reverse-edge
    term, the constant bond-feature half of the message folded into the residual once per tile.  Steady path (tile plan) included."""
    from chemprop_amd import synth
    from chemprop_amd.nn import AtomMessagePassing
    from oracle import dmpnn_torch as ot

    bmg = synth.random_batch(n_mols, "qm9", seed=78)
    torch.manual_seed(6)
    mp = AtomMessagePassing(**kw).eval()
    with torch.no_grad():
        ref = ot.atom_forward(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, ot.MPWeights.from_module(mp), depth=mp.depth,
                              activation=kw.get("activation", "relu"))
    mp = mp.to(gpu_device)
    bmg.to(gpu_device)
    with torch.no_grad():
        for i in range(4):   # (two validated batches on the full plan, then the tile plan)
            out = mp(bmg)
            assert mp.__dict__.get("_dmpnn_route") == "mega16/atom", (i, mp.__dict__.get("_dmpnn_route"))
            assert parity_err(out.cpu().numpy(), ref.numpy()) <= TOL, i
    # a batch with a molecule beyond the tile keeps the per-step chain
    big = synth.random_batch(8, "synth40", seed=1)
    big.to(gpu_device)
    with torch.no_grad():
        o2 = mp(big)
    assert torch.isfinite(o2).all() and mp.__dict__.get("_dmpnn_route") == "rows/atom"
