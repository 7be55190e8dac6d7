"""Build container only (needs /root/reference): the restatements of the path's neighbours — atom messages, the
mol-atom-bond blocks, the aggregations, the feed-forward stack, the batching — against the EXECUTED reference classes on
randomized configurations.  Together with the frozen goldens (which travel to the GPU box) this is what pins `oracle/`."""
import numpy as np
import pytest
import torch

from conftest import parity_err
from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not present (GPU box)")

ACTS = ["relu", "leakyrelu", "tanh", "elu"]


def _case(seed):
    from chemprop_amd import synth

    rng = np.random.default_rng(7000 + seed)
    kind = ["qm9", "zinc", "cgr"][seed % 3]
    layout = ["interleaved", "block", "shuffled"][int(rng.integers(0, 3))]
    kw = dict(d_h=int(rng.choice([8, 20, 48])), depth=int(rng.integers(1, 5)), bias=bool(rng.integers(0, 2)),
              activation=ACTS[int(rng.integers(0, 4))], undirected=bool(rng.integers(0, 2)) and layout != "shuffled")
    if kind == "cgr":
        kw.update(d_v=106, d_e=28)
    mgs = synth.random_molgraphs(int(rng.integers(1, 10)), kind, seed=300 + seed, layout=layout)
    return rng, mgs, kw


@pytest.mark.parametrize("seed", range(8))
def test_atom_messages(seed):
    from oracle import dmpnn_torch as ot

    ref_shim.install()
    from chemprop.data.collate import BatchMolGraph
    from chemprop.nn.message_passing.base import AtomMessagePassing

    rng, mgs, kw = _case(seed)
    bmg = BatchMolGraph(mgs)
    torch.manual_seed(seed)
    mp = AtomMessagePassing(**kw).eval()
    with torch.no_grad():
        ref = mp(bmg)
        out = ot.atom_forward(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, ot.MPWeights.from_module(mp), depth=mp.depth,
                              activation=kw["activation"], undirected=mp.undirected)
    assert parity_err(out.numpy(), ref.numpy()) <= 1e-6, kw


@pytest.mark.parametrize("seed", range(10))
def test_mol_atom_bond_blocks(seed):
    from oracle import dmpnn_torch as ot

    ref_shim.install()
    from chemprop.data.collate import BatchMolGraph
    from chemprop.nn.message_passing.mol_atom_bond import MABAtomMessagePassing, MABBondMessagePassing

    rng, mgs, kw = _case(seed)
    atom = bool(seed % 2)
    d_vd, d_ed = int(rng.integers(0, 2)) * 2, int(rng.integers(0, 2)) * 3
    which = int(rng.integers(0, 3))  # both read-outs, vertices only, edges only
    kw.update(d_vd=d_vd or None, d_ed=d_ed or None, return_vertex_embeddings=which != 2, return_edge_embeddings=which != 1)
    bmg = BatchMolGraph(mgs)
    torch.manual_seed(seed)
    mp = (MABAtomMessagePassing if atom else MABBondMessagePassing)(**kw).eval()
    g = torch.Generator().manual_seed(seed)
    V_d = torch.randn(bmg.V.shape[0], d_vd, generator=g) if d_vd and which != 2 else None
    E_d = torch.randn(bmg.E.shape[0], d_ed, generator=g) if d_ed and which != 1 else None
    with torch.no_grad():
        ref = mp(bmg, V_d, E_d)
        out = ot.mab_forward(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, ot.MABWeights.from_state_dict(mp.state_dict()),
                             atom_messages=atom, depth=mp.depth, activation=kw["activation"], undirected=mp.undirected,
                             V_d=V_d, E_d=E_d)
    for got, want in zip(out, ref):
        assert (got is None) == (want is None)
        if got is not None:
            assert parity_err(got.numpy(), want.numpy()) <= 1e-6, kw


@pytest.mark.parametrize("seed", range(6))
def test_aggregations_and_ffn_and_batching(seed):
    from oracle import agg_torch as oa
    from oracle import collate_numpy as oc
    from oracle import ffn_torch as of

    ref_shim.install()
    from chemprop.data.collate import BatchMolGraph
    from chemprop.nn import agg as ragg
    from chemprop.nn.ffn import MLP

    rng, mgs, _ = _case(seed)
    bmg = BatchMolGraph(mgs)
    want = {k: getattr(bmg, k).numpy() for k in ("V", "E", "edge_index", "rev_edge_index", "batch")}
    got = oc.collate(mgs)
    for k, v in want.items():
        assert got[k].dtype == v.dtype and np.array_equal(got[k], v), k
    h = int(rng.choice([5, 16, 33]))
    H = torch.randn(bmg.V.shape[0], h, generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        assert torch.equal(oa.mean(H, bmg.batch), ragg.MeanAggregation()(H, bmg.batch))
        assert torch.equal(oa.sum_(H, bmg.batch), ragg.SumAggregation()(H, bmg.batch))
        assert torch.equal(oa.norm(H, bmg.batch, 37.0), ragg.NormAggregation(norm=37.0)(H, bmg.batch))
        torch.manual_seed(seed)
        att = ragg.AttentiveAggregation(output_size=h)
        assert parity_err(oa.attentive(H, bmg.batch, att.W.weight, att.W.bias).numpy(), att(H, bmg.batch).numpy()) <= 1e-6
        act = ACTS[seed % 4]
        torch.manual_seed(seed)
        mlp = MLP.build(h, int(rng.integers(1, 5)), hidden_dim=int(rng.choice([7, 32])), n_layers=int(rng.integers(0, 3)), activation=act).eval()
        sd = mlp.state_dict()
        out = of.mlp_forward(H, [v for k, v in sd.items() if k.endswith("weight")], [v for k, v in sd.items() if k.endswith("bias")], act)
        assert parity_err(out.numpy(), mlp(H).numpy()) <= 1e-6
