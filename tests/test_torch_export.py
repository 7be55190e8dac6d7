"""``torch.export`` of a model that holds the engine's block (round-1 VERDICT "What's missing" 7).

Mirrors the reference's ``tests/integration/test_export.py:17-46``: export on one batch with dynamic ``num_atoms`` /
``num_edges``, run the exported program on a batch of other sizes — ``E == 0`` included — and compare with eager.  The
exported program must contain the engine's operator (no tensor-op fallback), so running it needs the GPU.
"""
import pytest
import torch
from torch import nn


def _register_pytree():
    """What the reference's ``batch_mol_graph_pytree`` fixture (``tests/conftest.py:20-53``) does for its own class."""
    from torch.utils import _pytree as pytree

    from chemprop_amd.data import BatchMolGraph

    if BatchMolGraph in pytree.SUPPORTED_NODES:
        return

    def flatten(b):
        return [b.V, b.E, b.edge_index, b.rev_edge_index, b.batch], len(b)

    def unflatten(children, n_mols):
        return BatchMolGraph.from_tensors(*children, n_mols)

    names = ("V", "E", "edge_index", "rev_edge_index", "batch")

    def flatten_with_keys(b):
        children, ctx = flatten(b)
        return [(pytree.GetAttrKey(k), c) for k, c in zip(names, children)], ctx

    pytree.register_pytree_node(BatchMolGraph, flatten, unflatten, serialized_type_name="chemprop_amd.data.BatchMolGraph",
                                flatten_with_keys_fn=flatten_with_keys)


class _Model(nn.Module):
    """Encoder + sum over the atoms of a molecule + linear head (the shape of ``MPNN.forward``, ``models/model.py:126-146``)."""

    def __init__(self, mp, n_out=1):
        super().__init__()
        self.message_passing = mp
        self.head = nn.Linear(mp.output_dim, n_out)

    def forward(self, bmg):
        H = self.message_passing(bmg)
        pooled = torch.zeros(len(bmg), H.shape[1], dtype=H.dtype, device=H.device).index_add_(0, bmg.batch, H)
        return self.head(pooled)


def _export(model, bmg):
    num_atoms, num_edges = torch.export.Dim("num_atoms"), torch.export.Dim("num_edges")
    shapes = {"bmg": [{0: num_atoms}, {0: num_edges}, {1: num_edges}, {0: num_edges}, {0: num_atoms}]}
    return torch.export.export(model, (bmg,), dynamic_shapes=shapes, strict=False)


def test_export_records_the_engine_operator():
    """CPU: tracing needs no device — the fake implementation gives the shapes; the graph holds ONE engine node."""
    from chemprop_amd import synth
    from chemprop_amd.nn import BondMessagePassing

    _register_pytree()
    torch.manual_seed(0)
    model = _Model(BondMessagePassing(d_h=64, depth=3)).eval()
    ep = _export(model, synth.random_batch(4, "qm9", seed=1))
    targets = [str(n.target) for n in ep.graph.nodes if n.op == "call_function"]
    assert sum("chemprop_amd.bond_message_passing" in t for t in targets) == 1, targets
    assert not any("index_select" in t or "scatter" in t for t in targets)  # no tensor-op restatement of the block in the graph
    out_spec = [n for n in ep.graph.nodes if n.op == "output"][0]
    assert out_spec is not None


@pytest.mark.gpu
@pytest.mark.parametrize("act,bias", [("relu", False), ("prelu", True)])
def test_exported_program_runs_the_kernels_on_other_sizes(act, bias, gpu_device):
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph
    from chemprop_amd.nn import BondMessagePassing
    from oracle import dmpnn_torch as ot

    _register_pytree()
    torch.manual_seed(1)
    mp = BondMessagePassing(depth=3, activation=act, bias=bias)
    model = _Model(mp).eval().to(gpu_device)
    export_b = synth.random_batch(6, "qm9", seed=2)
    export_b.to(gpu_device)
    ep = _export(model, export_b)
    run = ep.module()
    # (the number of molecules is part of the pytree context — as in the reference's fixture — so every batch has 6 of them;
    #  atoms and edges are the dynamic sizes)
    for b in (synth.random_batch(6, "qm9", seed=3), synth.random_batch(6, "zinc", seed=4), synth.random_batch(6, "synth40", seed=5),
              BatchMolGraph([synth.random_molgraph(__import__("numpy").random.default_rng(0), n_atoms=1) for _ in range(6)])):  # E == 0
        with torch.no_grad():
            mp.cpu()
            ref_h = ot.forward_bmg(b, ot.MPWeights.from_module(mp), depth=3, activation=act,
                                   prelu_weight=mp.tau.weight.detach() if act == "prelu" else None)
        model.to(gpu_device)
        b.to(gpu_device)
        with torch.inference_mode():
            eager = model(b)
            got = run(b)
        torch.testing.assert_close(got, eager)
        pooled = torch.zeros(len(b), ref_h.shape[1]).index_add_(0, b.batch.cpu(), ref_h)
        want = pooled @ model.head.weight.detach().cpu().T + model.head.bias.detach().cpu()
        torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_chain_graphs_of_the_reference_unit_test(gpu_device):
    """``tests/unit/nn/test_message_passing.py:29-49`` for the bond encoder: exported on the 3-atom chain, run on the 5-atom
    chain — all-ones features, BLOCK edge layout (all forward edges, then all reverse edges: ``rev[e] != e ^ 1``)."""
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph
    from chemprop_amd.nn import BondMessagePassing
    from oracle import dmpnn_torch as ot

    _register_pytree()
    torch.manual_seed(3)
    mp = BondMessagePassing().eval()
    w = ot.MPWeights.from_module(mp)
    export_graph, inference_graph = BatchMolGraph([synth.chain_molgraph(3)]), BatchMolGraph([synth.chain_molgraph(5)])
    with torch.no_grad():
        want = ot.forward_bmg(inference_graph, w, depth=3)
    mp = mp.to(gpu_device)
    export_graph.to(gpu_device)
    inference_graph.to(gpu_device)
    num_atoms, num_edges = torch.export.Dim("num_atoms", min=2), torch.export.Dim("num_edges", min=2)
    shapes = {"bmg": [{0: num_atoms}, {0: num_edges}, {1: num_edges}, {0: num_edges}, {0: num_atoms}]}
    exported = torch.export.export(mp, (export_graph,), dynamic_shapes=shapes, strict=False)
    with torch.inference_mode():
        expected = mp(inference_graph)
        actual = exported.module()(inference_graph)
    torch.testing.assert_close(actual, expected)
    torch.testing.assert_close(actual.cpu(), want, rtol=1e-5, atol=1e-5)
