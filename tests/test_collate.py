"""f3 — batching (chemprop/data/collate.py:37-62): the one-buffer wire format of chemprop_amd/data.py and the device
kernel ``dmpnn_collate`` against the oracle restatement and against the index tensors the EXECUTED reference
``BatchMolGraph`` produced (frozen in tests/golden/mab/*.npz, whose meta records the molecule lists).  Integer work:
bit-exact."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

MAB = sorted(glob.glob(os.path.join(GOLDEN_DIR, "mab", "*.npz")))
KEYS = ("V", "E", "edge_index", "rev_edge_index", "batch")


def _golden(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    return {k: z[k] for k in KEYS}, meta


def _molgraphs(meta):
    from chemprop_amd import synth

    n, kind, seed = meta["graphs"]
    return synth.random_molgraphs(n, kind, seed=seed)


def _same(got: dict, want: dict):
    for k in KEYS:
        g, w = np.asarray(got[k]), np.asarray(want[k])
        assert g.dtype == w.dtype and g.shape == w.shape, (k, g.dtype, w.dtype, g.shape, w.shape)
        assert np.array_equal(g, w), k


@pytest.mark.parametrize("path", MAB, ids=[os.path.basename(p)[:-4] for p in MAB])
def test_oracle_and_wire_format_vs_executed_reference(path):
    from chemprop_amd.data import BatchMolGraph, PackedBatch
    from oracle import collate_numpy as oc

    want, meta = _golden(path)
    mgs = _molgraphs(meta)
    _same(oc.collate(mgs), want)                                     # the restatement is pinned
    pb = PackedBatch(mgs)
    assert (pb.n_mols, pb.n_atoms, pb.n_edges) == (meta["n_mols"], want["V"].shape[0], want["E"].shape[0])
    _same(oc.unpack_wire(pb.buf.numpy()), want)                      # what dmpnn_collate must make of these bytes
    host = BatchMolGraph(mgs)                                        # the host mirror batches the same way
    _same({k: getattr(host, k).numpy() for k in KEYS}, want)


def _odd_molgraphs():
    """Edge cases of the batching: a lone atom (no bonds), an empty molecule, a two-atom molecule, a ring."""
    from chemprop_amd.data import MolGraph

    f = lambda n, d: np.arange(n * d, dtype=np.float32).reshape(n, d) / 7
    e = lambda pairs: (np.array([[a for a, b in pairs] + [b for a, b in pairs], [b for a, b in pairs] + [a for a, b in pairs]], dtype=np.int64),
                       np.concatenate([np.arange(len(pairs)) + len(pairs), np.arange(len(pairs))]).astype(np.int64))
    out = []
    for n, pairs in ((1, []), (0, []), (2, [(0, 1)]), (5, [(0, 1), (1, 2), (2, 3), (3, 4), (4, 0)]), (1, []), (3, [(0, 2), (2, 1)])):
        ei, rev = e(pairs) if pairs else (np.zeros((2, 0), np.int64), np.zeros(0, np.int64))
        out.append(MolGraph(V=f(n, 4), E=f(2 * len(pairs), 3), edge_index=ei, rev_edge_index=rev))
    return out


def test_wire_format_edge_cases():
    from chemprop_amd.data import PackedBatch
    from oracle import collate_numpy as oc

    mgs = _odd_molgraphs()
    pb = PackedBatch(mgs)
    _same(oc.unpack_wire(pb.buf.numpy()), oc.collate(mgs))
    assert len(pb) == 6 and pb.buf.numel() % 16 == 0
    empty = PackedBatch([])
    assert (empty.n_mols, empty.n_atoms, empty.n_edges) == (0, 0, 0)
    with pytest.raises(RuntimeError):  # batching is a HIP kernel: no host fallback behind to_device
        pb.to_device("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("path", MAB[:4], ids=[os.path.basename(p)[:-4] for p in MAB[:4]])
def test_device_collate_vs_executed_reference(path, gpu_device):
    from chemprop_amd.data import PackedBatch

    want, meta = _golden(path)
    bmg = PackedBatch(_molgraphs(meta), pin=True).to_device(gpu_device)
    assert len(bmg) == meta["n_mols"]
    _same({k: getattr(bmg, k).cpu().numpy() for k in KEYS}, want)


@pytest.mark.gpu
def test_device_collate_edge_cases(gpu_device):
    from chemprop_amd.data import PackedBatch
    from oracle import collate_numpy as oc

    mgs = _odd_molgraphs()
    bmg = PackedBatch(mgs).to_device(gpu_device)
    _same({k: getattr(bmg, k).cpu().numpy() for k in KEYS}, oc.collate(mgs))
    e = PackedBatch([]).to_device(gpu_device)
    assert e.V.shape[0] == 0 and e.edge_index.shape == (2, 0) and e.batch.numel() == 0 and len(e) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_mols,kind", [(512, "qm9"), (4096, "zinc")])
def test_device_collate_full_size_and_forward(n_mols, kind, gpu_device):
    """BASELINE sizes: bit-exact against the oracle, and the block fed from the packed batch gives the same bits as
    the block fed from the host-built batch."""
    from chemprop_amd import synth
    from chemprop_amd.data import BatchMolGraph, PackedBatch
    from chemprop_amd.nn import BondMessagePassing
    from oracle import collate_numpy as oc

    mgs = synth.random_molgraphs(n_mols, kind, seed=11)
    bmg = PackedBatch(mgs, pin=True).to_device(gpu_device)
    _same({k: getattr(bmg, k).cpu().numpy() for k in KEYS}, oc.collate(mgs))
    host = BatchMolGraph(mgs)
    host.to(gpu_device)
    torch.manual_seed(0)
    mp = BondMessagePassing().eval().to(gpu_device)
    with torch.no_grad():
        assert torch.equal(mp(bmg), mp(host))
