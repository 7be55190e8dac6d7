import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


class Golden:
    """One frozen case of tests/golden/*.npz (made by tests/golden/make_golden.py from the executed reference)."""

    def __init__(self, path):
        z = np.load(path)
        self.arr = {k: z[k] for k in z.files}
        self.meta = json.loads(bytes(self.arr.pop("meta")).decode())
        self.name = self.meta["name"]
        self.cfg = self.meta["cfg"]

    def __getitem__(self, k):
        if k == "G" and "G" not in self.arr:  # large cases: the cotangent is regenerated, not stored (make_golden.py)
            import torch

            gen = torch.Generator().manual_seed(1000 + self.meta["seed"])
            assert not self.cfg.get("d_vd")
            self.arr["G"] = torch.randn(self.arr["out"].shape, generator=gen).numpy()
        return self.arr[k]

    def __contains__(self, k):
        return k in self.arr

    def weights(self):
        """state_dict of the block as numpy (stored, or regenerated from the seed and sha-checked)."""
        import hashlib

        import torch

        stored = {k[2:]: v for k, v in self.arr.items() if k.startswith("w.")}
        if not stored:
            from chemprop_amd.nn import BondMessagePassing

            torch.manual_seed(self.meta["seed"])
            mp = BondMessagePassing(**self.cfg)
            stored = {k: v.detach().numpy() for k, v in mp.state_dict().items()}
        m = hashlib.sha256()
        for k in sorted(stored):
            m.update(k.encode())
            m.update(np.ascontiguousarray(stored[k], dtype=np.float32).tobytes())
        assert m.hexdigest() == self.meta["weights_sha"], f"{self.name}: weights do not match the golden sha (RNG drift?)"
        return stored

    def module(self, device="cpu"):
        import torch

        from chemprop_amd.nn import BondMessagePassing

        mp = BondMessagePassing(**self.cfg)
        mp.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in self.weights().items()})
        return mp.eval().to(device)

    def bmg(self, device="cpu"):
        import torch

        from chemprop_amd.data import BatchMolGraph

        t = lambda k: torch.from_numpy(self.arr[k])
        b = BatchMolGraph.from_tensors(t("V"), t("E"), t("edge_index"), t("rev_edge_index"), t("batch"), self.meta["n_mols"])
        if device != "cpu":
            b.to(device)
        return b


def golden_paths():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def golden_ids():
    return [os.path.basename(p)[:-4] for p in golden_paths()]


@pytest.fixture(params=golden_paths(), ids=golden_ids())
def golden(request):
    return Golden(request.param)


def parity_err(got, ref):
    """Norm-wise relative error of SURVEY §8(d): max|got-ref| / max(1, max|ref|)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"shape {got.shape} != {ref.shape}"
    if ref.size == 0:
        return 0.0
    return float(np.max(np.abs(got - ref)) / max(1.0, float(np.max(np.abs(ref)))))


def parity_err_unfloored(got, ref):
    """max|got-ref| / max|ref| — the same ratio WITHOUT the floor of 1 in the denominator (round-1 VERDICT: with
    max|ref| < 1 the floored metric is an absolute 1e-5; the at-size tests report and hold both)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if ref.size == 0:
        return 0.0
    m = float(np.max(np.abs(ref)))
    return float(np.max(np.abs(got - ref)) / m) if m > 0 else float(np.max(np.abs(got - ref)))


TOL = 1e-5  # BASELINE.json north_star: "<= 1e-5 relative fp32"


@pytest.fixture(scope="session")
def gpu_device():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("`-m gpu` tests need a HIP device (torch.cuda.is_available() is False); "
                    "they never fall back to CPU")
    from chemprop_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def adam_comparable(arrs, name: str, steps: int):
    """Entries of parameter ``name`` whose golden gradients are NOT fp32 rounding noise around zero (``|g_s| >= 1e-6 max|g_s|`` or exactly 0, in
    every step ``s``): Adam normalises a gradient by its own magnitude, so an entry whose true gradient is 0 and whose computed gradient is
    noise of 1e-8 moves by a full ``lr`` in a direction that is the noise's sign — two correct fp32 implementations (here: the
    restatement and the executed reference, both on the CPU) differ there by 2 lr.  Parameters after Adam steps are compared on the
    other entries; the gradients themselves are compared everywhere."""
    keep = None
    for s in range(steps):
        key = f"g{s}.{name}"
        if key not in arrs:
            return None
        g = np.abs(np.asarray(arrs[key], dtype=np.float64))
        m = (g >= 1e-6 * max(float(g.max()), 1e-30)) | (g == 0.0)     # (an exactly zero gradient is no noise: Adam leaves the entry alone)
        keep = m if keep is None else (keep & m)
    return keep


def parity_err_where(got, ref, mask):
    """``parity_err`` over the entries ``mask`` selects (``None``: all), normalised by the WHOLE reference tensor."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    if ref.size == 0:
        return 0.0
    d = np.abs(got - ref)
    if mask is not None:
        assert mask.mean() > 0.9, "almost every entry must stay in the comparison"
        d = d[mask]
    return float(d.max() / max(1.0, float(np.max(np.abs(ref))))) if d.size else 0.0
