"""f4 — the predictor's feed-forward stack ``chemprop.nn.ffn.MLP`` (chemprop/nn/ffn.py:24-68): oracle restatement and
the HIP-kernel mirror against goldens frozen from the executed reference (tests/golden/make_golden_ffn.py)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, TOL, parity_err

FFN = sorted(glob.glob(os.path.join(GOLDEN_DIR, "ffn", "*.npz")))


class Case:
    def __init__(self, path):
        z = np.load(path)
        self.arr = {k: z[k] for k in z.files}
        self.meta = json.loads(bytes(self.arr.pop("meta")).decode())
        self.cfg = self.meta["cfg"]

    def __getitem__(self, k):
        return self.arr[k]

    def state_dict(self):
        return {k[2:]: torch.from_numpy(np.array(v)) for k, v in self.arr.items() if k.startswith("w.")}

    def module(self, device="cpu"):
        from chemprop_amd.ffn import MLP

        m = MLP.build(**self.cfg)
        m.load_state_dict(self.state_dict())
        return m.eval().to(device)


@pytest.fixture(params=FFN, ids=[os.path.basename(p)[:-4] for p in FFN])
def ffn_case(request):
    return Case(request.param)


def test_goldens_exist():
    assert len(FFN) >= 6


def test_oracle_matches_golden(ffn_case):
    from oracle import ffn_torch as of

    sd = ffn_case.state_dict()
    Ws = [v for k, v in sd.items() if k.endswith("weight")]
    bs = [v for k, v in sd.items() if k.endswith("bias")]
    out = of.mlp_forward(torch.from_numpy(ffn_case["X"]), Ws, bs, ffn_case.cfg.get("activation", "relu"))
    assert parity_err(out.numpy(), ffn_case["out"]) <= 1e-6


def test_mirror_reproduces_the_reference_rng_stream(ffn_case):
    """Same construction order as ffn.py:37-58: identical initial weights and state_dict keys under the same seed."""
    from chemprop_amd.ffn import MLP

    torch.manual_seed(ffn_case.meta["seed"])
    m = MLP.build(**ffn_case.cfg)
    sd = ffn_case.state_dict()
    assert list(m.state_dict().keys()) == list(sd.keys())
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert m.input_dim == ffn_case.cfg["input_dim"] and m.output_dim == ffn_case.cfg["output_dim"]


def test_fails_loudly_off_device(ffn_case):
    with pytest.raises(RuntimeError):
        ffn_case.module()(torch.from_numpy(ffn_case["X"]))


@pytest.mark.gpu
def test_forward_and_gradients_vs_executed_reference(ffn_case, gpu_device):
    m = ffn_case.module(gpu_device)
    X = torch.from_numpy(ffn_case["X"]).to(gpu_device).requires_grad_(True)
    out = m(X)  # grad enabled: the autograd wrappers around the same kernels
    assert parity_err(out.detach().cpu().numpy(), ffn_case["out"]) <= TOL
    (out * torch.from_numpy(ffn_case["G"]).to(gpu_device)).sum().backward()
    assert parity_err(X.grad.cpu().numpy(), ffn_case["gX"]) <= 2e-5
    for k, p in m.named_parameters():
        assert parity_err(p.grad.cpu().numpy(), ffn_case["g." + k]) <= 2e-5, k
    with torch.no_grad():  # inference: one launch per layer, activation fused into the previous layer's epilogue
        assert parity_err(m(X.detach()).cpu().numpy(), ffn_case["out"]) <= TOL


@pytest.mark.gpu
def test_dropout_and_custom_activation_take_the_module_route(gpu_device):
    """Active dropout / a user nn.Module activation run as the torch modules themselves between the kernels."""
    from chemprop_amd.ffn import MLP

    torch.manual_seed(3)
    m = MLP.build(24, 3, hidden_dim=16, n_layers=2, dropout=0.5, activation=torch.nn.Softplus()).to(gpu_device)
    X = torch.randn(11, 24, device=gpu_device)
    m.eval()
    with torch.no_grad():
        a = m(X)
        ref = torch.nn.Sequential.forward(m.cpu(), X.cpu())  # torch's own CPU ops on the same parameters
    assert parity_err(a.cpu().numpy(), ref.numpy()) <= TOL
    m.to(gpu_device).train()
    torch.manual_seed(5)
    b1 = m(X)
    torch.manual_seed(5)
    b2 = m(X)
    assert torch.equal(b1, b2) and not torch.equal(b1.detach(), a)
