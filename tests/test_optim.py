"""f4: the fused optimizer step (``chemprop_amd/optim.py``, ``csrc/dmpnn_optim.hip``) against ``torch.optim.Adam`` — what
chemprop's training loop uses (``models/model.py:208-231``)."""
import copy

import pytest
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_flat_adam_equals_torch_adam(wd, gpu_device):
    from chemprop_amd import distributed as ddp
    from chemprop_amd.nn import BondMessagePassing
    from chemprop_amd.optim import FlatAdam

    torch.manual_seed(0)
    a = BondMessagePassing(d_h=64, depth=2, bias=True).to(gpu_device)
    b = copy.deepcopy(a)
    ref = torch.optim.Adam(b.parameters(), lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    sync = ddp.GradSync(list(a.parameters()), modules=[a])
    opt = FlatAdam(sync, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    g = torch.Generator(device=gpu_device).manual_seed(1)
    for step in range(6):
        lr = 3e-3 * (1.0 + 0.1 * step)           # (a schedule changes the rate every step)
        for pa, pb in zip(a.parameters(), b.parameters()):
            grad = torch.randn(pb.shape, device=gpu_device, generator=g)
            pb.grad = grad.clone()
            pa.grad.copy_(grad)                  # (the views of the flat buffer)
        for grp in ref.param_groups:
            grp["lr"] = lr
        ref.step()
        v0 = [p._version for p in a.parameters()]
        opt.step(lr=lr)
        assert all(p._version > v for p, v in zip(a.parameters(), v0))   # the engine's weight caches key on this
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            torch.testing.assert_close(pa, pb, rtol=2e-6, atol=2e-7, msg=lambda m: f"step {step} {n}: {m}")


@pytest.mark.gpu
def test_training_with_the_fused_step_follows_torch_adam(gpu_device):
    """The whole step on the engine — forward, backward into the flat gradient buffer, fused Adam — against the same model
    trained by torch.optim.Adam through ordinary ``.grad`` tensors: same loss curve."""
    from chemprop_amd import distributed as ddp, synth
    from chemprop_amd.nn import BondMessagePassing
    from chemprop_amd.optim import FlatAdam

    torch.manual_seed(2)
    bmg = synth.random_batch(48, "qm9", seed=3)
    bmg.to(gpu_device)
    a = BondMessagePassing(d_h=64, depth=3).to(gpu_device).train()
    b = copy.deepcopy(a)
    target = torch.randn(bmg.V.shape[0], 64, device=gpu_device)
    ref = torch.optim.Adam(b.parameters(), lr=1e-3)
    sync = ddp.GradSync(list(a.parameters()), modules=[a])
    opt = FlatAdam(sync, lr=1e-3)
    la, lb = [], []
    for _ in range(8):
        ref.zero_grad()
        loss_b = ((b(bmg) - target) ** 2).mean()
        loss_b.backward()
        ref.step()
        opt.zero_grad()
        loss_a = ((a(bmg) - target) ** 2).mean()
        loss_a.backward()
        sync.allreduce()
        opt.step()
        la.append(float(loss_a.detach())); lb.append(float(loss_b.detach()))
    assert lb[-1] < lb[0]
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-4 * max(1.0, abs(y)), (la, lb)
    # and the trained weights serve inference at once (the pre-split cache follows the bumped versions)
    a.eval(); b.eval()
    with torch.no_grad():
        torch.testing.assert_close(a(bmg), b(bmg), rtol=5e-4, atol=5e-5)


@pytest.mark.gpu
def test_flat_adam_checkpoint_round_trip(gpu_device):
    """state_dict / load_state_dict (the reference's checkpoints carry the optimizer state): a resumed run continues bit for bit;
    torch_state / load_torch_state exchange the moments with ``torch.optim.Adam``."""
    from chemprop_amd import distributed as ddp
    from chemprop_amd.nn import BondMessagePassing
    from chemprop_amd.optim import FlatAdam

    def make():
        torch.manual_seed(0)
        m = BondMessagePassing(d_h=32, depth=2, bias=True).to(gpu_device)
        s = ddp.GradSync(list(m.parameters()), modules=[m])
        return m, s, FlatAdam(s, lr=2e-3, betas=(0.9, 0.98))

    g = torch.Generator(device=gpu_device).manual_seed(3)
    grads = [[torch.randn(p.shape, device=gpu_device, generator=g) for p in make()[0].parameters()] for _ in range(5)]

    def run(m, opt, steps):
        for gs in steps:
            for p, gr in zip(m.parameters(), gs):
                p.grad.copy_(gr)
            opt.step()

    a, _, oa = make()
    run(a, oa, grads)                              # five steps straight
    b, _, ob = make()
    run(b, ob, grads[:3])                          # three steps, checkpoint, a fresh process, two more
    ck = {"model": {k: v.clone() for k, v in b.state_dict().items()}, "opt": ob.state_dict()}
    c, _, oc = make()
    c.load_state_dict(ck["model"])
    oc.load_state_dict(ck["opt"])
    assert oc.steps == 3
    run(c, oc, grads[3:])
    for (n, pa), pc in zip(a.named_parameters(), c.parameters()):
        assert torch.equal(pa, pc), n
    with pytest.raises(ValueError):
        wrong = BondMessagePassing(d_h=16, depth=2).to(gpu_device)
        FlatAdam(ddp.GradSync(list(wrong.parameters()))).load_state_dict(ck["opt"])
    # moments to torch.optim.Adam and back
    ts = oa.torch_state()
    assert float(ts[0]["step"]) == 5 and ts[0]["exp_avg"].shape == next(a.parameters()).shape
    d, _, od = make()
    od.load_torch_state(ts)
    assert od.steps == 5 and torch.equal(od.m, oa.m) and torch.equal(od.v, oa.v)
