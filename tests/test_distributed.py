"""N > 1 path on CPU: world_size-2 gloo.  Sharding is by molecule, the forward needs no collective,
and the ONE collective of a training step (flat gradient all-reduce) reproduces the single-process
gradient of the union batch — because molecules are independent the gradient is additive."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chemprop_amd import distributed as ddp
from chemprop_amd import synth
from chemprop_amd.data import BatchMolGraph
from chemprop_amd.nn import BondMessagePassing
from oracle import dmpnn_torch as ot


def test_hash_partition_is_a_partition():
    n, world = 1003, 8
    shards = [ddp.hash_partition(n, r, world, equalize=False) for r in range(world)]
    allidx = np.concatenate(shards)
    assert len(allidx) == n and len(np.unique(allidx)) == n
    sizes = [len(s) for s in shards]
    assert max(sizes) - min(sizes) < 0.25 * n / world  # hash is well mixed
    eq = [ddp.hash_partition(n, r, world) for r in range(world)]
    assert len({len(s) for s in eq}) == 1 and len(eq[0]) == min(sizes)
    # deterministic, sampler-independent
    assert np.array_equal(ddp.hash_partition(n, 3, world), ddp.hash_partition(n, 3, world))
    with pytest.raises(ValueError):
        ddp.hash_partition(10, 8, 8)


def test_weighted_partition_balances_edges():
    mgs = synth.random_molgraphs(400, "zinc", seed=4)
    w = [m.edge_index.shape[1] for m in mgs]
    loads = []
    for r in range(8):
        idx = ddp.hash_partition(len(mgs), r, 8, equalize=False, weights=w)
        loads.append(sum(w[i] for i in idx))
    assert (max(loads) - min(loads)) / np.mean(loads) < 0.05


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_mols, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    mgs = synth.random_molgraphs(n_mols, "qm9", seed=0)
    torch.manual_seed(100 + rank)  # deliberately different initial weights per rank ...
    mp_ = BondMessagePassing(d_h=48)
    ddp.broadcast_params(mp_)      # ... made identical by the broadcast
    idx = ddp.hash_partition(n_mols, rank, world, equalize=False)
    bmg = BatchMolGraph([mgs[i] for i in idx])
    w = ot.MPWeights(mp_.W_i.weight, mp_.W_h.weight, mp_.W_o.weight, mp_.W_o.bias)
    out = ot.forward_bmg(bmg, w, depth=mp_.depth)  # CPU stand-in for the engine (it has no CPU path)
    # per-molecule cotangent so the union loss is well defined whatever the shard
    g = torch.stack([torch.full((48,), float(i % 7) - 3.0) for i in idx])[bmg.batch]
    (out * g).sum().backward()
    ddp.allreduce_grads(list(mp_.parameters()))
    torch.save({k: p.grad.clone() for k, p in mp_.named_parameters()} | {"w0": mp_.W_h.weight.detach().clone()},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_union_batch(tmp_path):
    world, n_mols = 2, 24
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_mols, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k  # replicas agree bit-for-bit after the all-reduce
    # single-process reference on the union batch, with rank 0's (broadcast) weights
    mgs = synth.random_molgraphs(n_mols, "qm9", seed=0)
    torch.manual_seed(100)
    ref = BondMessagePassing(d_h=48)
    assert torch.equal(ref.W_h.weight, r0["w0"])
    bmg = BatchMolGraph(mgs)
    w = ot.MPWeights(ref.W_i.weight, ref.W_h.weight, ref.W_o.weight, ref.W_o.bias)
    out = ot.forward_bmg(bmg, w, depth=ref.depth)
    g = torch.stack([torch.full((48,), float(i % 7) - 3.0) for i in range(n_mols)])[bmg.batch]
    (out * g).sum().backward()
    for k, p in ref.named_parameters():
        err = float((p.grad - r0[k]).abs().max() / max(1.0, float(p.grad.abs().max())))
        assert err <= 1e-5, (k, err)


def _worker_sync(rank, world, port, n_mols, out_dir):
    """The same step through GradSync (flat buffer, asynchronous exchange) and through the sequential allreduce_grads; a
    parameter without a gradient on one rank only (frozen W_d branch unused there) must not dead-lock or shift the buffer."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    mgs = synth.random_molgraphs(n_mols, "qm9", seed=1)
    torch.manual_seed(7)
    mp_ = BondMessagePassing(d_h=32, d_vd=3, bias=True)
    idx = ddp.hash_partition(n_mols, rank, world, equalize=False)
    bmg = BatchMolGraph([mgs[i] for i in idx])
    params = list(mp_.parameters())

    def backward():
        w = ot.MPWeights(mp_.W_i.weight, mp_.W_h.weight, mp_.W_o.weight, mp_.W_o.bias, mp_.W_i.bias, mp_.W_h.bias,
                         mp_.W_d.weight, mp_.W_d.bias)
        V_d = torch.ones(bmg.V.shape[0], 3) if rank == 0 else None     # rank 1 never touches W_d: its gradient stays None
        out = ot.forward_bmg(bmg, w, depth=mp_.depth, V_d=V_d)
        (out * (1.0 + rank)).sum().backward()

    for p in params:
        p.grad = None
    backward()
    ddp.allreduce_grads(params)
    seq = [p.grad.clone() for p in params]
    sync = ddp.GradSync(params, modules=[mp_])
    for rep in range(2):      # twice: the views survive a step, the second exchange waits for the first
        sync.zero_grad()
        backward()
        sync.allreduce()
        sync.wait()
        for p, v, want in zip(params, sync.views, seq):
            assert p.grad.data_ptr() == v.data_ptr()
            assert torch.allclose(p.grad, want, rtol=0, atol=0), rep
    # optimizer.zero_grad(set_to_none=True) detaches the parameters; the next exchange folds the stray gradients back in
    for p in params:
        p.grad = None
    backward()
    sync.allreduce(); sync.wait()
    for p, want in zip(params, seq):
        assert torch.equal(p.grad, want)
    torch.save({"ok": True}, os.path.join(out_dir, f"sync{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _worker_sliced(rank, world, port, n_mols, out_dir):
    """A step that exchanges its gradients in TWO slices, each as soon as it is final (model.FusedTrainer at N > 1: the head's
    while the block's backward still runs), against the sequential all-reduce of everything."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    mgs = synth.random_molgraphs(n_mols, "qm9", seed=3)
    torch.manual_seed(5)
    mp_ = BondMessagePassing(d_h=40, bias=True)
    head = torch.nn.Sequential(torch.nn.Linear(40, 24), torch.nn.ReLU(), torch.nn.Linear(24, 2))
    idx = ddp.hash_partition(n_mols, rank, world, equalize=False)
    bmg = BatchMolGraph([mgs[i] for i in idx])
    params = list(mp_.parameters()) + list(head.parameters())

    def backward():
        w = ot.MPWeights(mp_.W_i.weight, mp_.W_h.weight, mp_.W_o.weight, mp_.W_o.bias, mp_.W_i.bias, mp_.W_h.bias)
        out = head(ot.forward_bmg(bmg, w, depth=mp_.depth))
        (out * (1.0 + rank)).sum().backward()

    for p in params:
        p.grad = None
    backward()
    ddp.allreduce_grads(params)
    seq = [p.grad.clone() for p in params]
    sync = ddp.GradSync(params, modules=[mp_])
    lo_h, hi_h = sync.range_of(list(head.parameters()))
    lo_b, hi_b = sync.range_of(list(mp_.parameters()))
    assert (lo_b, hi_h) == (0, sync.flat.numel()) and hi_b == lo_h          # two adjacent slices cover the buffer
    with pytest.raises(ValueError):
        sync.range_of([params[0], params[-1]])                              # not a contiguous run
    for rep in range(2):
        sync.zero_grad()
        backward()
        sync.allreduce(lo_h, hi_h)      # the head's slice first ...
        sync.allreduce(lo_b, hi_b)      # ... then the block's: two collectives in flight
        assert len(sync.works) == 2
        sync.wait()
        assert not sync.works
        for p, want in zip(params, seq):
            assert torch.equal(p.grad, want), rep
    torch.save({"ok": True}, os.path.join(out_dir, f"sliced{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sync_sliced_exchange_equals_sequential(tmp_path):
    world = 2
    mp.spawn(_worker_sliced, args=(world, _free_port(), 14, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "sliced0.pt").exists() and (tmp_path / "sliced1.pt").exists()


def test_hash_partition_padding_reports_the_own_length():
    n, world = 5, 8      # more ranks than molecules: some shards are empty and must still be padded to the common length
    got = [ddp.hash_partition(n, r, world, pad=True, return_own=True) for r in range(world)]
    assert len({len(idx) for idx, _ in got}) == 1 and len(got[0][0]) >= 1
    own = np.concatenate([idx[:k] for idx, k in got])
    assert sorted(own.tolist()) == list(range(n))                     # every molecule owned exactly once ...
    assert any(k == 0 for _, k in got)                                # ... although some ranks own none (all padding)


def test_grad_sync_equals_sequential_allreduce(tmp_path):
    world = 2
    mp.spawn(_worker_sync, args=(world, _free_port(), 16, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "sync0.pt").exists() and (tmp_path / "sync1.pt").exists()


def test_backward_on_calling_thread_is_scoped_and_changes_no_gradient():
    """The training-loop context (one process per GPU): autograd multithreading off inside, restored outside, same gradients."""
    from chemprop_amd.distributed import backward_on_calling_thread

    assert torch.autograd.is_multithreading_enabled()
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    x = torch.randn(7, 5)
    lin(x).square().sum().backward()
    want = [p.grad.clone() for p in lin.parameters()]
    lin.zero_grad()
    with backward_on_calling_thread():
        assert not torch.autograd.is_multithreading_enabled()
        lin(x).square().sum().backward()
    assert torch.autograd.is_multithreading_enabled()
    for p, w in zip(lin.parameters(), want):
        assert torch.equal(p.grad, w)


def test_hash_partition_padding_keeps_every_molecule():
    n, world = 37, 4
    padded = [ddp.hash_partition(n, r, world, pad=True) for r in range(world)]
    assert len({len(s) for s in padded}) == 1
    assert set(np.concatenate(padded).tolist()) == set(range(n))          # nothing dropped (prediction / evaluation)
    trimmed = [ddp.hash_partition(n, r, world) for r in range(world)]
    assert len(np.concatenate(trimmed)) <= n


@pytest.mark.gpu
def test_backward_writes_into_the_flat_gradient_buffer(gpu_device):
    """GradSync on one GPU: the block's backward kernels write straight into the views of the flat buffer (p.grad IS the
    view, autograd gets None for them), and the values equal the ordinary path's."""
    from conftest import parity_err

    bmg = synth.random_batch(64, "qm9", seed=3)
    torch.manual_seed(1)
    a = BondMessagePassing(d_h=128, bias=True).to(gpu_device).train()
    b = BondMessagePassing(d_h=128, bias=True).to(gpu_device).train()
    b.load_state_dict(a.state_dict())
    bmg.to(gpu_device)
    G = torch.randn(bmg.V.shape[0], 128, device=gpu_device)
    a(bmg).backward(G)
    sync = ddp.GradSync(list(b.parameters()), modules=[b])
    for rep in range(2):
        sync.flat.fill_(float("nan"))       # the kernels must overwrite every entry of their views
        b(bmg).backward(G)
        sync.allreduce(); sync.wait()
        for (k, p), (_, q), v in zip(b.named_parameters(), a.named_parameters(), sync.views):
            assert p.grad.data_ptr() == v.data_ptr(), k
            assert parity_err(p.grad.cpu().numpy(), q.grad.cpu().numpy()) == 0.0, (k, rep)


# ------------------------------------------------------------------------------------------------
# round 4: the STAGED training step of a data-parallel job (model.FusedTrainer at world > 1) executed — two ranks on ONE GPU,
# gradients exchanged through gloo: STEP_FORWARD -> all-reduce(head slice) -> STEP_BACKWARD -> all-reduce(block slice) -> update
# ------------------------------------------------------------------------------------------------
def _make_model(bn, dev):
    from chemprop_amd import agg as cagg
    from chemprop_amd.model import MPNN, RegressionFFN

    torch.manual_seed(11)
    return MPNN(BondMessagePassing(d_h=64), cagg.MeanAggregation(), RegressionFFN(n_tasks=2, input_dim=64, hidden_dim=32),
                batch_norm=bn).to(dev).train()


def _worker_staged(rank, world, port, n_mols, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chemprop_amd.model import FusedTrainer

    dev = torch.device("cuda", 0)   # (every rank on the same device: the collectives go through the host, the kernels are the real ones)
    torch.cuda.set_device(dev)
    mine = dist.new_group([0])      # (every rank creates both groups, in the same order)
    other = dist.new_group([1])
    own = mine if rank == 0 else other
    res = {}
    n_steps = 3
    for bn in (False, True):
        batches = [[synth.random_batch(n_mols, "qm9", seed=40 + 10 * s + r) for r in range(world)] for s in range(n_steps)]
        ys = [[torch.randn(n_mols, 2, generator=torch.Generator().manual_seed(7 + 10 * s + r)) for r in range(world)] for s in range(n_steps)]
        staged, single = _make_model(bn, dev), _make_model(bn, dev)
        tr_staged = FusedTrainer(staged, lr=1e-3)                 # the default group: world 2 -> the staged step
        tr_single = FusedTrainer(single, lr=1e-3, group=own)      # a group of this rank alone: the ONE-call step
        assert tr_staged._world() == 2 and tr_single._world() == 1
        losses, gsum, gown = [], [], []
        for s in range(n_steps):
            b = batches[s][rank]
            b.to(dev)
            y = ys[s][rank].to(dev)
            # the one-call step of THIS rank's batch on a twin with the SAME parameters: its gradients, summed over the ranks by an
            # ordinary all-reduce, are what the staged step's two sliced exchanges must have left in its flat buffer
            single.load_state_dict(staged.state_dict())
            l1 = tr_single.step(b, y)
            g1 = tr_single.sync.flat.detach().clone()
            dist.all_reduce(g1)
            l2 = tr_staged.step(b, y)
            tr_staged.sync.wait()
            torch.cuda.synchronize()
            losses.append((float(l1[0]), float(l2[0])))
            gsum.append((g1.cpu(), tr_staged.sync.flat.detach().cpu().clone()))
        res[bn] = dict(losses=losses, gsum=gsum, params=tr_staged.opt.flat.detach().cpu().clone(),
                       bn=None if not bn else (staged.bn.running_mean.cpu().clone(), staged.bn.running_var.cpu().clone(), int(staged.bn.num_batches_tracked),
                                               single.bn.running_mean.cpu().clone(), single.bn.running_var.cpu().clone()))
        if not bn:
            # the UNION batch in one process, one call per step: molecules are independent and the loss is a mean over equally many
            # targets per rank, so its gradient is the mean of the ranks' gradients -> the same parameters after the same steps
            from chemprop_amd.data import BatchMolGraph

            union = _make_model(False, dev)
            tr_u = FusedTrainer(union, lr=1e-3, group=own)
            for s in range(n_steps):
                mgs = sum((synth.random_molgraphs(n_mols, "qm9", seed=40 + 10 * s + r) for r in range(world)), [])
                ub = BatchMolGraph(mgs)
                ub.to(dev)
                tr_u.step(ub, torch.cat(ys[s]).to(dev))
            torch.cuda.synchronize()
            res["union_params"] = tr_u.opt.flat.detach().cpu().clone()
    torch.save(res, os.path.join(out_dir, f"staged{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_staged_data_parallel_step_on_one_gpu(tmp_path, gpu_device):
    """``FusedTrainer.step`` at world 2 (round-3 VERDICT: "designed but unexercised"): two processes on cuda:0, gloo.
      * every staged step's loss = the one-call step's on the same batch and parameters;
      * its flat gradient buffer after the two sliced exchanges = the all-reduced sum of the ranks' one-call gradients;
      * both ranks hold the same parameters afterwards, equal to a one-process run on the UNION batch;
      * batch-norm buffers are per-rank statistics (DDP semantics: no SyncBatchNorm in the reference), counted once per step."""
    from conftest import parity_err

    world, n_mols = 2, 48
    mp.spawn(_worker_staged, args=(world, _free_port(), n_mols, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"staged{k}.pt") for k in range(world)]
    for bn in (False, True):
        for k in range(world):
            for s, (l1, l2) in enumerate(r[k][bn]["losses"]):
                assert abs(l1 - l2) <= 2e-6 * max(1.0, abs(l1)), (bn, k, s, l1, l2)
            for s, (want, got) in enumerate(r[k][bn]["gsum"]):
                assert parity_err(got.numpy(), want.numpy()) <= 5e-6, (bn, k, s)
        assert torch.equal(r[0][bn]["params"], r[1][bn]["params"])          # the ranks stay in lock step, bit for bit
    assert parity_err(r[0][False]["params"].numpy(), r[0]["union_params"].numpy()) <= 2e-5
    for k in range(world):
        m_s, v_s, n, m_1, v_1 = r[k][True]["bn"]
        assert n == 3
        assert parity_err(m_s.numpy(), m_1.numpy()) <= 1e-6 and parity_err(v_s.numpy(), v_1.numpy()) <= 1e-6
    assert not torch.equal(r[0][True]["bn"][0], r[1][True]["bn"][0])         # (per-rank statistics: the batches differ)
