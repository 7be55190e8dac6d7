"""Multi-GPU: pure data parallelism over molecules.

No edge crosses a molecule (``chemprop/data/collate.py:48-56``), so the forward of the path needs
NO collective: each rank (one process per GPU) owns a shard of molecules and batches it on its own.
The only exchange of a training step is the gradient all-reduce, which the reference gets from
Lightning's DDP (``chemprop/cli/train.py:1930-1943``).  Here it is one flat RCCL all-reduce
(``torch.distributed`` backend ``"nccl"`` is RCCL on ROCm; ~1.27 MB at d_h = 300: latency-bound on
xGMI, so ONE bucket, not DDP's 25 MB multi-bucket default tuned for NVSwitch).
"""
from __future__ import annotations

from typing import Iterable, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def hash_partition(n_items: int, rank: int, world: int, seed: int = 0, equalize: bool = True,
                   weights: Optional[Sequence[float]] = None, pad: bool = False, return_own: bool = False):
    """Indices of the molecules owned by ``rank``: ``hash(molecule id) mod world`` (BASELINE.json
    north_star), independent of any sampler state, identical on every rank without communication.

    ``equalize`` trims every shard to the smallest one so that all ranks run the same number of
    steps (a rank with an extra batch would dead-lock the gradient all-reduce) — for TRAINING; it drops molecules, so
    prediction / evaluation sharding passes ``pad=True`` (shards are padded with repeats instead; ``return_own=True`` then returns
    ``(indices, n_own)`` so that the caller can drop the padded tail) or ``equalize=False``.  With ``weights``
    (e.g. directed-edge counts) shards are instead built greedily in hash order so that the summed
    weight — the actual work — is balanced (ZINC-like size spread).
    """
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    ids = np.arange(n_items, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(ids + np.uint64(seed) * np.uint64(0x632BE59BD9B4E019))
    if weights is None:
        owner = (h % np.uint64(world)).astype(np.int64)
        shards = [np.flatnonzero(owner == r) for r in range(world)]
    else:
        w = np.asarray(weights, dtype=np.float64)
        order = np.argsort(h, kind="stable")
        load = np.zeros(world)
        buckets = [[] for _ in range(world)]
        for i in order:  # hash order, lightest rank first: deterministic and balanced
            r = int(np.argmin(load))
            buckets[r].append(int(i))
            load[r] += w[i]
        shards = [np.sort(np.asarray(b, dtype=np.int64)) for b in buckets]
    if equalize and pad:
        # prediction / evaluation: nothing may be dropped — short shards repeat their first molecules up to the longest
        # (the caller discards the padded tail: the returned indices beyond the shard's own length are duplicates)
        # ``return_own=True`` also returns how many leading indices are the shard's OWN (everything behind them is padding); an
        # empty shard is padded from the first non-empty one with n_own = 0 — every rank then runs the same number of steps
        m = max(len(s) for s in shards)
        own = [len(s) for s in shards]
        donor = next((s for s in shards if len(s)), np.zeros(0, dtype=np.int64))
        shards = [np.concatenate([s, np.resize(s if len(s) else donor, m - len(s))]) if (len(s) < m and len(donor)) else s for s in shards]
        return (shards[rank], own[rank]) if return_own else shards[rank]
    elif equalize:
        m = min(len(s) for s in shards)
        shards = [s[:m] for s in shards]
    return (shards[rank], len(shards[rank])) if return_own else shards[rank]


def backward_on_calling_thread():
    """Context manager for the training loop of a one-process-per-GPU job: ``loss.backward()`` runs on the thread that calls
    it (``torch.autograd.set_multithreading_enabled(False)``).

    By default the autograd engine hands the backward pass of a graph that lives on a device to that device's worker thread
    and waits for it.  With one device per process the worker buys nothing, and the hand-off — two thread wake-ups and a GIL
    transfer for a Python ``autograd.Function`` such as the block's — costs ~95 µs per backward on the MI355X hosts: more
    than any kernel of a 512-molecule training step, which was HOST-bound because of it (237–345 µs per step over the hosts
    of the pool; 210 µs, the device time, inside this context: ``scripts/probe_train_host2.py``).  Nothing about the
    arithmetic or the order of the kernels changes."""
    return torch.autograd.set_multithreading_enabled(False)


def allreduce_grads(params: Iterable[torch.nn.Parameter], group=None, average: bool = False) -> None:
    """Sum (or mean) the gradients of ``params`` over all ranks with ONE flat all-reduce (the simple, sequential form:
    :class:`GradSync` is the one a training loop keeps).  EVERY parameter that requires a gradient takes part on every rank
    — one whose gradient is ``None`` here (an unused W_d, a rank whose shard was empty) contributes zeros — so the flat sizes
    agree across ranks whatever each rank's batch touched."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat.div_(world)
    o = 0
    for p in ps:
        n = p.numel()
        g = flat[o:o + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        o += n


def force_collective() -> bool:
    """``DMPNN_FORCE_COLLECTIVE=1`` (tests, one-GPU boxes): run the gradient exchange — the all-reduce on the communication stream, its
    event, the stream-level wait — also at world size 1, where it is arithmetically the identity, so that the RCCL half of
    :class:`GradSync` and the staged step of ``model.FusedTrainer`` execute on the hardware that is there."""
    import os

    return os.environ.get("DMPNN_FORCE_COLLECTIVE", "0") == "1"


class GradSync:
    """The gradient exchange of a data-parallel training step, SURVEY 8e: ONE pre-allocated flat fp32 buffer holds every
    gradient, ``p.grad`` of every parameter is a view into it, and the message-passing block's backward kernels
    (``dmpnn_backward``) write their results straight into those views — no ``cat``, no copies back.  (Those kernels
    OVERWRITE their views: one backward per exchange; gradients of other modules accumulate into theirs as usual, so call
    ``zero_grad()`` of this class once per step.)  ``allreduce()``
    launches ONE all-reduce of the whole buffer on a communication stream behind the work already queued on the compute
    stream and returns at once; ``wait()`` (call it before the optimizer reads the gradients, or before the next backward
    overwrites them) makes the compute stream wait for it — a stream dependency, not a host sync.  RCCL over xGMI is
    latency-bound at 1.3 MB, so one bucket (DDP's 25 MB multi-bucket default is tuned for other fabrics and sizes).

    Works with any process group (``gloo`` on the CPU in the tests).  Parameters whose gradient was produced elsewhere
    (autograd of other modules assigns fresh tensors) are folded into the buffer at ``allreduce()``."""

    def __init__(self, params: Iterable[torch.nn.Parameter], modules: Iterable[torch.nn.Module] = (), group=None, average: bool = False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradSync: no parameter requires a gradient")
        self.group, self.average = group, average
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4  # (16-byte aligned views: the kernels store float4)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.n_collectives = 0  # all-reduces launched so far (diagnostics / tests)
        self.works: list = []   # pending exchanges of this step: (work, slice of the flat buffer, event on the communication stream)
        for p, v in zip(self.params, self.views):
            p.grad = v
        # views a block's backward kernels have OVERWRITTEN since the last zero_grad() / allreduce() / optimizer step (shared
        # with every block below): a second backward through the same view in one step must accumulate instead
        self._written: set = set()
        # the engine's blocks write their gradients into the views directly (backward.FusedMP reads this attribute)
        by_id = {id(p): v for p, v in zip(self.params, self.views)}
        for m in modules:
            for blk in m.modules():
                names = {}
                for lin, short in (("W_i", "i"), ("W_h", "h"), ("W_o", "o"), ("W_d", "d")):
                    layer = getattr(blk, lin, None)
                    if isinstance(layer, torch.nn.Linear):
                        if id(layer.weight) in by_id:
                            names["W_" + short] = by_id[id(layer.weight)]
                        if layer.bias is not None and id(layer.bias) in by_id:
                            names["b_" + short] = by_id[id(layer.bias)]
                if names:
                    blk.__dict__["_dmpnn_grad_views"] = names
                    blk.__dict__["_dmpnn_grad_written"] = self._written

    def zero_grad(self) -> None:
        """One fill of the flat buffer; every ``p.grad`` stays (or becomes again) its view.  Use this instead of
        ``optimizer.zero_grad()`` (whose ``set_to_none`` would detach the parameters from the buffer: still correct —
        ``allreduce`` folds stray gradients back in — but it costs the copies this class exists to avoid)."""
        self.wait()
        self.flat.zero_()
        self._written.clear()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def new_step(self) -> None:
        """The gradients of this step have been consumed (optimizer step, or an exchange has been launched): the next backward
        through a block may overwrite its views again.  ``FlatAdam.step`` and ``allreduce`` call it."""
        self._written.clear()

    def _gather(self, lo: int = 0, hi: Optional[int] = None) -> None:
        hi = self.flat.numel() if hi is None else hi
        for p, v, o in zip(self.params, self.views, self.offsets):
            if o < lo or o >= hi:
                continue
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                v.copy_(g)
            p.grad = v

    def range_of(self, params: Iterable[torch.nn.Parameter]) -> tuple[int, int]:
        """``(lo, hi)`` of the flat buffer covering ``params`` — which must be a contiguous run of this exchange's parameters (a
        sub-module's, in ``model.parameters()`` order) — for a partial ``allreduce(lo, hi)``."""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)) or len(idx) != len(ids):
            raise ValueError("range_of: the parameters must be a contiguous run of the exchange's parameters")
        hi = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.offsets) else self.flat.numel()
        return self.offsets[idx[0]], hi

    def allreduce(self, lo: Optional[int] = None, hi: Optional[int] = None) -> None:
        """Launch the exchange of this step's gradients (returns at once).  With ``(lo, hi)``: of that slice of the flat buffer
        only — a step may exchange disjoint slices one after the other, each as soon as its gradients are final (the
        predictor's and batch norm's, complete before the block's backward pass starts, go out while that pass runs:
        ``model.FusedTrainer``); ``wait()`` waits for all of them."""
        whole = lo is None and hi is None
        lo = 0 if lo is None else int(lo)
        hi = self.flat.numel() if hi is None else int(hi)
        if whole:
            self.wait()
        self._gather(lo, hi)
        if whole or hi == self.flat.numel():
            self._written.clear()  # (the step's gradient is complete: the next backward starts a new one)
        if not (dist.is_available() and dist.is_initialized()) or hi <= lo:
            return
        if dist.get_world_size(self.group) == 1 and not force_collective():
            return
        self.n_collectives += 1
        buf = self.flat[lo:hi]
        if self.stream is None:
            self.works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), buf, None))
            return
        self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self.stream):
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            work.wait()  # (stream-level on the communication stream: what follows there is ordered behind the collective)
            if self.average:
                buf.div_(dist.get_world_size(self.group))
            done = torch.cuda.Event()
            done.record(self.stream)
        self.works.append((work, buf, done))

    def wait(self) -> None:
        """Order everything queued on the compute stream from here on behind the exchange(s) (no host sync on a GPU)."""
        if not self.works:
            return
        for work, buf, done in self.works:
            if done is None:
                work.wait()
                if self.average:
                    buf.div_(dist.get_world_size(self.group))
            else:
                torch.cuda.current_stream(self.flat.device).wait_event(done)
        self.works = []


def broadcast_params(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica start from rank ``src``'s weights (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    ts = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if not ts:
        return
    # floating-point tensors travel as one flat fp32 buffer; anything else (integer buffers: step counters, ...) in its own dtype
    fl = [t for t in ts if t.is_floating_point()]
    if fl:
        flat = torch.cat([t.reshape(-1).float() for t in fl])
        dist.broadcast(flat, src=src, group=group)
        o = 0
        for t in fl:
            n = t.numel()
            t.copy_(flat[o:o + n].view_as(t).to(t.dtype))
            o += n
    for t in ts:
        if not t.is_floating_point():
            dist.broadcast(t, src=src, group=group)
