"""f4: the optimizer step of the training loop as ONE launch.

chemprop trains with ``torch.optim.Adam`` (``models/model.py:208-231``; a Noam schedule sets the learning rate per step).
``FlatAdam`` does the same arithmetic over flat buffers: the parameters become views of ONE buffer, their gradients are the
views of :class:`chemprop_amd.distributed.GradSync`'s buffer (which the backward kernels of the block write into and ONE
RCCL all-reduce sums), and a step is one HBM-bound elementwise kernel (``csrc/dmpnn_optim.hip``) instead of the ~8 launches
of the foreach implementation — at 0.3 M parameters the step is launch-bound, so the launch count is its cost.

Every tensor-API update bumps ``Tensor._version``, which is what the engine's caches of pre-split weights key on; a kernel
writing through raw pointers does not, so ``step()`` bumps the versions itself (``torch.autograd.graph.increment_version``).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib, engine
from .distributed import GradSync

__all__ = ["FlatAdam"]


class FlatAdam:
    """``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (amsgrad off) over the parameters of a :class:`GradSync`.

    ``step(lr=None)`` waits (stream dependency) for the pending gradient exchange, divides by the world size when the
    exchange summed (``GradSync(average=False)``), updates, and bumps the parameters' versions.  ``lr`` may change every
    step (Noam schedule: pass the scheduler's value)."""

    def __init__(self, sync: GradSync, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        self.sync = sync
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.steps = 0
        dev = sync.flat.device
        engine._require_device(sync.flat, "GradSync buffer")
        if any(p.dtype != torch.float32 for p in sync.params):
            raise TypeError("FlatAdam: fp32 parameters only")
        # the parameters as views of one buffer, laid out exactly like the gradient buffer
        self.flat = torch.zeros_like(sync.flat)
        with torch.no_grad():
            for p, o in zip(sync.params, sync.offsets):
                view = self.flat[o:o + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
        self.m = torch.zeros_like(sync.flat)
        self.v = torch.zeros_like(sync.flat)
        self._dev = dev

    def step(self, lr: Optional[float] = None) -> None:
        s = self.sync
        s.wait()
        s._gather()  # (a gradient autograd assigned as a fresh tensor is folded into the buffer first)
        self.steps += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.steps
        bc2 = 1.0 - b2 ** self.steps
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size(s.group)
        scale = 1.0 if (s.average or world == 1) else 1.0 / world
        with engine._OnDevice(self._dev):
            _lib.check(_lib.load().dmpnn_adam_step(
                self.flat.data_ptr(), s.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.flat.numel(),
                C.c_float(self.lr if lr is None else float(lr)), C.c_float(b1), C.c_float(b2), C.c_float(self.eps),
                C.c_float(self.weight_decay), C.c_float(bc1), C.c_float(math.sqrt(bc2)), C.c_float(scale), None,
                engine._stream_ptr(self._dev)), "dmpnn_adam_step")
        for p in s.params:  # (the engine's weight caches key on the autograd version)
            torch.autograd.graph.increment_version(p)
        s.new_step()  # (the gradients are consumed: the next backward may overwrite the views again)

    def zero_grad(self) -> None:
        self.sync.zero_grad()

    # ---- checkpointing (the reference's Lightning checkpoints carry the optimizer state: moments and step count) ----
    def state_dict(self) -> dict:
        """The flat moments, the step count and the hyper-parameters; ``layout`` (names are positions: shapes in flat-buffer order)
        lets ``load_state_dict`` refuse a model of another shape."""
        return {"step": self.steps, "exp_avg": self.m.detach().clone(), "exp_avg_sq": self.v.detach().clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                "layout": [tuple(p.shape) for p in self.sync.params]}

    def load_state_dict(self, sd: dict) -> None:
        layout = [tuple(p.shape) for p in self.sync.params]
        if [tuple(x) for x in sd["layout"]] != layout or sd["exp_avg"].numel() != self.m.numel():
            raise ValueError("FlatAdam.load_state_dict: the state belongs to parameters of another shape / order")
        self.steps = int(sd["step"])
        self.m.copy_(sd["exp_avg"].to(self.m.device))
        self.v.copy_(sd["exp_avg_sq"].to(self.v.device))
        self.lr, self.betas, self.eps, self.weight_decay = float(sd["lr"]), (float(sd["betas"][0]), float(sd["betas"][1])), float(sd["eps"]), float(sd["weight_decay"])

    def torch_state(self) -> dict:
        """The same state in ``torch.optim.Adam.state_dict()['state']`` form (per parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``,
        keyed by position): what a checkpoint written by the reference's trainer holds, for moving a run between the two."""
        out = {}
        for i, (p, o) in enumerate(zip(self.sync.params, self.sync.offsets)):
            n = p.numel()
            out[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": self.m[o:o + n].view_as(p).clone(),
                      "exp_avg_sq": self.v[o:o + n].view_as(p).clone()}
        return out

    def load_torch_state(self, state: dict) -> None:
        for i, (p, o) in enumerate(zip(self.sync.params, self.sync.offsets)):
            st = state[i]
            n = p.numel()
            self.m[o:o + n].view_as(p).copy_(st["exp_avg"].to(self.m.device))
            self.v[o:o + n].view_as(p).copy_(st["exp_avg_sq"].to(self.v.device))
            self.steps = int(st["step"])
