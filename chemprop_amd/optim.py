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

__all__ = ["FlatAdam", "CLIP_MODES"]

CLIP_MODES = {"norm": 0, "value": 1}   # enum dmpnn_clip_mode


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
        self.clip_ws = torch.zeros(int(_lib.load().dmpnn_clip_grad_ws_bytes()) // 4, dtype=torch.float32, device=dev)

    def _grad_scale(self) -> float:
        s = self.sync
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size(s.group)
        return 1.0 if (s.average or world == 1) else 1.0 / world

    def clip_grad(self, clip_val: float, algorithm: str = "norm") -> None:
        """``torch.nn.utils.clip_grad_norm_(params, clip_val)`` / ``clip_grad_value_`` over the flat gradient buffer (what Lightning's
        ``Trainer(gradient_clip_val=..., gradient_clip_algorithm=...)`` runs between the backward pass and ``optimizer.step``,
        ``cli/train.py:1937``): after the pending exchange, on the AVERAGED gradient, no host read.  The total norm stays on the
        device in ``self.clip_ws[256]``."""
        if clip_val is None or not (float(clip_val) > 0):
            return
        if algorithm not in CLIP_MODES:
            raise ValueError(f"gradient_clip_algorithm must be one of {sorted(CLIP_MODES)}, got {algorithm!r}")
        s = self.sync
        s.wait()
        s._gather()
        with engine._OnDevice(self._dev):
            _lib.check(_lib.load().dmpnn_clip_grad(s.flat.data_ptr(), s.flat.numel(), C.c_float(float(clip_val)), CLIP_MODES[algorithm],
                                                   C.c_float(self._grad_scale()), self.clip_ws.data_ptr(), engine._stream_ptr(self._dev)),
                       "dmpnn_clip_grad")

    def step(self, lr: Optional[float] = None, clip: Optional[tuple] = None) -> None:
        s = self.sync
        s.wait()
        s._gather()  # (a gradient autograd assigned as a fresh tensor is folded into the buffer first)
        if clip is not None:
            self.clip_grad(*clip)
        self.steps += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.steps
        bc2 = 1.0 - b2 ** self.steps
        scale = self._grad_scale()
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
