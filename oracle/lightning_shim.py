"""A stand-in for the part of ``lightning.pytorch`` (pinned ``lightning >= 2.0``, reference ``pyproject.toml:32``; written against
2.5) that ``chemprop train`` drives around ``MPNN`` (``cli/train.py:1890-1999``, ``models/model.py:148-231``).

TEST INFRASTRUCTURE ONLY (like ``oracle/ref_shim.py``, which installs it as ``lightning.pytorch``): Lightning is a third-party
dependency that is neither installed here nor vendored under ``/root/reference``, so what is restated is its PUBLISHED control
flow — the hooks a ``LightningModule`` sees, in the order and with the bookkeeping that decide whether a drop-in survives
``Trainer.fit``.  Nothing in ``chemprop_amd/`` imports this module.  What is restated, each with the Lightning source it follows:

* ``Trainer.fit`` → ``_verify_loop_configurations`` (``trainer/configuration_validator.py``: manual optimization refuses
  ``gradient_clip_val`` and ``accumulate_grad_batches``), ``strategy.setup`` (module to device, optional DDP wrap, then
  ``configure_optimizers``), the fit loop (``loops/fit_loop.py``, ``loops/training_epoch_loop.py``): per batch the AUTOMATIC
  optimization closure (``loops/optimization/automatic.py``: ``training_step`` → ``optimizer_zero_grad`` → ``backward``, run INSIDE
  ``optimizer.step(closure)``; ``on_before_optimizer_step`` and ``configure_gradient_clipping`` between the closure and the update:
  ``plugins/precision/precision.py``) or, with ``automatic_optimization = False``, the bare ``training_step``
  (``loops/optimization/manual.py``); learning-rate schedulers with ``interval="step"`` stepped by the loop in automatic mode only;
  validation at the end of every epoch under ``on_validation_model_eval`` / ``no_grad``.
* ``trainer.global_step`` = completed OPTIMIZER steps (``optim_progress.optimizer.step.total.completed``; manual mode: steps taken
  through ``LightningOptimizer.step``) — what ``ModelCheckpoint`` keys on.
* ``LightningOptimizer`` (``core/optimizer.py``): ``step(closure)`` = progress ``ready`` → ``strategy.optimizer_step`` → progress
  ``completed``.
* ``LightningModule.log`` → the result collection (``trainer/connectors/logger_connector/result.py``): the same key logged twice in
  one hook with different metadata raises; ``on_step`` / ``on_epoch`` forks (``train_loss_step`` / ``train_loss_epoch``), epoch
  values as batch-size weighted means, ``Metric`` objects through ``compute()``.
* ``ModelCheckpoint`` (``callbacks/model_checkpoint.py``): skips a save when ``_last_global_step_saved == trainer.global_step``;
  top-1 on ``monitor``; ``best_model_path``; ``save_last``; the checkpoint dictionary of ``trainer.save_checkpoint``
  (``checkpoint_connector.dump_checkpoint``: ``epoch``, ``global_step``, ``state_dict``, ``optimizer_states``, ``lr_schedulers``,
  ``hyper_parameters`` + the module's ``on_save_checkpoint``).  ``EarlyStopping`` (``callbacks/early_stopping.py``): patience on
  the monitored value.
* ``LightningModule.load_from_checkpoint`` (``core/saving.py: _load_state``): ``cls(**hyper_parameters)`` → ``on_load_checkpoint``
  → ``load_state_dict``.
"""
from __future__ import annotations

import inspect
import os
from typing import Any, Optional

import torch
from torch import nn


class MisconfigurationException(Exception):
    pass


class _AttributeDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class HyperparametersMixin:
    """``core/mixins/hparams_mixin.py``: collect the caller's ``__init__`` arguments."""

    def save_hyperparameters(self, *args, ignore=None, frame=None, logger=True):
        ignore = set(ignore or ())
        frame = frame or inspect.currentframe().f_back
        local = frame.f_locals
        code = frame.f_code
        names = [p for p in code.co_varnames[:code.co_argcount + code.co_kwonlyargcount] if p != "self"]
        hp = _AttributeDict()
        for n in names:
            if n in local and n not in ignore:
                hp[n] = local[n]
        self._hparams = hp

    @property
    def hparams(self):
        if not hasattr(self, "_hparams"):
            self._hparams = _AttributeDict()
        return self._hparams


# ------------------------------------------------------------------------------------------------------------------
# progress trackers (loops/progress.py)
# ------------------------------------------------------------------------------------------------------------------
class _Progress:
    def __init__(self):
        self.ready = 0
        self.completed = 0


# ------------------------------------------------------------------------------------------------------------------
# the result collection of self.log
# ------------------------------------------------------------------------------------------------------------------
class _ResultMetric:
    def __init__(self, meta):
        self.meta = meta
        self.value = None          # last step value
        self.cumulated = 0.0       # sum of value * batch_size
        self.weight = 0.0
        self.metric = None         # a Metric object logged as the value


class _ResultCollection:
    def __init__(self):
        self.items: dict[str, _ResultMetric] = {}

    def log(self, fx: str, name: str, value, meta: dict, batch_size: int):
        key = f"{fx}.{name}"
        if key not in self.items:
            self.items[key] = _ResultMetric(meta)
        elif self.items[key].meta != meta:
            # result.py: _ResultCollection.log
            raise MisconfigurationException(
                f"You called `self.log({name}, ...)` twice in `{fx}` with different arguments. This is not allowed")
        rm = self.items[key]
        if hasattr(value, "compute") and isinstance(value, nn.Module):
            rm.metric = value
            rm.value = getattr(value, "_forward_cache", None)
        else:
            v = value.detach() if isinstance(value, torch.Tensor) else torch.as_tensor(float(value))
            rm.value = v
            rm.cumulated = rm.cumulated + v.float() * batch_size
            rm.weight += batch_size

    def metrics(self, on_step: bool) -> dict:
        out = {}
        for key, rm in self.items.items():
            name = key.split(".", 1)[1]
            m = rm.meta
            if on_step and m["on_step"]:
                v = rm.value
                if v is not None:
                    out[name + ("_step" if m["on_epoch"] else "")] = v
                    out[name] = v
            if not on_step and m["on_epoch"]:
                v = rm.metric.compute() if rm.metric is not None else (rm.cumulated / max(rm.weight, 1e-30))
                out[name + ("_epoch" if m["on_step"] else "")] = v
                out[name] = v
        return out

    def reset(self):
        for rm in self.items.values():
            if rm.metric is not None and hasattr(rm.metric, "reset"):
                rm.metric.reset()
        self.items = {}


# ------------------------------------------------------------------------------------------------------------------
# LightningOptimizer (core/optimizer.py)
# ------------------------------------------------------------------------------------------------------------------
class LightningOptimizer:
    def __init__(self, optimizer, trainer):
        self._optimizer = optimizer
        self._trainer = trainer

    @property
    def optimizer(self):
        return self._optimizer

    def __getattr__(self, name):   # param_groups, state, defaults, zero_grad, ...
        return getattr(self._optimizer, name)

    def step(self, closure=None, **kwargs):
        tr = self._trainer
        tr._on_before_optimizer_step_progress()
        if closure is None:
            closure = lambda: None
        elif not callable(closure):
            raise MisconfigurationException("When `optimizer.step(closure)` is called, the closure should be callable")
        out = tr.strategy.optimizer_step(self._optimizer, closure, **kwargs)
        tr._on_after_optimizer_step_progress()
        return out


# ------------------------------------------------------------------------------------------------------------------
# LightningModule (core/module.py, core/hooks.py)
# ------------------------------------------------------------------------------------------------------------------
class LightningModule(nn.Module, HyperparametersMixin):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__["_trainer"] = None
        self.__dict__["_current_fx_name"] = None
        self.__dict__["_automatic_optimization"] = True

    # ---- attributes ----
    @property
    def automatic_optimization(self) -> bool:
        return self.__dict__.get("_automatic_optimization", True)

    @automatic_optimization.setter
    def automatic_optimization(self, v: bool) -> None:
        self.__dict__["_automatic_optimization"] = bool(v)

    @property
    def trainer(self):
        tr = self.__dict__.get("_trainer")
        if tr is None:
            raise RuntimeError(f"{type(self).__name__} is not attached to a `Trainer`.")   # core/module.py: trainer
        return tr

    @trainer.setter
    def trainer(self, tr):
        self.__dict__["_trainer"] = tr

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    @property
    def current_epoch(self) -> int:
        tr = self.__dict__.get("_trainer")
        return tr.current_epoch if tr is not None else 0

    @property
    def global_step(self) -> int:
        tr = self.__dict__.get("_trainer")
        return tr.global_step if tr is not None else 0

    def optimizers(self, use_pl_optimizer: bool = True):
        tr = self.trainer
        opts = [LightningOptimizer(o, tr) for o in tr.optimizers] if use_pl_optimizer else list(tr.optimizers)
        return opts[0] if len(opts) == 1 else opts

    def lr_schedulers(self):
        cfgs = self.trainer.lr_scheduler_configs
        if not cfgs:
            return None
        s = [c["scheduler"] for c in cfgs]
        return s[0] if len(s) == 1 else s

    # ---- logging ----
    def log(self, name, value, prog_bar=False, logger=None, on_step=None, on_epoch=None, reduce_fx="mean", enable_graph=False,
            sync_dist=False, sync_dist_group=None, add_dataloader_idx=True, batch_size=None, metric_attribute=None, rank_zero_only=False):
        tr = self.__dict__.get("_trainer")
        if tr is None:      # core/module.py: log — "not attached": a warning, nothing logged
            return
        fx = self.__dict__.get("_current_fx_name")
        if fx is None:
            raise MisconfigurationException("You are trying to `self.log()` but the loop's result collection is not registered yet")
        training = fx in ("training_step", "on_train_batch_start", "on_train_batch_end")
        on_step = training if on_step is None else on_step
        on_epoch = (not training) if on_epoch is None else on_epoch
        if hasattr(value, "compute") and isinstance(value, nn.Module) and metric_attribute is None:
            # core/module.py: log — a Metric is looked up among the module's attributes
            for n, m in self.named_modules():
                if m is value:
                    metric_attribute = n
                    break
            else:
                raise MisconfigurationException(f"Could not find the `LightningModule` attribute for the `torchmetrics.Metric` logged: {name}")
        meta = dict(prog_bar=prog_bar, logger=True if logger is None else logger, on_step=on_step, on_epoch=on_epoch, reduce_fx=reduce_fx,
                    enable_graph=enable_graph, sync_dist=sync_dist, metric_attribute=metric_attribute)
        tr._results.log(fx, name, value, meta, 1 if batch_size is None else int(batch_size))

    # ---- hooks with Lightning's default bodies ----
    def on_fit_start(self): pass
    def on_fit_end(self): pass
    def on_train_start(self): pass
    def on_train_end(self): pass
    def on_train_epoch_start(self): pass
    def on_train_epoch_end(self): pass
    def on_train_batch_start(self, batch, batch_idx): pass
    def on_train_batch_end(self, outputs, batch, batch_idx): pass
    def on_before_zero_grad(self, optimizer): pass
    def on_before_backward(self, loss): pass
    def on_after_backward(self): pass
    def on_before_optimizer_step(self, optimizer): pass
    def on_validation_epoch_start(self): pass
    def on_validation_epoch_end(self): pass
    def on_save_checkpoint(self, checkpoint): pass
    def on_load_checkpoint(self, checkpoint): pass
    def setup(self, stage): pass
    def teardown(self, stage): pass

    def on_validation_model_eval(self):     # core/hooks.py
        self.eval()

    def on_validation_model_train(self):
        self.train()

    def backward(self, loss, *args, **kwargs):     # core/module.py: backward
        loss.backward(*args, **kwargs)

    def manual_backward(self, loss, *args, **kwargs):
        if self.automatic_optimization:
            raise MisconfigurationException("`manual_backward` is only for manual optimization")
        self.trainer.strategy.backward(loss, None, *args, **kwargs)

    def optimizer_zero_grad(self, epoch, batch_idx, optimizer):     # core/module.py
        optimizer.zero_grad()

    def optimizer_step(self, epoch, batch_idx, optimizer, optimizer_closure=None):     # core/module.py
        optimizer.step(closure=optimizer_closure)

    def lr_scheduler_step(self, scheduler, metric):
        if metric is None:
            scheduler.step()
        else:
            scheduler.step(metric)

    def clip_gradients(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):     # core/module.py
        tr = self.trainer
        if gradient_clip_val is None:
            gradient_clip_val = tr.gradient_clip_val or 0.0
        if gradient_clip_algorithm is None:
            gradient_clip_algorithm = tr.gradient_clip_algorithm or "norm"
        if gradient_clip_val <= 0:
            return
        params = [p for g in optimizer.param_groups for p in g["params"]]
        if gradient_clip_algorithm == "value":
            torch.nn.utils.clip_grad_value_(params, clip_value=gradient_clip_val)
        else:
            torch.nn.utils.clip_grad_norm_(params, gradient_clip_val)

    def configure_gradient_clipping(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
        self.clip_gradients(optimizer, gradient_clip_val=gradient_clip_val, gradient_clip_algorithm=gradient_clip_algorithm)

    # ---- checkpoints (core/saving.py: _load_state) ----
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict=True, **kwargs):
        ck = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
        hp = dict(ck.get("hyper_parameters", {}))
        sig = inspect.signature(cls.__init__)
        if not any(p.kind is p.VAR_KEYWORD for p in sig.parameters.values()):
            hp = {k: v for k, v in hp.items() if k in sig.parameters}
        hp.update(kwargs)
        obj = cls(**hp)
        obj.on_load_checkpoint(ck)
        obj.load_state_dict(ck["state_dict"], strict=strict if strict is not None else True)
        return obj


# ------------------------------------------------------------------------------------------------------------------
# strategies
# ------------------------------------------------------------------------------------------------------------------
class SingleDeviceStrategy:
    """``strategies/single_device.py`` + ``plugins/precision/precision.py`` (32-true)."""

    def __init__(self):
        self.trainer = None
        self.model = None              # what the loops call: the module itself, or its DDP wrapper
        self.lightning_module = None

    def setup(self, trainer, module):
        self.trainer, self.lightning_module = trainer, module
        self.model = self._setup_model(module)

    def _setup_model(self, module):
        return module

    def training_step(self, *args):
        return self.lightning_module.training_step(*args)

    def backward(self, loss, optimizer, *args, **kwargs):
        # strategies/strategy.py: backward -> precision.backward -> LightningModule.backward
        m = self.lightning_module
        m.on_before_backward(loss)
        m.backward(loss, *args, **kwargs)
        m.on_after_backward()

    def optimizer_step(self, optimizer, closure, **kwargs):
        # plugins/precision/precision.py: optimizer_step / _wrap_closure / _after_closure / _clip_gradients
        tr, m = self.trainer, self.lightning_module

        def wrapped():
            out = closure()
            for cb in tr.callbacks:
                if hasattr(cb, "on_before_optimizer_step"):
                    cb.on_before_optimizer_step(tr, m, optimizer)
            m.on_before_optimizer_step(optimizer)
            if m.automatic_optimization:     # (manual optimization: the user clips)
                m.configure_gradient_clipping(optimizer, gradient_clip_val=tr.gradient_clip_val,
                                              gradient_clip_algorithm=tr.gradient_clip_algorithm)
            return out

        return optimizer.step(closure=wrapped, **kwargs)


class DDPStrategy(SingleDeviceStrategy):
    """``strategies/ddp.py``: the module is wrapped in ``torch.nn.parallel.DistributedDataParallel`` (the process group must be
    initialised by the test) and ``training_step`` is called THROUGH the wrapper (``_forward_redirection``), so DDP's pre- / post-forward
    bookkeeping runs around it as it does under Lightning."""

    def _setup_model(self, module):
        from torch.nn.parallel import DistributedDataParallel as DDP

        dev = module.device
        w = _Redirect(module)
        return DDP(w, device_ids=[dev.index] if dev.type == "cuda" else None)

    def training_step(self, *args):
        return self.model(*args)


class _Redirect(nn.Module):
    """``overrides/base.py`` / ``strategies/strategy.py: _ForwardRedirection``: DDP's ``forward`` lands in ``training_step``."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args):
        return self.module.training_step(*args)


# ------------------------------------------------------------------------------------------------------------------
# callbacks
# ------------------------------------------------------------------------------------------------------------------
class Callback:
    pass


class ModelCheckpoint(Callback):
    """``callbacks/model_checkpoint.py`` for ``save_top_k = 1``."""

    def __init__(self, dirpath=None, filename=None, monitor=None, verbose=False, save_last=None, save_top_k=1, save_weights_only=False,
                 mode="min", auto_insert_metric_name=True, every_n_train_steps=None, train_time_interval=None, every_n_epochs=None,
                 save_on_train_epoch_end=None, enable_version_counter=True):
        self.dirpath, self.filename, self.monitor, self.mode = str(dirpath), filename, monitor, mode
        self.save_last, self.auto_insert_metric_name = save_last, auto_insert_metric_name
        self.save_on_train_epoch_end = save_on_train_epoch_end
        self.best_model_path, self.best_model_score, self.last_model_path = "", None, ""
        self._last_global_step_saved = 0
        self.n_saved = 0

    def _should_skip_saving_checkpoint(self, trainer) -> bool:
        return trainer.sanity_checking or trainer.state_fn != "fit" or self._last_global_step_saved == trainer.global_step

    def _should_save_on_train_epoch_end(self, trainer) -> bool:
        if self.save_on_train_epoch_end is not None:
            return self.save_on_train_epoch_end
        return trainer.val_dataloader is None      # (check_val_every_n_epoch == 1: validation runs, save after it)

    def on_train_epoch_end(self, trainer, module):
        if not self._should_skip_saving_checkpoint(trainer) and self._should_save_on_train_epoch_end(trainer):
            self._save(trainer)

    def on_validation_end(self, trainer, module):
        if not self._should_skip_saving_checkpoint(trainer) and not self._should_save_on_train_epoch_end(trainer):
            self._save(trainer)

    def _format(self, trainer, metrics) -> str:
        name = self.filename or "{epoch}-{step}"
        vals = dict(metrics)
        vals.update(epoch=trainer.current_epoch, step=trainer.global_step)
        import re

        def sub(mo):
            key, fmt = mo.group(1), mo.group(2) or ""
            v = vals.get(key, 0)
            v = float(v) if isinstance(v, torch.Tensor) else v
            s = format(v, fmt[1:]) if fmt else str(v)
            return s if not self.auto_insert_metric_name else f"{key}={s}"

        return re.sub(r"\{([^{}:]+)(:[^{}]*)?\}", sub, name)

    def _save(self, trainer):
        metrics = dict(trainer.callback_metrics)
        os.makedirs(self.dirpath, exist_ok=True)
        if self.monitor is not None:
            if self.monitor not in metrics:
                raise MisconfigurationException(
                    f"`ModelCheckpoint(monitor={self.monitor!r})` could not find the monitored key in the returned metrics: {sorted(metrics)}")
            cur = float(metrics[self.monitor])
            better = self.best_model_score is None or (cur < self.best_model_score if self.mode == "min" else cur > self.best_model_score)
            if better:
                path = os.path.join(self.dirpath, self._format(trainer, metrics) + ".ckpt")
                if self.best_model_path and os.path.isfile(self.best_model_path) and self.best_model_path != path:
                    os.remove(self.best_model_path)
                trainer.save_checkpoint(path)
                self.best_model_path, self.best_model_score = path, cur
                self.n_saved += 1
        else:
            path = os.path.join(self.dirpath, self._format(trainer, metrics) + ".ckpt")
            trainer.save_checkpoint(path)
            self.best_model_path = path
            self.n_saved += 1
        self._last_global_step_saved = trainer.global_step
        if self.save_last:
            self.last_model_path = os.path.join(self.dirpath, "last.ckpt")
            trainer.save_checkpoint(self.last_model_path)


class EarlyStopping(Callback):
    def __init__(self, monitor, min_delta=0.0, patience=3, verbose=False, mode="min", **_):
        self.monitor, self.min_delta, self.patience, self.mode = monitor, abs(min_delta or 0.0), patience, mode
        self.best, self.wait = None, 0

    def _check(self, trainer):
        m = trainer.callback_metrics
        if self.monitor not in m:
            return
        cur = float(m[self.monitor])
        if self.best is None or (cur < self.best - self.min_delta if self.mode == "min" else cur > self.best + self.min_delta):
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                trainer.should_stop = True

    def on_validation_end(self, trainer, module):
        self._check(trainer)

    def on_train_epoch_end(self, trainer, module):
        if trainer.val_dataloader is None:
            self._check(trainer)


# ------------------------------------------------------------------------------------------------------------------
# Trainer
# ------------------------------------------------------------------------------------------------------------------
class Trainer:
    def __init__(self, logger=None, enable_progress_bar=False, accelerator="auto", devices="auto", max_epochs=None, callbacks=None,
                 gradient_clip_val=None, gradient_clip_algorithm=None, deterministic=None, accumulate_grad_batches=1, strategy=None,
                 max_steps=-1, enable_checkpointing=True, **_):
        self.max_epochs = 1000 if max_epochs is None else max_epochs
        self.max_steps = max_steps
        self.callbacks = list(callbacks or [])
        self.gradient_clip_val, self.gradient_clip_algorithm = gradient_clip_val, gradient_clip_algorithm
        self.accumulate_grad_batches = accumulate_grad_batches
        self.strategy = strategy if strategy is not None else SingleDeviceStrategy()
        self.optimizers, self.lr_scheduler_configs = [], []
        self.current_epoch = 0
        self._epochs_done = 0               # epochs whose batches have all been run (fit_loop.epoch_progress.current.processed)
        self.should_stop = False
        self.sanity_checking = False
        self.state_fn = None
        self.callback_metrics: dict = {}
        self.logged_metrics: dict = {}
        self._results = _ResultCollection()
        self._auto_step = _Progress()       # optim_progress.optimizer.step (automatic optimization)
        self._manual_step = _Progress()     # manual_optimization.optim_step_progress
        self.lightning_module = None
        self.train_dataloader = self.val_dataloader = None
        self.num_training_batches = float("inf")
        self.hook_trace: list = []          # (tests look at the order of the hooks)

    # ---- progress ----
    @property
    def global_step(self) -> int:
        m = self.lightning_module
        if m is None or m.automatic_optimization:
            return self._auto_step.completed
        return self._manual_step.completed

    def _on_before_optimizer_step_progress(self):
        m = self.lightning_module
        (self._auto_step if m.automatic_optimization else self._manual_step).ready += 1

    def _on_after_optimizer_step_progress(self):
        m = self.lightning_module
        (self._auto_step if m.automatic_optimization else self._manual_step).completed += 1

    @property
    def estimated_stepping_batches(self):
        return self.num_training_batches * self.max_epochs

    @property
    def checkpoint_callback(self):
        return next((c for c in self.callbacks if isinstance(c, ModelCheckpoint)), None)

    @property
    def model(self):
        return self.strategy.model

    # ---- fit ----
    def _verify(self, module):
        # trainer/configuration_validator.py: __verify_manual_optimization_support
        if module.automatic_optimization:
            return
        if self.gradient_clip_val is not None and self.gradient_clip_val > 0:
            raise MisconfigurationException(
                "Automatic gradient clipping is not supported for manual optimization."
                f" Remove `Trainer(gradient_clip_val={self.gradient_clip_val})` or switch to automatic optimization.")
        if self.accumulate_grad_batches != 1:
            raise MisconfigurationException(
                "Automatic gradient accumulation is not supported for manual optimization."
                f" Remove `Trainer(accumulate_grad_batches={self.accumulate_grad_batches})` or switch to automatic optimization.")

    def _call(self, module, name, *args):
        prev = module.__dict__.get("_current_fx_name")
        module.__dict__["_current_fx_name"] = name
        try:
            self.hook_trace.append(name)
            return getattr(module, name)(*args)
        finally:
            module.__dict__["_current_fx_name"] = prev

    def _callbacks(self, name, module, *args):
        for cb in self.callbacks:
            fn = getattr(cb, name, None)
            if fn is not None:
                fn(self, module, *args)

    def _setup_optimizers(self, module):
        cfg = self._call(module, "configure_optimizers")
        scheds = []
        if isinstance(cfg, dict):
            opts = [cfg["optimizer"]]
            s = cfg.get("lr_scheduler")
            if s is not None:
                s = dict(s) if isinstance(s, dict) else {"scheduler": s}
                s.setdefault("interval", "epoch")
                s.setdefault("frequency", 1)
                scheds = [s]
        elif isinstance(cfg, (list, tuple)):
            opts = list(cfg)
        else:
            opts = [cfg]
        self.optimizers, self.lr_scheduler_configs = opts, scheds

    def fit(self, model, train_dataloaders=None, val_dataloaders=None, ckpt_path=None):
        module = model
        self.lightning_module = module
        module.trainer = self
        self.state_fn = "fit"
        self.train_dataloader, self.val_dataloader = train_dataloaders, val_dataloaders
        self.num_training_batches = len(train_dataloaders)
        self._verify(module)
        self._call(module, "setup", "fit")
        self.strategy.setup(self, module)         # (model_to_device is the test's business; DDP wrap; then the optimizers)
        self._setup_optimizers(module)
        if ckpt_path is not None:
            self._restore(ckpt_path, module)
        self._call(module, "on_fit_start")
        module.train()
        self._call(module, "on_train_start")
        while self.current_epoch < self.max_epochs and not self.should_stop:
            self._run_epoch(module)
            self.current_epoch += 1
        self._call(module, "on_train_end")
        self._call(module, "on_fit_end")
        self.state_fn = None

    def _run_epoch(self, module):
        self._call(module, "on_train_epoch_start")
        n_batches = len(self.train_dataloader)
        for batch_idx, batch in enumerate(self.train_dataloader):
            self._call(module, "on_train_batch_start", batch, batch_idx)
            accumulate = (batch_idx + 1) % self.accumulate_grad_batches != 0 and batch_idx + 1 != n_batches
            if module.automatic_optimization:
                out = self._automatic(module, batch, batch_idx, accumulate)
            else:
                out = self._call_training_step(module, batch, batch_idx)     # loops/optimization/manual.py
            # training_epoch_loop.py: advance — schedulers with interval "step" (automatic optimization only)
            if module.automatic_optimization and not accumulate:
                for c in self.lr_scheduler_configs:
                    if c["interval"] == "step" and (self.global_step % c["frequency"]) == 0:
                        self._call(module, "lr_scheduler_step", c["scheduler"], None)
            self._callbacks("on_train_batch_end", module, out, batch, batch_idx)
            self._call(module, "on_train_batch_end", out, batch, batch_idx)
            step_metrics = self._results.metrics(on_step=True)
            self.callback_metrics.update(step_metrics)
            self.logged_metrics.update(step_metrics)
            if self.max_steps > 0 and self.global_step >= self.max_steps:
                self.should_stop = True
                break
        self._epochs_done = self.current_epoch + 1
        # training_epoch_loop / fit_loop: on_advance_end — epoch-level values, then the epoch-end hooks, then reset
        epoch_metrics = self._results.metrics(on_step=False)
        self.callback_metrics.update(epoch_metrics)
        self.logged_metrics.update(epoch_metrics)
        self._results.reset()
        if self.val_dataloader is not None:
            self._validate(module)
        self._call(module, "on_train_epoch_end")
        self._callbacks("on_train_epoch_end", module)
        if module.automatic_optimization:
            for c in self.lr_scheduler_configs:
                if c["interval"] == "epoch":
                    self._call(module, "lr_scheduler_step", c["scheduler"], None)

    def _call_training_step(self, module, batch, batch_idx):
        prev = module.__dict__.get("_current_fx_name")
        module.__dict__["_current_fx_name"] = "training_step"
        try:
            self.hook_trace.append("training_step")
            return self.strategy.training_step(batch, batch_idx)
        finally:
            module.__dict__["_current_fx_name"] = prev

    def _automatic(self, module, batch, batch_idx, accumulate):
        # loops/optimization/automatic.py: _AutomaticOptimization.run / Closure
        opt = self.optimizers[0]
        result = {}

        def closure():
            out = self._call_training_step(module, batch, batch_idx)
            loss = out["loss"] if isinstance(out, dict) else out
            result["loss"] = loss
            if batch_idx % self.accumulate_grad_batches == 0:     # _make_zero_grad_fn: the first batch of an accumulation window
                self._call(module, "on_before_zero_grad", opt)
                self._call(module, "optimizer_zero_grad", self.current_epoch, batch_idx, opt)
            if loss is not None:
                self.hook_trace.append("backward")
                self.strategy.backward(loss / self.accumulate_grad_batches if self.accumulate_grad_batches != 1 else loss, opt)
            return loss

        if accumulate:
            closure()
        else:
            self._call(module, "optimizer_step", self.current_epoch, batch_idx, LightningOptimizer(opt, self), closure)
        loss = result.get("loss")
        return None if loss is None else {"loss": loss.detach()}

    def _validate(self, module):
        self._call(module, "on_validation_model_eval")
        res, self._results = self._results, _ResultCollection()
        with torch.no_grad():
            self._call(module, "on_validation_epoch_start")
            for i, batch in enumerate(self.val_dataloader):
                self._call(module, "validation_step", batch, i)
            self._call(module, "on_validation_epoch_end")
        m = self._results.metrics(on_step=False)
        self.callback_metrics.update(m)
        self.logged_metrics.update(m)
        self._results.reset()
        self._results = res
        self._call(module, "on_validation_model_train")
        self._callbacks("on_validation_end", module)

    # ---- checkpoints (trainer/connectors/checkpoint_connector.py: dump_checkpoint) ----
    def save_checkpoint(self, path):
        m = self.lightning_module
        ck: dict[str, Any] = {
            "epoch": self.current_epoch, "global_step": self.global_step, "pytorch-lightning_version": "2.5.0",
            "state_dict": m.state_dict(),
            "optimizer_states": [o.state_dict() for o in self.optimizers],
            "lr_schedulers": [c["scheduler"].state_dict() for c in self.lr_scheduler_configs],
            "hyper_parameters": dict(m.hparams),
            "loops": {"auto_step": self._auto_step.completed, "manual_step": self._manual_step.completed, "epochs_done": self._epochs_done},
        }
        m.on_save_checkpoint(ck)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(ck, path)

    def _restore(self, path, module):
        ck = torch.load(path, map_location=module.device, weights_only=False)
        module.on_load_checkpoint(ck)
        module.load_state_dict(ck["state_dict"])
        for o, sd in zip(self.optimizers, ck["optimizer_states"]):
            o.load_state_dict(sd)
        for c, sd in zip(self.lr_scheduler_configs, ck["lr_schedulers"]):
            c["scheduler"].load_state_dict(sd)
        self._auto_step.completed = self._auto_step.ready = ck["loops"]["auto_step"]
        self._manual_step.completed = self._manual_step.ready = ck["loops"]["manual_step"]
        # (a checkpoint written at or after the end of an epoch resumes with the next one)
        self.current_epoch = self._epochs_done = int(ck["loops"].get("epochs_done", ck["epoch"] + 1))


def install_into(sys_modules: dict) -> None:
    """Make this module what ``import lightning.pytorch as pl`` finds (called by ``oracle/ref_shim.install``)."""
    lp = sys_modules["lightning.pytorch"]
    lp.LightningModule = LightningModule
    lp.Trainer = Trainer
    lp.Callback = Callback
    cb = sys_modules["lightning.pytorch.callbacks"]
    cb.ModelCheckpoint, cb.EarlyStopping, cb.Callback = ModelCheckpoint, EarlyStopping, Callback
    st = sys_modules["lightning.pytorch.strategies"]
    st.DDPStrategy = DDPStrategy
    sys_modules["lightning.pytorch.core.mixins"].HyperparametersMixin = HyperparametersMixin
    sys_modules["lightning.fabric.utilities.data"].AttributeDict = _AttributeDict
    ex = sys_modules.get("lightning.pytorch.utilities.exceptions")
    if ex is not None:
        ex.MisconfigurationException = MisconfigurationException
