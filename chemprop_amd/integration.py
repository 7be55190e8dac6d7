"""Drop-in for an installed chemprop: a subclass of the REAL ``chemprop.nn.BondMessagePassing`` that
overrides ``forward`` only.  Parameters, ``hparams`` (``cls`` is reported as the
reference class so a saved checkpoint loads in stock chemprop, ``models/model.py:267-271``),
``state_dict`` keys, ``output_dim``, ``graph_transform`` / ``V_d_transform`` and every attribute the
CLI reads (``cli/predict.py:256-263``, ``cli/train.py:943-969,1826-1828``) are inherited untouched, so
Lightning training / evaluation and the CLI run as they are.

chemprop (with rdkit, lightning, ...) is NOT importable in the build container; this module imports
it lazily and raises a clear error when it is missing.  The parity tests therefore exercise the
state-dict-compatible mirror ``chemprop_amd.nn.BondMessagePassing`` against the executed reference.

    from chemprop_amd.integration import HipBondMessagePassing, accelerate
    mp = HipBondMessagePassing(d_h=300, depth=3)            # instead of chemprop.nn.BondMessagePassing
    model = chemprop.models.MPNN(mp, agg, ffn, ...)          # everything else unchanged
    accelerate(existing_model)                               # or: swap the block of a built / loaded model
"""
from __future__ import annotations

from typing import Optional

from torch import Tensor

from .nn import EngineStateMixin, atom_message_passing_forward, bond_message_passing_forward

_cls_cache = None


def _reference_class():
    try:
        from chemprop.nn import BondMessagePassing as Ref  # noqa: WPS433
    except Exception as e:  # pragma: no cover - chemprop is absent from the build container
        raise ImportError(
            "chemprop_amd.integration needs an importable `chemprop` (with rdkit / lightning); "
            "without it use the state-dict-compatible mirror chemprop_amd.nn.BondMessagePassing") from e
    return Ref


def hip_bond_message_passing_class():
    """Build (once) ``class HipBondMessagePassing(chemprop.nn.BondMessagePassing)``."""
    global _cls_cache
    if _cls_cache is not None:
        return _cls_cache
    Ref = _reference_class()

    class HipBondMessagePassing(EngineStateMixin, Ref):  # type: ignore[misc, valid-type]
        """``chemprop.nn.BondMessagePassing`` whose ``forward`` (base.py:196-212) runs on the MI355X
        HIP kernels.  Raises on non-HIP tensors: there is no CPU fallback inside the engine — keep the
        stock class for CPU runs."""

        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self.hparams["cls"] = Ref  # checkpoints stay loadable by stock chemprop

        def forward(self, bmg, V_d: Optional[Tensor] = None) -> Tensor:
            return bond_message_passing_forward(self, bmg, V_d)

    _cls_cache = HipBondMessagePassing
    return HipBondMessagePassing


_atom_cache = None


def hip_atom_message_passing_class():
    """``class HipAtomMessagePassing(chemprop.nn.AtomMessagePassing)`` (f2): ``forward`` only."""
    global _atom_cache
    if _atom_cache is not None:
        return _atom_cache
    try:
        from chemprop.nn import AtomMessagePassing as Ref  # noqa: WPS433
    except Exception as e:  # pragma: no cover
        raise ImportError("chemprop_amd.integration needs an importable `chemprop`") from e

    class HipAtomMessagePassing(Ref):  # type: ignore[misc, valid-type]
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self.hparams["cls"] = Ref

        def forward(self, bmg, V_d: Optional[Tensor] = None) -> Tensor:
            return atom_message_passing_forward(self, bmg, V_d)

    _atom_cache = (Ref, HipAtomMessagePassing)
    return _atom_cache


_mab_cache = None


def hip_mab_message_passing_classes():
    """``{reference class: HIP subclass}`` for ``chemprop.nn.MABBondMessagePassing`` / ``MABAtomMessagePassing``
    (f2, ``mol_atom_bond.py:284-388``): ``forward(bmg, V_d, E_d) -> (H_v, H_e)`` only is overridden."""
    global _mab_cache
    if _mab_cache is not None:
        return _mab_cache
    try:
        from chemprop.nn import MABAtomMessagePassing, MABBondMessagePassing  # noqa: WPS433
    except Exception as e:  # pragma: no cover
        raise ImportError("chemprop_amd.integration needs an importable `chemprop`") from e
    from .mab import mab_forward

    out = {}
    for Ref, atom in ((MABBondMessagePassing, False), (MABAtomMessagePassing, True)):
        def make(Ref=Ref, atom=atom):
            class Hip(Ref):  # type: ignore[misc, valid-type]
                atom_messages = atom

                def __init__(self, *args, **kwargs):
                    super().__init__(*args, **kwargs)
                    self.hparams["cls"] = Ref  # checkpoints stay loadable by stock chemprop

                def forward(self, bmg, V_d: Optional[Tensor] = None, E_d: Optional[Tensor] = None):
                    return mab_forward(self, bmg, V_d, E_d)

            Hip.__name__ = Hip.__qualname__ = "Hip" + Ref.__name__
            return Hip

        out[Ref] = make()
    _mab_cache = out
    return out


_mlp_cache = None


def hip_mlp_class():
    """``(chemprop.nn.ffn.MLP, HipMLP)`` (f4, ``nn/ffn.py:24-68``): the predictor's feed-forward stack; ``forward`` only."""
    global _mlp_cache
    if _mlp_cache is not None:
        return _mlp_cache
    try:
        from chemprop.nn.ffn import MLP as Ref  # noqa: WPS433
    except Exception as e:  # pragma: no cover
        raise ImportError("chemprop_amd.integration needs an importable `chemprop`") from e
    from .ffn import mlp_forward

    class HipMLP(Ref):  # type: ignore[misc, valid-type]
        def forward(self, X: Tensor) -> Tensor:
            return mlp_forward(self, X)

    _mlp_cache = (Ref, HipMLP)
    return _mlp_cache


_mpnn_cache = None


def _hip_adam_class():
    """``torch.optim.Optimizer`` face of :class:`chemprop_amd.optim.FlatAdam` — what ``HipMPNN.configure_optimizers`` hands Lightning
    in place of the reference's ``torch.optim.Adam`` (``models/model.py:208-209``).  Same param groups (the learning-rate scheduler
    writes ``param_groups[0]["lr"]`` as before), ``step(closure)`` with Lightning's automatic-optimization closure semantics, and a
    ``state_dict`` in ``torch.optim.Adam``'s own format, so a checkpoint moves between this optimizer and the stock one."""
    import weakref

    import torch

    class HipAdam(torch.optim.Optimizer):
        def __init__(self, module, param_groups, defaults):
            groups = [{k: v for k, v in g.items()} for g in param_groups]
            super().__init__(groups, dict(defaults))
            self._module = weakref.ref(module)
            self._pending = None          # (a state dict loaded before the flat buffers exist)

        # -- the flat buffers live with the module (they follow its device); asked for at every use --
        def _st(self):
            m = self._module()
            if m is None:
                raise RuntimeError("HipAdam: the module it optimises is gone")
            st = m._hip_state()
            if self._pending is not None:
                sd, self._pending = self._pending, None
                self._load_torch(st, sd)
            return st

        def _all_params(self):
            return [p for g in self.param_groups for p in g["params"]]

        @torch.no_grad()
        def step(self, closure=None):
            loss = None
            if closure is not None:
                with torch.enable_grad():
                    loss = closure()     # (Lightning, automatic optimization: training_step -> zero_grad -> backward -> clipping)
            m = self._module()
            if m is None:
                raise RuntimeError("HipAdam: the module it optimises is gone")
            if m.__dict__.pop("_hip_applied", False):
                return loss              # the fused call of this closure already held the backward pass, the clip AND this update
            st = self._st()
            g, fl = self.param_groups[0], st["opt"]
            fl.betas, fl.eps, fl.weight_decay = (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"]), float(g["weight_decay"])
            fl.step(float(g["lr"]))
            return loss

        def zero_grad(self, set_to_none: bool = True) -> None:
            self._st()["sync"].zero_grad()    # (one fill of the flat buffer; every p.grad stays its view)

        # -- torch.optim.Adam's state-dict format: state[i] = {step, exp_avg, exp_avg_sq} for the i-th parameter of the groups --
        def state_dict(self):
            st = self._st()
            fl, sync = st["opt"], st["sync"]
            where = {id(p): (o, p.numel()) for p, o in zip(sync.params, sync.offsets)}
            state, groups, i = {}, [], 0
            for g in self.param_groups:
                idx = []
                for p in g["params"]:
                    if fl.steps > 0 and id(p) in where:
                        o, n = where[id(p)]
                        state[i] = {"step": torch.tensor(float(fl.steps)), "exp_avg": fl.m[o:o + n].view_as(p).clone(),
                                    "exp_avg_sq": fl.v[o:o + n].view_as(p).clone()}
                    idx.append(i)
                    i += 1
                groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": idx})
            return {"state": state, "param_groups": groups}

        def load_state_dict(self, sd):
            for g, sg in zip(self.param_groups, sd["param_groups"]):
                for k, v in sg.items():
                    if k != "params":
                        g[k] = v
            m = self._module()
            if m is not None and m.__dict__.get("_hip") is not None:
                self._load_torch(m._hip_state(), sd)
            else:
                self._pending = sd

        def _load_torch(self, st, sd):
            fl, sync = st["opt"], st["sync"]
            where = {id(p): (o, p.numel()) for p, o in zip(sync.params, sync.offsets)}
            fl.m.zero_()
            fl.v.zero_()
            fl.steps = 0
            for i, p in enumerate(self._all_params()):
                e = sd["state"].get(i)
                if e is None or id(p) not in where:
                    continue
                o, n = where[id(p)]
                fl.m[o:o + n].view_as(p).copy_(e["exp_avg"].to(fl.m.device))
                fl.v[o:o + n].view_as(p).copy_(e["exp_avg_sq"].to(fl.v.device))
                fl.steps = max(fl.steps, int(e["step"]))

    return HipAdam


def _should_accumulate(tr, batch_idx: int, accumulate: int) -> bool:
    """Does the Trainer only accumulate on this micro-batch (no optimizer step behind it)?  Lightning's own answer where it has one
    (``loops/training_epoch_loop.py: _should_accumulate``), else its rule: step every ``accumulate`` batches and on the epoch's last."""
    loop = getattr(getattr(tr, "fit_loop", None), "epoch_loop", None)
    fn = getattr(loop, "_should_accumulate", None)
    if callable(fn):
        try:
            return bool(fn())
        except Exception:
            pass
    n = getattr(tr, "num_training_batches", None)
    if not isinstance(n, int):
        dl = getattr(tr, "train_dataloader", None)
        try:
            n = len(dl) if dl is not None else None
        except TypeError:
            n = None
    return (batch_idx + 1) % accumulate != 0 and (n is None or batch_idx + 1 != n)


def _clip_algorithm(a) -> str:
    """``"norm"`` / ``"value"`` from what Lightning holds (``None``: its default, norm; a ``GradClipAlgorithmType`` str-enum; a string)."""
    if a is None:
        return "norm"
    return str(getattr(a, "value", a)).lower()


def hip_mpnn_class():
    """Build (once) ``class HipMPNN(chemprop.models.MPNN)``: the reference's LightningModule whose ``training_step``
    (``models/model.py:148-161``) + optimizer step (``:208-231``) is ONE ``dmpnn_train_step`` call per batch — under Lightning's
    AUTOMATIC optimization, i.e. under the ``Trainer`` that ``chemprop train`` builds (``cli/train.py:1912-1999``) as it is:

    * ``configure_optimizers`` returns the reference's own dictionary with the ``torch.optim.Adam`` replaced by ``HipAdam`` (same
      param groups; one flat buffer of parameters, one of gradients, one launch per update: ``optim.FlatAdam``) and the reference's
      Noam-like ``LambdaLR`` re-created on it with the reference's own ``lr_lambda`` (``schedulers.py``).  Lightning steps the
      scheduler, counts ``trainer.global_step`` through ``LightningOptimizer.step`` (what ``ModelCheckpoint`` keys on,
      ``cli/train.py:1912-1919``), saves ``optimizer.state_dict()`` (``torch.optim.Adam``'s format) in its checkpoints.
    * ``training_step``: where :class:`chemprop_amd.model.FusedTrainer` applies (a bond block with a built-in activation, sum / mean /
      norm aggregation, batch norm, regression MLP, MSE / MAE, no ``V_d`` / ``X_d``) the whole step — K0, forward, head, backward,
      clip, Adam — is enqueued by that one call with this step's learning rate and ``Trainer(gradient_clip_val)``
      (``cli/train.py:1937``); the hooks Lightning runs afterwards inside ``optimizer.step(closure)`` — ``backward``,
      ``configure_gradient_clipping``, the optimizer's own update — find the work done and return.  Everything else is the MODULE
      path: the reference's arithmetic through autograd on the HIP kernels, returned as a loss with a graph; Lightning's closure runs
      ``backward`` (→ the gradient exchange), the clip over the flat buffer, ``HipAdam.step``.  ``train_loss`` is logged ONCE per step.
    * more than one rank (``--devices N``: ``DDPStrategy`` wraps the module, ``cli/train.py:1934,1943``): the module owns the gradient
      exchange — ONE flat all-reduce over ``torch.distributed``'s default group (RCCL), in two slices under the fused step — and
      switches the wrapper's own reducer off (``require_backward_grad_sync = False``, what ``no_sync()`` sets): the fused step never
      runs autograd, and the module path would otherwise reduce twice.
    * without a ``Trainer`` ``training_step`` is the reference's: a loss with a graph, nothing updated.
    * ``forward``, ``fingerprint``, ``validation_step``, ``predict_step``, ``load_from_checkpoint``, hparams, state-dict keys are
      the reference's, untouched; the blocks inside are swapped for their HIP subclasses (:func:`accelerate`).
    """
    global _mpnn_cache
    if _mpnn_cache is not None:
        return _mpnn_cache
    try:
        from chemprop.models.model import MPNN as Ref  # noqa: WPS433
    except Exception as e:  # pragma: no cover
        raise ImportError("chemprop_amd.integration needs an importable `chemprop`") from e
    import torch

    HipAdam = _hip_adam_class()

    class HipMPNN(Ref):  # type: ignore[misc, valid-type]
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            accelerate(self)
            self.__dict__["_hip"] = None          # (flat buffers + fused trainer: built lazily on the device the module was moved to)
            self.__dict__["_hip_adam_state"] = None
            self.__dict__["_hip_optimizer"] = None

        # ---- the flat optimizer state shared by the fused step and the module path ----
        def _hip_state(self):
            st = self.__dict__.get("_hip")
            dev = next(self.parameters()).device
            if st is not None and st["dev"] == dev:
                return st
            from .distributed import GradSync
            from .model import FusedTrainer
            from .optim import FlatAdam

            st = {"dev": dev, "fused": None, "why": None, "route": None}
            try:
                tr = FusedTrainer(self, lr=float(self.init_lr))
                st["fused"], st["sync"], st["opt"] = tr, tr.sync, tr.opt
            except NotImplementedError as e:   # (a model the fused step does not implement: module path on the same flat Adam)
                st["why"] = str(e)
                st["sync"] = GradSync([p for p in self.parameters() if p.requires_grad], modules=[self])
                st["opt"] = FlatAdam(st["sync"], lr=float(self.init_lr))
            saved = self.__dict__.get("_hip_adam_state")
            if saved is not None:
                st["opt"].load_state_dict(saved)
                self.__dict__["_hip_adam_state"] = None
            self.__dict__["_hip"] = st
            return st

        def configure_optimizers(self):
            from torch.optim.lr_scheduler import LambdaLR

            cfg = super().configure_optimizers()          # the reference's: Adam(self.parameters(), init_lr) + Noam-like LambdaLR
            ref_opt, sc = cfg["optimizer"], cfg["lr_scheduler"]
            sched = sc["scheduler"] if isinstance(sc, dict) else sc
            if not isinstance(sched, LambdaLR):
                raise TypeError(f"HipMPNN.configure_optimizers: expected the reference's LambdaLR schedule, got {type(sched).__name__}")
            for g in ref_opt.param_groups:    # (LambdaLR's constructor stamped the start rate; the new scheduler stamps it again)
                g.pop("initial_lr", None)
                g["lr"] = ref_opt.defaults["lr"]
            opt = HipAdam(self, ref_opt.param_groups, ref_opt.defaults)
            new = LambdaLR(opt, list(sched.lr_lambdas) if len(sched.lr_lambdas) > 1 else sched.lr_lambdas[0])
            self.__dict__["_hip_optimizer"] = opt
            cfg["optimizer"] = opt
            cfg["lr_scheduler"] = {**sc, "scheduler": new} if isinstance(sc, dict) else new
            return cfg

        # ---- Lightning's context ----
        def _hip_trainer(self):
            """``(trainer, our optimizer)`` when a Trainer drives this module with the optimizer ``configure_optimizers`` built; else
            ``(trainer or None, None)`` — a user's own optimizer, or no Trainer: the reference's semantics."""
            tr = getattr(self, "_trainer", None)      # (LightningModule.trainer raises when unattached: core/module.py)
            if tr is None:
                return None, None
            opt = self.__dict__.get("_hip_optimizer")
            if opt is None or not any(o is opt for o in getattr(tr, "optimizers", ())):
                return tr, None
            return tr, opt

        def _hip_own_the_exchange(self, tr):
            """More than one rank under a DDP wrap: the gradient exchange is this module's (``GradSync``); the wrapper's reducer stays
            off — what ``DistributedDataParallel.no_sync()`` sets, for good."""
            from torch.nn.parallel import DistributedDataParallel as DDP

            w = getattr(getattr(tr, "strategy", None), "model", None)
            if isinstance(w, DDP) and w.require_backward_grad_sync:
                w.require_backward_grad_sync = False

        def training_step(self, batch, batch_idx):
            tr, opt = self._hip_trainer()
            if opt is None:
                # no Trainer, or a Trainer with an optimizer that is not ours: the reference's step (a loss with a graph; whoever
                # drives the loop calls backward and steps) on the HIP kernels of the swapped blocks
                return Ref.training_step(self, batch, batch_idx)
            bmg, V_d, X_d, targets, weights, lt_mask, gt_mask = batch
            st = opt._st()     # (the flat buffers of this device; a state dict loaded before they existed goes in now)
            self._hip_own_the_exchange(tr)
            self.__dict__["_hip_applied"] = False
            loss, logged = None, False
            fused = st["fused"]
            accumulate = int(getattr(tr, "accumulate_grad_batches", 1) or 1)
            # Trainer(accumulate_grad_batches > 1): Lightning zeroes the gradients on the FIRST micro-batch of a window and steps on
            # the last; the block's backward kernels OVERWRITE their gradient views once per exchange (GradSync._written), so the
            # exchange — which re-arms the views — must run on the stepping micro-batch only (backward() below)
            self.__dict__["_hip_accumulating"] = accumulate > 1 and _should_accumulate(tr, batch_idx, accumulate)
            if fused is not None and V_d is None and X_d is None and self.training and accumulate == 1:
                try:
                    clip = (getattr(tr, "gradient_clip_val", None), _clip_algorithm(getattr(tr, "gradient_clip_algorithm", None)))
                    g = opt.param_groups[0]
                    fl = st["opt"]
                    fl.betas, fl.eps, fl.weight_decay = (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"]), float(g["weight_decay"])
                    out = fused.step(bmg, targets, weights, lt_mask, gt_mask, lr=float(g["lr"]), clip=clip)
                    loss = out[0]
                    self.__dict__["_hip_applied"] = True
                    st["route"] = "fused:" + str(fused.last_route)
                except NotImplementedError as e:   # (a batch the fused step refuses, e.g. dropout on a batch beyond the tile kernels)
                    st["why"] = str(e)
            if loss is None:
                # the module path: a loss with a graph; Lightning's closure runs backward() (below), the clip, HipAdam.step()
                st["sync"].wait()
                if X_d is None and self.training:
                    # everything behind the block as ONE autograd node on the head kernels where they implement this model
                    # (chemprop_amd.model.head_loss: aggregation, batch norm, predictor, criterion + their backward in one call)
                    from .model import criterion_kind, head_loss

                    bounded = criterion_kind(self.criterion)[1]
                    loss = head_loss(self, self.message_passing(bmg, V_d), bmg.batch, len(bmg), targets, weights,
                                     lt_mask if bounded else None, gt_mask if bounded else None)
                if loss is None:
                    loss = Ref.training_step(self, batch, batch_idx)   # the reference's own arithmetic AND its own train_loss log
                    logged = True
                st["route"] = "module"
            if not logged:
                # (the reference logs the criterion Metric object — epoch value = sum L / sum mask; the batch's scalar weighted by the
                #  batch size is the same number whenever no target is missing)
                self.log("train_loss", loss.detach(), batch_size=len(bmg), prog_bar=True, on_epoch=True)
            return loss

        def on_train_start(self) -> None:
            super().on_train_start()
            tr, opt = self._hip_trainer()
            if opt is not None:
                self._hip_own_the_exchange(tr)

        # ---- the hooks Lightning's automatic optimization runs inside optimizer.step(closure) ----
        def backward(self, loss, *args, **kwargs):
            if self.__dict__.get("_hip_applied"):
                return                       # the fused call held the backward pass
            from .distributed import backward_on_calling_thread

            with backward_on_calling_thread():
                super().backward(loss, *args, **kwargs)
            st = self.__dict__.get("_hip")
            if st is not None and self._hip_trainer()[1] is not None and not self.__dict__.get("_hip_accumulating"):
                # ONE flat all-reduce per optimizer step (no-op on one rank; the clip and the update wait for it on the stream).  Inside
                # an accumulation window the views stay marked as written: the next micro-batch's block backward ADDS to them
                st["sync"].allreduce()

        def configure_gradient_clipping(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
            if self.__dict__.get("_hip_applied"):
                return                       # clipped inside the fused call (dmpnn_step_args.clip_val)
            st = self.__dict__.get("_hip")
            if st is None or self._hip_trainer()[1] is None:
                return super().configure_gradient_clipping(optimizer, gradient_clip_val=gradient_clip_val,
                                                           gradient_clip_algorithm=gradient_clip_algorithm)
            if gradient_clip_val is not None and float(gradient_clip_val) > 0:
                st["opt"].clip_grad(float(gradient_clip_val), _clip_algorithm(gradient_clip_algorithm))

        # ---- device moves re-create the flat buffers: carry the moments over ----
        def _apply(self, fn, *args, **kwargs):
            st = self.__dict__.get("_hip")
            if st is not None:
                self.__dict__["_hip_adam_state"] = {k: (v.cpu() if hasattr(v, "cpu") else v) for k, v in st["opt"].state_dict().items()}
                self.__dict__["_hip"] = None
            return super()._apply(fn, *args, **kwargs)

    _mpnn_cache = (Ref, HipMPNN)
    return _mpnn_cache


def __getattr__(name):
    if name == "HipMPNN":
        return hip_mpnn_class()[1]
    if name == "HipMLP":
        return hip_mlp_class()[1]
    if name == "HipBondMessagePassing":
        return hip_bond_message_passing_class()
    if name == "HipAtomMessagePassing":
        return hip_atom_message_passing_class()[1]
    if name in ("HipMABBondMessagePassing", "HipMABAtomMessagePassing"):
        return {c.__name__: c for c in hip_mab_message_passing_classes().values()}[name]
    raise AttributeError(name)


_agg_cache = None


def hip_aggregation_classes():
    """``{reference class: HIP subclass}`` for ``chemprop.nn.agg`` Mean / Sum / Norm / Attentive aggregation
    (f1: the step after the block, ``models/model.py:131``); ``forward`` only is overridden."""
    global _agg_cache
    if _agg_cache is not None:
        return _agg_cache
    try:
        from chemprop.nn import agg as ref_agg  # noqa: WPS433
    except Exception as e:  # pragma: no cover
        raise ImportError("chemprop_amd.integration needs an importable `chemprop`") from e
    from .agg import aggregation_forward

    out = {}
    for name, mode in (("MeanAggregation", "mean"), ("SumAggregation", "sum"), ("NormAggregation", "norm"),
                       ("AttentiveAggregation", "attentive")):
        Ref = getattr(ref_agg, name)

        def make(Ref=Ref, mode=mode, name=name):
            class Hip(Ref):  # type: ignore[misc, valid-type]
                def __init__(self, *args, **kwargs):
                    super().__init__(*args, **kwargs)
                    self.hparams["cls"] = Ref  # checkpoints stay loadable by stock chemprop

                def forward(self, H: Tensor, batch: Tensor) -> Tensor:
                    return aggregation_forward(self, H, batch, mode)

            Hip.__name__ = Hip.__qualname__ = "Hip" + name
            return Hip

        out[Ref] = make()
    _agg_cache = out
    return out


def accelerate(model, aggregation: bool = True, ffn: bool = True):
    """Swap the class of every ``BondMessagePassing`` / ``AtomMessagePassing`` / ``MAB*MessagePassing`` block (and, unless ``aggregation=False``, of every
    Mean / Sum / Norm / Attentive aggregation and, unless ``ffn=False``, of every ``nn.ffn.MLP`` of the predictors) inside ``model`` (an ``MPNN``, a
    ``MulticomponentMessagePassing`` or the block itself) for the HIP subclass, in place.  No
    parameter is copied or re-created; optimizer state and checkpoints stay valid."""
    Ref = _reference_class()
    Hip = hip_bond_message_passing_class()
    aggs = hip_aggregation_classes() if aggregation else {}
    RefAtom, HipAtom = hip_atom_message_passing_class()
    try:
        mabs = hip_mab_message_passing_classes()
    except ImportError:  # (a chemprop older than the mol-atom-bond blocks)
        mabs = {}
    RefMLP, HipMLP = hip_mlp_class() if ffn else (None, None)
    n = 0
    for m in model.modules():
        if type(m) is Ref:
            m.__class__ = Hip
            n += 1
        elif type(m) is RefAtom:
            m.__class__ = HipAtom
            n += 1
        elif type(m) in aggs:
            m.__class__ = aggs[type(m)]
            n += 1
        elif type(m) in mabs:
            m.__class__ = mabs[type(m)]
            n += 1
        elif RefMLP is not None and type(m) is RefMLP:
            m.__class__ = HipMLP
            n += 1
    return n


# ------------------------------------------------------------------------------------------------------------------
# the switch: `CHEMPROP_MI355X=1 chemprop train ...` / `chemprop-amd train ...` / chemprop_amd.enable()
# ------------------------------------------------------------------------------------------------------------------
_enabled = None

# modules of the reference that bind the classes BY NAME at import time (cli/train.py:60-68, :1500, :1598, :1623; models/utils.py:5;
# cli/predict.py:34-35; cli/fingerprint.py:19) — rebinding the defining module alone would not reach them
_BINDING_MODULES = ("chemprop.nn", "chemprop.nn.message_passing", "chemprop.nn.message_passing.base", "chemprop.nn.message_passing.mol_atom_bond",
                    "chemprop.models", "chemprop.models.model", "chemprop.models.utils", "chemprop.cli.train", "chemprop.cli.predict",
                    "chemprop.cli.fingerprint", "chemprop.cli.hpopt", "chemprop")


def enable(fused_step: bool = True, verbose: bool = False) -> dict:
    """Make an installed chemprop build and load its models on the MI355X engine WITHOUT editing it (SURVEY §5 "Engine selection
    must not need a new CLI flag"): every place the reference binds ``BondMessagePassing`` / ``AtomMessagePassing`` /
    ``MABBondMessagePassing`` / ``MABAtomMessagePassing`` (and, with ``fused_step``, ``MPNN``) by name — ``chemprop.nn``,
    ``chemprop.models``, ``chemprop.cli.train`` (``cli/train.py:60-68``: ``build_model`` at ``:1500,:1598,:1623``) ... — is rebound to
    the HIP subclass, and ``load_model`` / ``MPNN.load_from_file`` (``models/utils.py:27-35``, ``models/model.py:318-329``: predict /
    fingerprint) accelerate what they load.  Checkpoints keep naming the reference classes (``hparams["cls"]``), so they load in a
    stock chemprop.  Idempotent; returns ``{module name: [names rebound]}``.

    Triggered by ``CHEMPROP_MI355X=1`` at ``import chemprop_amd`` (when chemprop is importable), by the ``chemprop-amd`` console entry
    (``chemprop_amd.cli: enable(); chemprop.cli.main.main()``), or by the one-line stub of INTEGRATION.md §2f in ``chemprop/__init__.py``."""
    global _enabled
    import importlib
    import sys

    Ref = _reference_class()
    mapping = {Ref: hip_bond_message_passing_class()}
    RefAtom, HipAtom = hip_atom_message_passing_class()
    mapping[RefAtom] = HipAtom
    try:
        mapping.update(hip_mab_message_passing_classes())
    except ImportError:
        pass
    if fused_step:
        RefM, HipM = hip_mpnn_class()
        mapping[RefM] = HipM
    for name in _BINDING_MODULES:           # (the CLI modules exist only in a full install; bind them if they import)
        if name not in sys.modules and name.startswith("chemprop.cli"):
            try:
                importlib.import_module(name)
            except Exception:
                pass
    done: dict = {}
    for name in _BINDING_MODULES:
        mod = sys.modules.get(name)
        if mod is None:
            continue
        for attr, val in list(vars(mod).items()):
            if isinstance(val, type) and val in mapping:
                setattr(mod, attr, mapping[val])
                done.setdefault(name, []).append(attr)
    # what predict / fingerprint load: MPNN.load_from_file builds the blocks from hparams["cls"] (the reference classes, by design)
    RefM = hip_mpnn_class()[0]
    if not getattr(RefM.load_from_file, "_hip_accelerated", False):
        orig = RefM.load_from_file.__func__

        def load_from_file(cls, *args, **kwargs):
            model = orig(cls, *args, **kwargs)
            accelerate(model)
            return model

        load_from_file._hip_accelerated = True
        RefM.load_from_file = classmethod(load_from_file)
        done.setdefault("chemprop.models.model", []).append("MPNN.load_from_file")
    _enabled = done
    if verbose:
        print(f"chemprop_amd.enable(): {done}")
    return done


def enabled() -> bool:
    return _enabled is not None
