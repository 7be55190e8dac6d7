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


def hip_mpnn_class():
    """Build (once) ``class HipMPNN(chemprop.models.MPNN)``: the reference's LightningModule whose ``training_step``
    (``models/model.py:148-161``) + optimizer step (``:208-231``) is ONE ``dmpnn_train_step`` call per batch.

    * ``automatic_optimization = False`` (Lightning's manual optimization): the step — K0, block forward, aggregation, batch norm,
      predictor, criterion, the backward pass of all of it, Adam — is enqueued by :class:`chemprop_amd.model.FusedTrainer`; the
      parameters live in ONE flat buffer (``optim.FlatAdam``: ``torch.optim.Adam``'s arithmetic as one launch), the gradients in
      another (``distributed.GradSync``: with more than one rank the step exchanges them itself over ``torch.distributed``'s
      default group, in two slices, so run one process per GPU with a strategy that does NOT wrap the module in DDP).
    * ``configure_optimizers`` is inherited: the reference's ``Adam`` + Noam-like ``LambdaLR`` (``schedulers.py``) stay the source
      of the learning rate — every step reads ``optimizer.param_groups[0]["lr"]`` and advances the scheduler — but the torch
      optimizer itself is never stepped.  The moments travel in the checkpoint under ``"hip_flat_adam"``
      (``on_save_checkpoint`` / ``on_load_checkpoint``).
    * What the fused step does not implement (``FusedTrainer`` refuses loudly: classification / MVE / evidential heads, predictor
      dropout, ``V_d`` / ``X_d`` inputs, attentive aggregation, atom / mol-atom-bond blocks) trains through the MODULE path in the
      same ``training_step``: the reference's own arithmetic (``super().training_step``) through autograd on the HIP kernels,
      ``loss.backward()``, the same flat Adam.
    * everything else — ``forward``, ``fingerprint``, ``validation_step``, ``predict_step``, ``load_from_checkpoint``, hparams,
      state-dict keys — is the reference's, untouched; the blocks inside are swapped for their HIP subclasses (:func:`accelerate`).
    """
    global _mpnn_cache
    if _mpnn_cache is not None:
        return _mpnn_cache
    try:
        from chemprop.models.model import MPNN as Ref  # noqa: WPS433
    except Exception as e:  # pragma: no cover
        raise ImportError("chemprop_amd.integration needs an importable `chemprop`") from e

    class HipMPNN(Ref):  # type: ignore[misc, valid-type]
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self.automatic_optimization = False
            accelerate(self)
            self.__dict__["_hip"] = None          # (trainer state: built lazily on the device the module was moved to)
            self.__dict__["_hip_adam_state"] = None

        # ---- the flat optimizer state shared by the fused step and the module path ----
        def _hip_state(self):
            st = self.__dict__.get("_hip")
            dev = next(self.parameters()).device
            if st is not None and st["dev"] == dev:
                return st
            from .distributed import GradSync
            from .model import FusedTrainer
            from .optim import FlatAdam

            st = {"dev": dev, "fused": None, "why": None}
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

        def _hip_lr_and_sched(self):
            """The reference's schedule as the source of this step's learning rate (``configure_optimizers``: Adam + LambdaLR)."""
            try:
                opt, sch = self.optimizers(), self.lr_schedulers()
            except Exception:   # (no trainer attached: a bare loop drives training_step)
                return float(self.init_lr), None
            if isinstance(opt, (list, tuple)):
                opt = opt[0]
            if isinstance(sch, (list, tuple)):
                sch = sch[0]
            return float(opt.param_groups[0]["lr"]), sch

        def training_step(self, batch, batch_idx):
            bmg, V_d, X_d, targets, weights, lt_mask, gt_mask = batch
            st = self._hip_state()
            lr, sch = self._hip_lr_and_sched()
            loss = None
            tr = st["fused"]
            if tr is not None and V_d is None and X_d is None and self.training:
                try:
                    out = tr.step(bmg, targets, weights, lt_mask, gt_mask, lr=lr)
                    loss = out[0]
                    st["route"] = "fused:" + str(tr.last_route)
                except NotImplementedError as e:   # (a batch the fused step refuses, e.g. dropout on a batch beyond the tile kernels)
                    st["why"] = str(e)
            if loss is None:
                # the module path: the reference's own training_step arithmetic through autograd, the same flat Adam
                sync, opt = st["sync"], st["opt"]
                sync.wait()
                sync.zero_grad()
                l = None
                if X_d is None and self.training:
                    # everything behind the block as ONE autograd node on the head kernels where they implement this model
                    # (chemprop_amd.model.head_loss: aggregation, batch norm, predictor, criterion + their backward in one call)
                    from .model import criterion_kind, head_loss

                    bounded = criterion_kind(self.criterion)[1]
                    l = head_loss(self, self.message_passing(bmg, V_d), bmg.batch, len(bmg), targets, weights,
                                  lt_mask if bounded else None, gt_mask if bounded else None)
                if l is None:
                    l = Ref.training_step(self, batch, batch_idx)   # the reference's own arithmetic, torch ops behind the block
                l.backward()
                sync.allreduce()
                opt.step(lr)
                loss = l.detach()
                st["route"] = "module"
            if sch is not None:
                sch.step()
            # (the reference logs the criterion Metric object — epoch value = sum L / sum mask; the batch's scalar weighted by the
            #  batch size is the same number whenever no target is missing)
            self.log("train_loss", loss, batch_size=len(bmg), prog_bar=True, on_epoch=True)
            return loss

        # ---- the flat Adam's moments in Lightning's checkpoint ----
        def on_save_checkpoint(self, checkpoint) -> None:
            st = self.__dict__.get("_hip")
            if st is not None:
                checkpoint["hip_flat_adam"] = {k: (v.cpu() if hasattr(v, "cpu") else v) for k, v in st["opt"].state_dict().items()}

        def on_load_checkpoint(self, checkpoint) -> None:
            sd = checkpoint.get("hip_flat_adam")
            if sd is not None:
                self.__dict__["_hip_adam_state"] = sd
                self.__dict__["_hip"] = None

        def _apply(self, fn, *args, **kwargs):
            st = self.__dict__.get("_hip")
            if st is not None:   # (a device move re-creates the flat buffers: carry the moments over)
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
