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


def __getattr__(name):
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
