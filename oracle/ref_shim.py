"""Import shim that lets the reference's hot-path SOURCE run, verbatim, in this container.

TEST INFRASTRUCTURE ONLY.  Nothing in ``chemprop_amd/`` may import this module.  It is used by
``tests/golden/make_golden.py`` (to freeze golden vectors from the executed reference) and by the
``-m "not gpu"`` tests that re-validate the restated oracle against the executed reference when
``/root/reference`` is present.  On the GPU box ``/root/reference`` does not exist; what exists there is
``oracle/_ref/`` — a git-ignored staging copy of exactly the reference files this shim imports, made by
``oracle/stage_ref.py`` at build time (never committed) — so the ``-m gpu`` tests can run the REAL
``chemprop.nn.BondMessagePassing`` subclass, ``MulticomponentMessagePassing`` and ``GraphTransform`` on a device.

Why a shim: ``import chemprop`` needs rdkit / lightning / torchmetrics / cuik_molmaker / astartes /
configargparse (none installed, no network) and Python >= 3.11 (``enum.StrEnum``, ``typing.Self``).
None of that is touched by ``chemprop/nn/message_passing/{base,mixins}.py`` at run time, so the
missing modules are replaced by inert stand-ins and the reference files themselves are imported from
where they lie.  No reference source is copied or modified.
"""
from __future__ import annotations

import enum
import os
import sys
import types
import typing

_HERE = os.path.dirname(os.path.abspath(__file__))
STAGED_ROOT = os.path.join(_HERE, "_ref")  # git-ignored copy made by oracle/stage_ref.py (travels to the GPU box)


def _default_root() -> str:
    env = os.environ.get("CHEMPROP_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isfile("/root/reference/chemprop/nn/message_passing/base.py"):
        return "/root/reference"
    return STAGED_ROOT


REFERENCE_ROOT = _default_root()


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "chemprop", "nn", "message_passing", "base.py"))


class _StubMeta(type):
    """Metaclass of auto-created stand-in classes: attribute access manufactures more stand-ins,
    so ``HybridizationType.SP`` or ``Chem.rdchem.BondType.SINGLE`` exist and are hashable."""

    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        sub = _StubMeta(name, (), {"__module__": cls.__module__})
        type.__setattr__(cls, name, sub)
        return sub

    def __call__(cls, *a, **k):
        return cls

    def __iter__(cls):
        return iter(())

    def __or__(cls, other):
        return typing.Union[cls, other]

    def __ror__(cls, other):
        return typing.Union[other, cls]


class _StubModule(types.ModuleType):
    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if full in sys.modules:
            return sys.modules[full]
        sub = _StubMeta(name, (), {"__module__": self.__name__})
        setattr(self, name, sub)
        return sub


def _stub_module(name: str) -> types.ModuleType:
    if name in sys.modules:
        return sys.modules[name]
    mod = _StubModule(name)
    sys.modules[name] = mod
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(_stub_module(parent), child, mod)
    return mod


_INSTALLED = False


def install() -> None:
    """Install the stand-ins and put the reference on ``sys.path`` (idempotent)."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")

    if not hasattr(enum, "StrEnum"):

        class StrEnum(str, enum.Enum):
            def __str__(self):
                return str(self.value)

            @staticmethod
            def _generate_next_value_(name, start, count, last_values):
                return name.lower()

        enum.StrEnum = StrEnum
    if not hasattr(typing, "Self"):
        typing.Self = typing.Any

    for name in (
        "rdkit", "rdkit.Chem", "rdkit.Chem.rdchem", "rdkit.Chem.Descriptors", "rdkit.Chem.AllChem",
        "rdkit.Chem.rdFingerprintGenerator", "rdkit.Chem.Scaffolds",
        "rdkit.Chem.Scaffolds.MurckoScaffold", "rdkit.Chem.rdmolops", "rdkit.DataStructs",
        "rdkit.Chem.rdMolDescriptors", "rdkit.Chem.Draw", "rdkit.RDLogger",
        "cuik_molmaker", "astartes", "astartes.molecules", "descriptastorus",
        "descriptastorus.descriptors", "myerson", "configargparse",
        "torchmetrics", "torchmetrics.classification", "torchmetrics.regression",
        "torchmetrics.functional", "torchmetrics.functional.classification",
        "torchmetrics.utilities", "torchmetrics.utilities.compute", "torchmetrics.utilities.data",
        "torchmetrics.utilities.checks", "torchmetrics.metric",
        "lightning", "lightning.pytorch", "lightning.pytorch.core", "lightning.pytorch.core.mixins",
        "lightning.pytorch.callbacks", "lightning.pytorch.loggers", "lightning.pytorch.strategies",
        "lightning.fabric", "lightning.fabric.utilities", "lightning.fabric.utilities.data",
        "lightning.pytorch.utilities", "lightning.pytorch.utilities.types",
    ):
        _stub_module(name)

    import torch.nn as nn

    lightning = sys.modules["lightning"]
    lightning.__version__ = "2.5.0"
    _stub_module("lightning.pytorch.utilities.exceptions")
    # lightning.pytorch: LightningModule / Trainer / ModelCheckpoint / EarlyStopping / DDPStrategy as restated control flow
    # (oracle/lightning_shim.py) — what `chemprop train` drives around MPNN (cli/train.py:1912-1999)
    from oracle import lightning_shim

    lightning_shim.install_into(sys.modules)

    # torchmetrics: the reference's losses / metrics subclass torchmetrics.Metric and an MPNN keeps them in an
    # nn.ModuleList (models/model.py:99-103), so the stand-in has to be a real nn.Module (state kept as attributes).
    import copy as _copy

    class Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self._shim_states = {}

        def add_state(self, name, default, dist_reduce_fx=None, persistent=False):
            self._shim_states[name] = default
            setattr(self, name, default.clone() if hasattr(default, "clone") else default)

        def forward(self, *a, **k):
            """``torchmetrics.Metric.forward`` for ``full_state_update = False`` (what ChempropMetric declares, metrics.py:62-63):
            the value of THIS batch — ``update`` on a fresh state, ``compute`` — while the batch's state is also added to the
            accumulated one (``dist_reduce_fx="sum"`` states).  ``MPNN.training_step`` back-propagates through this value."""
            acc = {n: getattr(self, n) for n in self._shim_states}
            for n, d in self._shim_states.items():
                setattr(self, n, d.clone() if hasattr(d, "clone") else d)
            self.update(*a, **k)
            val = self.compute()
            for n in self._shim_states:
                setattr(self, n, acc[n] + getattr(self, n).detach())
            self._forward_cache = val.detach() if hasattr(val, "detach") else val   # (what Lightning logs `on_step` for a Metric value)
            return val

        def _apply(self, fn, *a, **k):
            """``torchmetrics.Metric._apply``: the states (and their defaults) move with the module (``.to(device)``)."""
            super()._apply(fn, *a, **k)
            for n, d in list(self._shim_states.items()):
                if hasattr(d, "clone"):
                    self._shim_states[n] = fn(d)
                    setattr(self, n, fn(getattr(self, n)))
            return self

        def clone(self):
            return _copy.deepcopy(self)

        def reset(self):
            """``torchmetrics.Metric.reset``: the states return to their defaults (Lightning calls it at the end of an epoch)."""
            for n, d in self._shim_states.items():
                cur = getattr(self, n, None)
                v = d.clone() if hasattr(d, "clone") else d
                if hasattr(v, "to") and hasattr(cur, "device"):
                    v = v.to(cur.device)
                setattr(self, n, v)

    tm = sys.modules["torchmetrics"]
    tm.Metric = Metric
    sys.modules["torchmetrics.metric"].Metric = Metric
    for mod, names in (("torchmetrics", ("R2Score",)),
                       ("torchmetrics.classification", ("BinaryAUROC", "BinaryPrecisionRecallCurve", "BinaryAccuracy", "BinaryF1Score")),
                       ("torchmetrics.regression", ("R2Score",))):
        for n in names:
            setattr(sys.modules[mod], n, type(n, (Metric,), {"__module__": mod}))

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _INSTALLED = True


def load_reference_extras():
    """``(MulticomponentMessagePassing, GraphTransform, ScaleTransform, MPNN, nn)`` of the reference (``nn`` = ``chemprop.nn``)."""
    install()
    import chemprop.nn as cnn  # noqa: E402
    from chemprop.models.model import MPNN  # noqa: E402
    from chemprop.nn.message_passing.multi import MulticomponentMessagePassing  # noqa: E402
    from chemprop.nn.transforms import GraphTransform, ScaleTransform  # noqa: E402

    return MulticomponentMessagePassing, GraphTransform, ScaleTransform, MPNN, cnn


def load_reference():
    """Return ``(BondMessagePassing, BatchMolGraph, MolGraph)`` — the reference's own classes."""
    install()
    from chemprop.data.collate import BatchMolGraph  # noqa: E402
    from chemprop.data.molgraph import MolGraph  # noqa: E402
    from chemprop.nn.message_passing.base import BondMessagePassing  # noqa: E402

    return BondMessagePassing, BatchMolGraph, MolGraph
