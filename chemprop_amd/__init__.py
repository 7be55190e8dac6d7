"""chemprop_amd — MI355X (gfx950) engine for ONE path of chemprop: BondMessagePassing.forward.

Layout:  ``csrc/`` HIP kernels + the C-ABI of ``include/dmpnn.h``;  ``engine`` ctypes host side;
``nn`` / ``mab`` / ``agg`` / ``data`` host-side mirrors of the reference interface for this path and its
neighbours (atom messages, mol-atom-bond blocks, aggregation, batching);  ``integration`` drop-in subclasses for an
installed chemprop;  ``synth`` synthetic molecule-shaped batches.
"""
from .data import BatchMolGraph, MolGraph, PackedBatch  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    if name in ("BondMessagePassing", "AtomMessagePassing", "InvalidShapeError"):
        from . import nn as _nn

        return getattr(_nn, name)
    if name in ("MABBondMessagePassing", "MABAtomMessagePassing"):
        from . import mab as _mab

        return getattr(_mab, name)
    if name in ("enable", "enabled", "accelerate"):
        from . import integration as _integration

        return getattr(_integration, name)
    raise AttributeError(name)


def _enable_from_env() -> None:
    """``CHEMPROP_MI355X=1``: rebind the reference's classes at import (no new CLI flag) — when chemprop imports."""
    import os

    if os.environ.get("CHEMPROP_MI355X", "") not in ("1", "true", "on"):
        return
    try:
        from .integration import enable

        enable()
    except ImportError as e:  # (chemprop is not installed next to this package: nothing to rebind)
        import warnings

        warnings.warn(f"CHEMPROP_MI355X=1 but chemprop does not import ({e}); chemprop_amd.enable() skipped")


_enable_from_env()
