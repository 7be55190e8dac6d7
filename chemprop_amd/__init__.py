"""chemprop_amd — MI355X (gfx950) engine for ONE path of chemprop: BondMessagePassing.forward.

Layout:  ``csrc/`` HIP kernels + the C-ABI of ``include/dmpnn.h``;  ``engine`` ctypes host side;
``nn`` / ``data`` host-side mirror of the reference interface for this path;  ``integration``
drop-in subclass for an installed chemprop;  ``synth`` synthetic molecule-shaped batches.
"""
from .data import BatchMolGraph, MolGraph  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    if name in ("BondMessagePassing", "InvalidShapeError"):
        from . import nn as _nn

        return getattr(_nn, name)
    raise AttributeError(name)
