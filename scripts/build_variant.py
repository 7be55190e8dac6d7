#!/usr/bin/env python
"""Build a variant of the library for kernel A/B experiments on one GPU box:
    python scripts/build_variant.py <tag> [-DNAME[=V] ...]   ->  chemprop_amd/variants/libdmpnn_<tag>.so   (use: DMPNN_LIB=<path>)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chemprop_amd import _lib  # noqa: E402

tag = sys.argv[1]
defs = [a[2:] if a.startswith("-D") else a for a in sys.argv[2:]]
d = os.path.join(ROOT, "chemprop_amd", "variants")
os.makedirs(d, exist_ok=True)
print(_lib.build(force=True, defines=defs, out=os.path.join(d, f"libdmpnn_{tag}.so")))
