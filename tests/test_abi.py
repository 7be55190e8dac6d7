"""The C-ABI library builds, loads and exports every symbol include/dmpnn.h declares; argument
validation returns error codes (never throws / exits).  No compute call is made here (no GPU)."""
import ctypes as C
import os
import re

import pytest

from chemprop_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dmpnn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dmpnn_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_exported():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dmpnn.h but not exported"
    assert set(_lib.EXPORTS) == set(names), set(_lib.EXPORTS) ^ set(names)


def test_version_and_plan_bytes():
    lib = _lib.load()
    assert lib.dmpnn_version() == _lib.ABI_VERSION == 15
    assert lib.dmpnn_plan_bytes(0, 0) >= 64
    b = lib.dmpnn_plan_bytes(4319, 8328)
    assert b % 16 == 0 and b >= 4 * (16 + 9 * 8328 + 2 * 4319)
    off = (C.c_int64 * _lib.PLAN_NOFFSETS)()
    assert lib.dmpnn_plan_layout(10, 20, off) == 0
    arrays = list(off)[:11]
    assert arrays == sorted(arrays) and all(o % 4 == 0 for o in arrays)
    assert off[11] >= (20 + 24) // 25  # tile slots cover the smallest nominal tile stride


def test_can_fuse_is_a_shape_rule():
    """dmpnn_forward_can_fuse inspects shapes / alignment only (no device access)."""
    lib = _lib.load()
    a = _lib.FwdArgs()
    a.n_atoms, a.n_edges, a.d_v, a.d_e, a.d_h, a.depth = 100, 200, 72, 14, 300, 3
    a.ldv, a.lde, a.ldh = 72, 14, 300
    for f in ("V", "E", "W_i", "W_h", "H0", "Ms", "Mv"):
        setattr(a, f, 4096)
    a.ldout = 300
    assert lib.dmpnn_forward_can_fuse(C.byref(a)) == 2   # fused, and small enough for the whole-forward tile kernel
    a.n_atoms, a.n_edges = 100000, 200000
    assert lib.dmpnn_forward_can_fuse(C.byref(a)) == 1   # fused per depth step only
    a.n_atoms, a.n_edges = 100, 200
    a.flags = _lib.F_UNDIRECTED
    assert lib.dmpnn_forward_can_fuse(C.byref(a)) == 0
    a.flags = 0
    a.d_h, a.ldh = 512, 512
    assert lib.dmpnn_forward_can_fuse(C.byref(a)) == 0   # rows wider than one workgroup panel
    a.d_h, a.ldh, a.d_e, a.lde = 300, 300, 13, 13
    assert lib.dmpnn_forward_can_fuse(C.byref(a)) == 0   # odd feature width: 4-byte operand rows


def test_argument_errors_are_codes_not_crashes():
    lib = _lib.load()
    assert lib.dmpnn_forward(None, None) == -1
    assert b"null args" in lib.dmpnn_last_error_string()
    assert lib.dmpnn_linear_fwd(None, None) == -1
    assert lib.dmpnn_prepare(None, None, 4, 8, None, 0, None) == -1
    assert lib.dmpnn_prepare(None, None, -1, 8, None, 0, None) == -1
    assert lib.dmpnn_message_fwd(None, 1, 1, 4, None, 4, None, 4, 0, 0.0, None, 0, None) == -1
    a = _lib.FwdArgs()
    a.plan = 16  # non-null dummy; validation must stop before any dereference
    a.n_atoms, a.n_edges, a.d_v, a.d_e, a.d_h, a.depth, a.act = 1, 0, 4, 2, 8, 3, 0
    assert lib.dmpnn_forward(C.byref(a), None) == -1
    assert b"activation" in lib.dmpnn_last_error_string()


def test_struct_layout_matches_header_field_order():
    """ctypes mirrors must list the fields in the order of the C structs."""
    src = open(os.path.join(ROOT, "include", "dmpnn.h")).read()
    for struct, mirror in (("dmpnn_gemm_args", _lib.GemmArgs), ("dmpnn_fwd_args", _lib.FwdArgs), ("dmpnn_bwd_args", _lib.BwdArgs),
                           ("dmpnn_head_args", _lib.HeadArgs), ("dmpnn_step_args", _lib.StepArgs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                part = re.sub(r"\[[^\]]*\]\s*$", "", part.strip())   # (array members: W[DMPNN_MAX_FFN_LAYERS])
                fields.append(re.findall(r"([A-Za-z_0-9]+)\s*$", part.strip())[0])
        assert fields == [f[0] for f in mirror._fields_], struct


def test_struct_sizes_and_offsets_match_a_c_compiler(tmp_path):
    """sizeof / offsetof of every argument struct as gcc lays out include/dmpnn.h == the ctypes mirrors."""
    import shutil
    import subprocess

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    mirrors = dict(dmpnn_gemm_args=_lib.GemmArgs, dmpnn_fwd_args=_lib.FwdArgs, dmpnn_bwd_args=_lib.BwdArgs, dmpnn_head_args=_lib.HeadArgs,
                   dmpnn_step_args=_lib.StepArgs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dmpnn.h"', "int main(void) {"]
    for name, m in mirrors.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f in m._fields_:
            lines.append(f'printf("{name}.{f[0]} %zu\\n", offsetof({name}, {f[0]}));')
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for name, m in mirrors.items():
        assert int(out[name]) == C.sizeof(m), name
        for f in m._fields_:
            assert int(out[f"{name}.{f[0]}"]) == getattr(m, f[0]).offset, f"{name}.{f[0]}"
    assert _lib.MAX_FFN_LAYERS == 8


def test_build_is_gfx950_only():
    import subprocess

    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", _lib.LIB_PATH],
                         capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("llvm-objdump --offloading unavailable")
    assert "gfx950" in out.stdout
    assert "gfx942" not in out.stdout and "gfx90a" not in out.stdout


def test_tile_plans_of_any_size_shape_rules_and_argument_errors():
    """The loader-table / any-size tile plan entry points: shape rules and argument validation (no device work)."""
    lib = _lib.load()
    a = _lib.FwdArgs()
    a.n_atoms, a.n_edges, a.d_v, a.d_e, a.d_h, a.depth = 100000, 200000, 72, 14, 300, 3
    a.ldv, a.lde, a.ldh, a.ldout = 72, 14, 300, 300
    for f in ("V", "E", "W_i", "W_h", "H0", "Ms", "Mv"):
        setattr(a, f, 4096)
    assert lib.dmpnn_forward_can_fuse(C.byref(a)) == 1
    a.flags = _lib.F_LOADER_TILES
    assert lib.dmpnn_forward_can_fuse(C.byref(a)) == 2   # a tile plan of any batch size lifts the single-workgroup limit
    assert lib.dmpnn_tile_plan_any_size(37000, 73000) == 1 and lib.dmpnn_tile_plan_any_size(300000, 580000) == 1
    assert lib.dmpnn_tile_plan_any_size(5000, 0) == 0      # no room for the planner's scratch in an edge-less plan
    assert lib.dmpnn_tile_plan_any_size(0, 0) == 0
    cap = lib.dmpnn_max_tiles(4636, 9120)
    assert cap >= 2 * (9120 // 48 + 4636 // 32)
    # from_table: null plan, more tiles than the launch bound of the batch size -> error codes, nothing launched
    assert lib.dmpnn_prepare_tiles_from_table(4096, 4096, 10, 100, 200, None, 1 << 20, None) == -1
    assert lib.dmpnn_prepare_tiles_from_table(4096, 4096, cap + 1, 4636, 9120, 4096, 1 << 30, None) == -1
    assert b"launch bound" in lib.dmpnn_last_error_string()
    assert lib.dmpnn_prepare_tiles_from_table(4096, 4096, 10, 4636, 9120, 4096, 64, None) == -3   # plan buffer too small
    # prepare_with_batch (full plan + molecule tiles): the same argument rules as dmpnn_prepare, at sizes on both sides of the
    # single-workgroup plan; nothing launched
    for nV, nE in ((100, 200), (37000, 73000)):
        assert lib.dmpnn_prepare_with_batch(4096, 4096, 4096, nV, nE, None, 1 << 30, None) == -1       # null plan
        assert lib.dmpnn_prepare_with_batch(4096, 4096, 4096, nV, nE, 4096, 64, None) == -3            # plan buffer too small
        assert lib.dmpnn_prepare_with_batch(4096, 4096, 4096, nV, nE, 4100, 1 << 30, None) == -1       # plan not 16-byte aligned
        assert lib.dmpnn_prepare_with_batch(None, None, 4096, nV, nE, 4096, 1 << 30, None) == -1       # null index arrays
    # pack_tiles (host): bad arguments
    import numpy as np

    ao = np.array([0, 3, 6], np.int32)
    eo = np.array([0, 4, 8], np.int32)
    tr, ta = np.zeros(8, np.int32), np.zeros(8, np.int32)
    assert lib.dmpnn_pack_tiles(ao.ctypes.data, eo.ctypes.data, 2, tr.ctypes.data, ta.ctypes.data, 8) == 1
    assert tr[:2].tolist() == [0, 8] and ta[:2].tolist() == [0, 6]
    assert lib.dmpnn_pack_tiles(ao.ctypes.data, eo.ctypes.data, 2, tr.ctypes.data, ta.ctypes.data, 1) == -2   # cap too small
    assert lib.dmpnn_pack_tiles(None, None, 2, tr.ctypes.data, ta.ctypes.data, 8) == -2
    assert lib.dmpnn_pack_tiles(ao.ctypes.data, eo.ctypes.data, -1, tr.ctypes.data, ta.ctypes.data, 8) == -2
    # collate: sizes
    assert lib.dmpnn_collate(None, None, 0, None, None, None, 0, 0, None, None, None, None) == 0     # empty batch: nothing to do
    assert lib.dmpnn_collate(None, None, 0, None, None, None, 5, 0, None, None, 4096, None) == -1    # atoms without molecules
    assert lib.dmpnn_collate(4096, 4096, 1, None, None, None, 5, 4, None, None, 4096, None) == -1    # null edge arrays


def test_head_workspace_covers_the_four_launch_form_and_the_bounds_table_its_done_flags():
    """Host-only size rules (no GPU): the aggregation's bounds table is first | end | flag + 3 | done (round 5: the forward tile kernel
    marks the molecules whose aggregate it wrote), and ``dmpnn_head_ws_bytes`` grows by the hidden layer's split weight in both
    orientations, its row scales and the row blocks' partials exactly for the shapes the four-launch head takes (one hidden layer of at
    most 320 columns, at most 4 outputs, at most 1 024 molecules, not cross entropy)."""
    lib = _lib.load()
    assert lib.dmpnn_molagg_ws_bytes(0) == 16 and lib.dmpnn_molagg_ws_bytes(10) == (3 * 10 + 4) * 4

    def ws(n_mols, d, hidden, tasks, n_layers=2, loss=0):
        h = _lib.HeadArgs()
        h.n_atoms, h.n_mols, h.d_h = 9 * n_mols, n_mols, d
        h.n_layers = n_layers
        dims = [d] + [hidden] * (n_layers - 1) + [tasks]
        for i, v in enumerate(dims):
            h.dims[i] = v
        h.loss = loss
        h.n_classes = 2 if loss == _lib.LOSS["ce"] else 0
        return int(lib.dmpnn_head_ws_bytes(C.byref(h)))

    def extra(n_mols, n, k, t):   # al256 of: both split orientations, the row scales, the partials of ceil(n_mols / 16) row blocks
        al = lambda x: (x + 255) // 256 * 256
        return (al(-(-n // 16) * -(-k // 32) * 2048) + al(-(-k // 16) * -(-n // 32) * 2048) + al(4 * n) + al(4 * k)
                + al(-(-n_mols // 16) * (t * n + t + 2) * 4))

    base = ws(512, 300, 300, 8)                       # 8 outputs: the chain's layout
    for n_mols, d, hidden, t in ((512, 300, 300, 1), (77, 64, 128, 3), (1024, 300, 200, 4)):
        with_rows = ws(n_mols, d, hidden, t)
        # the same shape with the rule switched off by ONE property at a time: too many molecules / cross entropy
        assert with_rows - ws(n_mols, d, hidden, t, loss=_lib.LOSS["ce"]) == extra(n_mols, hidden, d, t), (n_mols, d, hidden, t)
    assert ws(1025, 300, 300, 1) < ws(1024, 300, 300, 1) + extra(1024, 300, 300, 1)   # beyond 1 024 molecules: the chain's workspace only
    assert ws(512, 300, 384, 1) == ws(512, 300, 384, 1, loss=_lib.LOSS["ce"])         # a hidden layer beyond 320 columns: the chain
    assert base > 0


def test_tile_waves_is_a_shape_rule(monkeypatch):
    """dmpnn_tile_waves (ABI 13): one tile per 512-thread workgroup (8 waves) where the launch has at most one tile per CU and the
    hidden size has 20 column tiles to split; the 4-wave form (two workgroups per CU) everywhere else."""
    lib = _lib.load()
    assert lib.dmpnn_tile_waves(4636, 9120, 300, 0) == 8       # the headline batch: ~230 tiles on 256 CUs
    assert lib.dmpnn_tile_waves(4636, 9120, 300, 230) == 8
    assert lib.dmpnn_tile_waves(40000, 80000, 300, 0) == 4     # 4 096 molecules: several tiles per CU
    assert lib.dmpnn_tile_waves(5200, 10034, 300, 0) == 4      # 576 molecules: 251 tiles by the estimate, 259 in fact — no second round of 8-wave tiles
    assert lib.dmpnn_tile_waves(4636, 9120, 128, 0) == 4       # d_h <= 128: the 1- / 2-column-tile instantiations
    assert lib.dmpnn_tile_waves(4636, 9120, 512, 0) == 4       # d_h > 320: not a tile-kernel shape at all

