"""Build and load ``libdmpnn_gfx950.so`` (the C-ABI of include/dmpnn.h) through ctypes.

The library is compiled IN-TREE with ``hipcc --offload-arch=gfx950`` (cross-compiles without a GPU)
and travels with the source snapshot.  It must be loaded *after* ``import torch`` so the process
binds torch's bundled HIP runtime (same SONAME ``libamdhip64.so.7``) and device pointers of torch
tensors are valid in it.  There is no fallback: if the library is missing or cannot be loaded,
every engine entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(_ROOT, "include")
# DMPNN_LIB: an alternative build of the same sources (kernel experiments: scripts/build_variant.py), same ABI
LIB_PATH = os.environ.get("DMPNN_LIB") or os.path.join(_HERE, "libdmpnn_gfx950.so")
SOURCES = ["dmpnn_abi.hip", "dmpnn_prepare.hip", "dmpnn_segment.hip", "dmpnn_gemm.hip", "dmpnn_gemm_p1.hip",
           "dmpnn_gemm_p2.hip", "dmpnn_gemm_p3.hip", "dmpnn_gemm_p4.hip", "dmpnn_gemm_s1.hip", "dmpnn_mega.hip", "dmpnn_mega16.hip", "dmpnn_mega16_lp.hip", "dmpnn_mega16_bwd.hip", "dmpnn_rows16.hip", "dmpnn_step16.hip", "dmpnn_bstep16.hip", "dmpnn_backward.hip", "dmpnn_molagg.hip", "dmpnn_collate.hip", "dmpnn_tiles_large.hip", "dmpnn_optim.hip", "dmpnn_wgrad16.hip", "dmpnn_head.hip"]
HEADERS = ["dmpnn_common.hpp", "dmpnn_spill_impl.hpp", "dmpnn_gemm_impl.hpp", "dmpnn_mega_impl.hpp", "dmpnn_mega16_impl.hpp", "dmpnn_mega16_bwd_impl.hpp", "dmpnn_rows16_impl.hpp", "dmpnn_seg16.hpp", "dmpnn_step16_impl.hpp"]
ABI_VERSION = 15
PLAN_NOFFSETS = 15

# every symbol include/dmpnn.h declares; tests check the .so exports all of them
EXPORTS = [
    "dmpnn_version", "dmpnn_debug_timestamps", "dmpnn_last_error_string", "dmpnn_last_launch_count", "dmpnn_plan_bytes",
    "dmpnn_plan_layout", "dmpnn_prepare", "dmpnn_prepare_light", "dmpnn_prepare_tiles", "dmpnn_message_fwd", "dmpnn_aggregate_fwd",
    "dmpnn_linear_fwd", "dmpnn_linear16_wsplit_bytes", "dmpnn_linear16_ok", "dmpnn_linear16_fwd", "dmpnn_update_fwd", "dmpnn_forward", "dmpnn_forward_can_fuse", "dmpnn_forward_wsplit_bytes", "dmpnn_forward_keep_bits_bytes", "dmpnn_forward_spill_bytes", "dmpnn_backward_ws_bytes", "dmpnn_backward", "dmpnn_message_bwd",
    "dmpnn_aggregate_bwd", "dmpnn_linear_wgrad_ws_bytes", "dmpnn_linear_wgrad",
    "dmpnn_molagg_ws_bytes", "dmpnn_molagg_bounds", "dmpnn_molagg_fwd", "dmpnn_molagg_bwd", "dmpnn_gather_rows", "dmpnn_collate", "dmpnn_pack_tiles", "dmpnn_max_tiles",
    "dmpnn_prepare_tiles_from_table", "dmpnn_prepare_with_batch", "dmpnn_tile_plan_any_size", "dmpnn_split_row_floats", "dmpnn_forward_can_fuse16", "dmpnn_adam_step",
    "dmpnn_full_plan_keeps_tiles", "dmpnn_head_ws_bytes", "dmpnn_head", "dmpnn_train_step", "dmpnn_forward_tiles", "dmpnn_forward_route", "dmpnn_dropout_keep",
    "dmpnn_clip_grad", "dmpnn_clip_grad_ws_bytes", "dmpnn_train_route", "dmpnn_forward_h0_bytes", "dmpnn_tile_waves", "dmpnn_debug_lds_poison",
]

ACT = {"none": 0, "relu": 1, "leakyrelu": 2, "prelu": 3, "tanh": 4, "elu": 5}
F_UNDIRECTED = 1
F_FUSED = 2
F_MEGA = 4
F_KEEP = 8
F_SPLIT16 = 16
F_WSPLIT_READY = 32
F_LOADER_TILES = 64
F_STORE16 = 128
F_ATOM = 1024
F_TILE_PLAN = 2048
F_H0_RESIDUAL = 256
F_ROW_FINALIZE = 512
ROUTES = ("general", "general16", "fused", "fused16", "mega", "mega16")  # enum dmpnn_route
PLAN_NOMEGA_MASK = 15  # ... | no piece tiles (a molecule larger than a tile)
PLAN_NOFUSE_MASK = 7  # asymmetric | index out of range | in-degree > 24


class GemmArgs(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("N", C.c_int64), ("K1", C.c_int64), ("K2", C.c_int64),
        ("A1", C.c_void_p), ("lda1", C.c_int64), ("gather1", C.c_void_p), ("gather1_rows", C.c_int64),
        ("A2", C.c_void_p), ("lda2", C.c_int64),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("bias", C.c_void_p),
        ("Cadd", C.c_void_p), ("ldcadd", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("Zpre", C.c_void_p), ("ldz", C.c_int64),
        ("act", C.c_int), ("act_slope", C.c_float), ("act_slope_ptr", C.c_void_p),
    ]


class FwdArgs(C.Structure):
    _fields_ = [
        ("plan", C.c_void_p), ("n_atoms", C.c_int64), ("n_edges", C.c_int64),
        ("d_v", C.c_int64), ("d_e", C.c_int64), ("d_h", C.c_int64), ("d_vd", C.c_int64),
        ("depth", C.c_int32), ("flags", C.c_uint32),
        ("act", C.c_int32), ("act_slope", C.c_float), ("act_slope_ptr", C.c_void_p),
        ("V", C.c_void_p), ("ldv", C.c_int64),
        ("E", C.c_void_p), ("lde", C.c_int64),
        ("V_d", C.c_void_p), ("ldvd", C.c_int64),
        ("W_i", C.c_void_p), ("b_i", C.c_void_p),
        ("W_h", C.c_void_p), ("b_h", C.c_void_p),
        ("W_o", C.c_void_p), ("b_o", C.c_void_p),
        ("W_d", C.c_void_p), ("b_d", C.c_void_p),
        ("ldh", C.c_int64), ("H0", C.c_void_p), ("Hs", C.c_void_p), ("n_hslots", C.c_int32),
        ("Ms", C.c_void_p), ("n_mslots", C.c_int32),
        ("Mv", C.c_void_p), ("Hv", C.c_void_p),
        ("out", C.c_void_p), ("ldout", C.c_int64),
        ("wsplit", C.c_void_p), ("wsplit_bytes", C.c_size_t),
        ("edge_index", C.c_void_p), ("rev_edge_index", C.c_void_p),
        ("n_tiles_launch", C.c_int64),
        ("spill_ws", C.c_void_p), ("spill_bytes", C.c_size_t),
        ("msplit", C.c_void_p), ("msplit_bytes", C.c_size_t),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
        ("keep_bits", C.c_void_p), ("keep_bits_bytes", C.c_size_t),
        ("h0_bytes", C.c_size_t),
    ]


class BwdArgs(C.Structure):
    _fields_ = [
        ("f", FwdArgs), ("gout", C.c_void_p), ("ldgout", C.c_int64),
        ("gW_i", C.c_void_p), ("gb_i", C.c_void_p), ("gW_h", C.c_void_p), ("gb_h", C.c_void_p),
        ("gW_o", C.c_void_p), ("gb_o", C.c_void_p), ("gW_d", C.c_void_p), ("gb_d", C.c_void_p),
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("g_edge", C.c_void_p), ("ld_gedge", C.c_int64),
    ]


MAX_FFN_LAYERS = 8
LOSS = {"mse": 0, "mae": 1, "bce": 2, "ce": 3, "mve": 4, "evidential": 5, "quantile": 6}
STEP_FORWARD, STEP_BACKWARD, STEP_UPDATE = 1, 2, 4


class HeadArgs(C.Structure):
    _fields_ = [
        ("n_atoms", C.c_int64), ("n_mols", C.c_int64), ("d_h", C.c_int64),
        ("batch", C.c_void_p),
        ("agg_mode", C.c_int32), ("agg_norm", C.c_float),
        ("bn_weight", C.c_void_p), ("bn_bias", C.c_void_p), ("bn_running_mean", C.c_void_p), ("bn_running_var", C.c_void_p),
        ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("bn_training", C.c_int32),
        ("n_layers", C.c_int32), ("act", C.c_int32), ("act_slope", C.c_float),
        ("W", C.c_void_p * MAX_FFN_LAYERS), ("b", C.c_void_p * MAX_FFN_LAYERS), ("dims", C.c_int64 * (MAX_FFN_LAYERS + 1)),
        ("loss", C.c_int32),
        ("targets", C.c_void_p), ("weights", C.c_void_p), ("task_weights", C.c_void_p), ("lt_mask", C.c_void_p), ("gt_mask", C.c_void_p),
        ("preds", C.c_void_p), ("loss_out", C.c_void_p),
        ("gW", C.c_void_p * MAX_FFN_LAYERS), ("gb", C.c_void_p * MAX_FFN_LAYERS), ("g_bn_weight", C.c_void_p), ("g_bn_bias", C.c_void_p),
        ("gHv", C.c_void_p), ("ldg", C.c_int64),
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("bn_num_batches_tracked", C.c_void_p),
        ("n_classes", C.c_int32),
        ("evid_v_kl", C.c_float), ("evid_eps", C.c_float), ("quantile_alpha", C.c_float),
    ]


class TrainRouteInfo(C.Structure):
    _fields_ = [("plan_kind", C.c_int32), ("route", C.c_int32), ("keep_rows", C.c_int32), ("keep_bits", C.c_int32), ("lean", C.c_int32)]


class StepArgs(C.Structure):
    _fields_ = [
        ("edge_index", C.c_void_p), ("rev_edge_index", C.c_void_p), ("batch", C.c_void_p), ("plan_bytes", C.c_size_t), ("plan_ready", C.c_int32),
        ("stages", C.c_int32),
        ("bwd", BwdArgs), ("head", HeadArgs),
        ("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n_params", C.c_int64),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
        ("bias_corr1", C.c_float), ("sqrt_bias_corr2", C.c_float), ("grad_scale", C.c_float), ("dev_scalars", C.c_void_p),
        ("clip_val", C.c_float), ("clip_mode", C.c_int32), ("clip_ws", C.c_void_p),
    ]


def sources() -> list[str]:
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]


def _stale() -> bool:
    if os.environ.get("DMPNN_LIB"):
        return False  # (a variant build is used as it is)
    if not os.path.isfile(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "dmpnn.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force: bool = False, verbose: bool = False, defines=(), out: Optional[str] = None) -> str:
    """Compile every HIP source for gfx950 into ``chemprop_amd/libdmpnn_gfx950.so`` (``out`` / ``defines``: a variant build)."""
    lib_path = out or LIB_PATH
    if not force and not _stale() and out is None:
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.isfile(hipcc):
        raise RuntimeError("hipcc not found: cannot build libdmpnn_gfx950.so")
    import concurrent.futures as cf
    import tempfile

    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}", *[f"-D{d}" for d in defines]]
    with tempfile.TemporaryDirectory(prefix="dmpnn_build_") as tmp:
        def cc(src):
            obj = os.path.join(tmp, os.path.basename(src) + ".o")
            cmd = [hipcc, *flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src} ({r.returncode}):\n{r.stdout}\n{r.stderr}")
            return obj

        with cf.ThreadPoolExecutor(max_workers=8) as ex:
            objs = list(ex.map(cc, sources()))
        # (linked inside the scratch directory: the link step leaves per-object unbundling temporaries next to its output)
        linked = os.path.join(os.path.dirname(objs[0]), "libdmpnn.so")
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", linked]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc link failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")
        shutil.copyfile(linked, lib_path + ".tmp")
        os.chmod(lib_path + ".tmp", 0o755)
    os.replace(lib_path + ".tmp", lib_path)
    _clean_unbundle_temporaries(lib_path)
    return lib_path


def _clean_unbundle_temporaries(lib_path: str) -> int:
    """Delete ``<lib>.N.hipv4-amdgcn-amd-amdhsa--gfx950`` / ``<lib>.N.host-x86_64-unknown-linux-gnu-``: what
    ``clang-offload-bundler --unbundle`` / ``llvm-objdump --offloading`` leave next to the library they inspect (git-ignored, but they
    would travel with every snapshot to the GPU box: 7 MB at the end of round 4)."""
    import glob

    n = 0
    for f in glob.glob(glob.escape(lib_path) + ".*"):
        tail = f[len(lib_path) + 1:]
        if tail.split(".", 1)[0].isdigit() and ("hipv4-" in tail or "host-" in tail):
            try:
                os.remove(f)
                n += 1
            except OSError:
                pass
    return n


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the library (building it first when a compiler is present and it is stale)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (bind torch's HIP runtime first)

    if _stale() and (shutil.which("hipcc") or os.path.isfile("/opt/rocm/bin/hipcc")):
        build()
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the MI355X engine has no CPU / eager fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.dmpnn_version.restype = C.c_int
    lib.dmpnn_last_error_string.restype = C.c_char_p
    lib.dmpnn_last_launch_count.restype = C.c_int
    lib.dmpnn_plan_bytes.restype = C.c_size_t
    lib.dmpnn_plan_bytes.argtypes = [C.c_int64, C.c_int64]
    lib.dmpnn_plan_layout.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    lib.dmpnn_forward_can_fuse.argtypes = [C.POINTER(FwdArgs)]
    lib.dmpnn_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.dmpnn_prepare_light.argtypes = lib.dmpnn_prepare.argtypes
    lib.dmpnn_prepare_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.dmpnn_message_fwd.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_uint, C.c_void_p]
    lib.dmpnn_aggregate_fwd.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    lib.dmpnn_linear_fwd.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
    lib.dmpnn_update_fwd.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    lib.dmpnn_forward.argtypes = [C.POINTER(FwdArgs), C.c_void_p]
    lib.dmpnn_backward_ws_bytes.argtypes = [C.POINTER(FwdArgs)]
    lib.dmpnn_backward.argtypes = [C.POINTER(BwdArgs), C.c_void_p]
    lib.dmpnn_message_bwd.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_int64, C.c_void_p]
    lib.dmpnn_aggregate_bwd.argtypes = lib.dmpnn_message_bwd.argtypes
    lib.dmpnn_linear_wgrad_ws_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int]
    lib.dmpnn_linear_wgrad.argtypes = [C.POINTER(GemmArgs), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    size_t_fns = ("dmpnn_plan_bytes", "dmpnn_backward_ws_bytes", "dmpnn_linear_wgrad_ws_bytes", "dmpnn_forward_wsplit_bytes", "dmpnn_forward_spill_bytes",
                  "dmpnn_forward_keep_bits_bytes",
                  "dmpnn_molagg_ws_bytes", "dmpnn_linear16_wsplit_bytes", "dmpnn_head_ws_bytes", "dmpnn_clip_grad_ws_bytes", "dmpnn_forward_h0_bytes")
    lib.dmpnn_linear16_wsplit_bytes.argtypes = [C.c_int64, C.c_int64]
    lib.dmpnn_linear16_ok.argtypes = [C.POINTER(GemmArgs)]
    lib.dmpnn_linear16_fwd.argtypes = [C.POINTER(GemmArgs), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.dmpnn_gather_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_void_p]
    lib.dmpnn_pack_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
    lib.dmpnn_pack_tiles.restype = C.c_int64
    lib.dmpnn_max_tiles.argtypes = [C.c_int64, C.c_int64]
    lib.dmpnn_prepare_with_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.dmpnn_prepare_with_batch.restype = C.c_int
    lib.dmpnn_tile_plan_any_size.argtypes = [C.c_int64, C.c_int64]
    lib.dmpnn_max_tiles.restype = C.c_int64
    lib.dmpnn_prepare_tiles_from_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                                                   C.c_void_p]
    lib.dmpnn_collate.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dmpnn_molagg_ws_bytes.argtypes = [C.c_int64]
    lib.dmpnn_molagg_bounds.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.dmpnn_molagg_fwd.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_float,
                                     C.c_void_p, C.c_int64, C.c_void_p]
    lib.dmpnn_molagg_bwd.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int,
                                     C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    lib.dmpnn_forward_wsplit_bytes.argtypes = [C.POINTER(FwdArgs)]
    lib.dmpnn_forward_keep_bits_bytes.argtypes = [C.POINTER(FwdArgs)]
    lib.dmpnn_forward_spill_bytes.argtypes = [C.POINTER(FwdArgs)]
    lib.dmpnn_forward_h0_bytes.argtypes = [C.POINTER(FwdArgs)]
    lib.dmpnn_forward_can_fuse16.argtypes = [C.POINTER(FwdArgs)]
    lib.dmpnn_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    lib.dmpnn_train_route.argtypes = [C.POINTER(FwdArgs), C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(TrainRouteInfo)]
    lib.dmpnn_clip_grad.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]
    lib.dmpnn_dropout_keep.argtypes = [C.c_uint64, C.c_int32, C.c_int64, C.c_int64, C.c_float]
    lib.dmpnn_forward_route.argtypes = [C.POINTER(FwdArgs), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.dmpnn_tile_waves.restype = C.c_int
    lib.dmpnn_tile_waves.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64]
    lib.dmpnn_forward_tiles.argtypes = [C.POINTER(FwdArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_size_t, C.c_void_p]
    lib.dmpnn_full_plan_keeps_tiles.argtypes = [C.c_int64, C.c_int64]
    lib.dmpnn_head_ws_bytes.argtypes = [C.POINTER(HeadArgs)]
    lib.dmpnn_head.argtypes = [C.POINTER(HeadArgs), C.c_void_p, C.c_int64, C.c_void_p]
    lib.dmpnn_train_step.argtypes = [C.POINTER(StepArgs), C.c_void_p]
    lib.dmpnn_split_row_floats.argtypes = [C.c_int64]
    lib.dmpnn_split_row_floats.restype = C.c_int64
    lib.dmpnn_debug_timestamps.argtypes = [C.c_void_p]
    for name in size_t_fns:
        getattr(lib, name).restype = C.c_size_t
    for name in EXPORTS:
        if name != "dmpnn_last_error_string" and name not in size_t_fns and name not in ("dmpnn_pack_tiles", "dmpnn_max_tiles", "dmpnn_split_row_floats"):
            getattr(lib, name).restype = C.c_int
    if lib.dmpnn_version() != ABI_VERSION:
        raise RuntimeError(f"libdmpnn ABI version {lib.dmpnn_version()} != {ABI_VERSION} (stale build?)")
    _lib = lib
    return lib


# ---- environment options on the hot path -------------------------------------------------------
# os.environ.get encodes the key and decodes the value on every call (~1 us; a forward asks a dozen times);
# its backing dict (bytes -> bytes, kept in step by every os.environ assignment) is a plain lookup.
_ENV_DATA = getattr(os.environ, "_data", None)
_ENV_KEYS: dict = {}


def opt(name: str, default: str) -> str:
    """``os.environ.get(name, default)`` without the per-call encode / decode."""
    if _ENV_DATA is None or os.name != "posix":
        return os.environ.get(name, default)
    k = _ENV_KEYS.get(name)
    if k is None:
        k = _ENV_KEYS[name] = name.encode()
    v = _ENV_DATA.get(k)
    return default if v is None else v.decode()


class DmpnnError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dmpnn_last_error_string().decode(errors="replace")
        raise DmpnnError(f"{what} failed (code {rc}): {msg}")
