"""ctypes binding of the C ABI in include/tsnet_abi.h.

`load()` opens the in-tree HIP library (wacv23_tsnet_amd/lib/libtsnet_hip.so, built by
`__graft_entry__.build()` / `python -m wacv23_tsnet_amd.build`).  There is no fallback: if the
library is missing or fails to load, importing code gets a RuntimeError telling it to build.

torch must be imported before the library is opened: the library depends on libamdhip64.so.7,
and the dynamic loader then re-uses the HIP runtime torch already mapped (same SONAME), so
torch's allocations and our kernel launches share one runtime and one stream namespace.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtsnet_hip.so")
MAX_SOURCES = 8
TIMING_CLASSES = 9
TIMING_NAMES = ("conv", "stats", "elementwise", "flow", "warp", "pack", "upsample", "other", "conv_res")

# every symbol include/tsnet_abi.h declares (tests/test_abi.py checks header <-> library <-> this list)
ABI_SYMBOLS = (
    "tsnet_abi_version", "tsnet_create", "tsnet_load_weights", "tsnet_finalize", "tsnet_destroy",
    "tsnet_last_error", "tsnet_num_params", "tsnet_param_info", "tsnet_packed_weights",
    "tsnet_forward", "tsnet_set_source_divisors", "tsnet_set_sources", "tsnet_forward_target", "tsnet_train_extras", "tsnet_stage_ptr",
    "tsnet_forward_macs", "tsnet_timing_enable", "tsnet_timing_read",
    "tsnet_op_conv2d", "tsnet_op_conv2d_cat", "tsnet_op_head", "tsnet_op_instnorm_stats", "tsnet_op_norm_act", "tsnet_op_upsample2x",
    "tsnet_op_flow", "tsnet_op_flow_k", "tsnet_flow_plan", "tsnet_op_warp", "tsnet_op_warp_k", "tsnet_op_last_error", "tsnet_frame_stats", "tsnet_demo_postprocess", "tsnet_fit_face_curves", "tsnet_raster_face", "tsnet_vl2ch", "tsnet_fit_pose_curves", "tsnet_raster_pose", "tsnet_label_bbox", "tsnet_resize_pad", "tsnet_resize_label", "tsnet_bench_conv", "tsnet_debug_counters", "tsnet_linspace", "tsnet_coord_table",
)


class TsnetCfg(C.Structure):
    """struct tsnet_cfg (include/tsnet_abi.h)."""
    _fields_ = [
        ("label_nc", C.c_int), ("n_blocks", C.c_int), ("n_downsampling", C.c_int), ("n_source", C.c_int),
        ("ngf", C.c_int), ("enc_blocks", C.c_int), ("addcoords", C.c_int), ("pose_composite", C.c_int),
        ("pose_mean", C.c_float * 3), ("height", C.c_int), ("width", C.c_int), ("max_batch", C.c_int), ("operand_mode", C.c_int),
    ]


_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


def bind(lib: C.CDLL) -> C.CDLL:
    """Attach argtypes/restypes for every ABI entry point."""
    lib.tsnet_abi_version.restype = C.c_int
    lib.tsnet_create.argtypes = [C.POINTER(TsnetCfg), C.POINTER(_vp)]
    lib.tsnet_load_weights.argtypes = [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), C.c_int]
    lib.tsnet_finalize.argtypes = [_vp, _vp]
    lib.tsnet_destroy.argtypes = [_vp]
    lib.tsnet_destroy.restype = None
    lib.tsnet_last_error.argtypes = [_vp]
    lib.tsnet_last_error.restype = C.c_char_p
    lib.tsnet_num_params.argtypes = [_vp]
    lib.tsnet_param_info.argtypes = [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    lib.tsnet_packed_weights.argtypes = [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]
    pp = C.POINTER(_vp)
    lib.tsnet_forward.argtypes = [_vp, pp, pp, pp, _vp, _vp, _vp, _vp, C.c_int, _vp]
    lib.tsnet_set_source_divisors.argtypes = [_vp, _fp, C.c_int]
    lib.tsnet_set_sources.argtypes = [_vp, pp, pp, pp, C.c_int, _vp]
    lib.tsnet_forward_target.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp]
    lib.tsnet_train_extras.argtypes = [_vp, pp, _vp, C.c_int, _vp, _vp, _vp]
    lib.tsnet_stage_ptr.argtypes = [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(C.c_size_t)]
    lib.tsnet_forward_macs.argtypes = [_vp, C.c_int]
    lib.tsnet_forward_macs.restype = C.c_double
    lib.tsnet_timing_enable.argtypes = [_vp, C.c_int]
    lib.tsnet_timing_read.argtypes = [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    lib.tsnet_op_conv2d.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, _vp, _vp]
    lib.tsnet_op_conv2d_cat.argtypes = [_vp, _vp] + [C.c_int] * 6 + [_vp, _vp] + [C.c_int] * 5 + [C.c_float, C.c_int, _vp, _vp]
    lib.tsnet_op_head.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _fp, _vp, _vp]
    lib.tsnet_op_instnorm_stats.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]
    lib.tsnet_op_norm_act.argtypes = [_vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]
    lib.tsnet_op_upsample2x.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]
    lib.tsnet_resize_label.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp]
    lib.tsnet_op_flow.argtypes = [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]
    lib.tsnet_op_flow_k.argtypes = [_vp, _vp, _vp, _vp] + [C.c_int] * 7 + [_vp, C.c_int, C.c_int, _fp, _vp]
    lib.tsnet_flow_plan.argtypes = [C.c_int] * 4
    lib.tsnet_op_warp.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]
    if hasattr(lib, "tsnet_op_warp_k"):          # absent from an older build opened beside this one (tools/forward_ab.py --lib2)
        lib.tsnet_op_warp_k.argtypes = [_vp, _vp] + [C.c_int] * 5 + [_vp, C.c_int, _fp, _vp]
    lib.tsnet_op_last_error.restype = C.c_char_p
    lib.tsnet_frame_stats.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp]
    lib.tsnet_demo_postprocess.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _fp, _vp, _vp]
    lib.tsnet_fit_face_curves.argtypes = [_vp, C.c_int, _vp]
    lib.tsnet_fit_pose_curves.argtypes = [_vp, C.c_int, C.c_int, _vp]
    lib.tsnet_raster_face.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]
    lib.tsnet_vl2ch.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]
    lib.tsnet_raster_pose.argtypes = [_vp, _vp] + [C.c_int] * 8 + [_vp, _vp]
    lib.tsnet_label_bbox.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]
    lib.tsnet_resize_pad.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp] + [C.c_int] * 7 + [_vp, _vp]
    lib.tsnet_bench_conv.argtypes = [C.c_int] * 12 + [_fp, _vp]
    lib.tsnet_debug_counters.argtypes = [C.POINTER(C.c_int64), C.c_int]
    lib.tsnet_debug_counters.restype = None
    lib.tsnet_linspace.argtypes = [C.c_int, _fp]
    lib.tsnet_linspace.restype = None
    lib.tsnet_coord_table.argtypes = [C.c_int, C.c_int, _fp]
    lib.tsnet_coord_table.restype = None
    return lib


_cached = None


def load_tools() -> C.CDLL:
    """The tools flavour of the library (build.py --tools): product kernels + superseded generations + ablation variants.
    For tools/*.py on the GPU box only; the package itself never opens it."""
    from . import build as _b
    if not os.path.exists(_b.OUT_TOOLS):
        raise RuntimeError(f"{_b.OUT_TOOLS} is missing: run `python -m wacv23_tsnet_amd.build --tools`")
    return bind(C.CDLL(_b.OUT_TOOLS))


def load() -> C.CDLL:
    """Open the HIP library; raises (never falls back) if it is not built."""
    global _cached
    if _cached is not None:
        return _cached
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for the TS-Net forward path.")
    try:
        _cached = bind(C.CDLL(LIB_PATH))
    except OSError as e:  # pragma: no cover - depends on the machine
        raise RuntimeError(f"failed to load {LIB_PATH}: {e}") from e
    if _cached.tsnet_abi_version() != 5:
        raise RuntimeError("libtsnet_hip.so ABI version mismatch; rebuild")
    return _cached
