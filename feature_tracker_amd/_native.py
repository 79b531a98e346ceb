"""ctypes binding of libftk_hip.so (C ABI in include/ftk.h).

The library is built in-tree by ``feature_tracker_amd/csrc/Makefile`` (hipcc, gfx950).  Loading
fails loudly when it is missing: there is no Python / CPU fallback for any compute entry point.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libftk_hip.so")

FTK_OK = 0
FTK_MAX_LEVELS = 12
ERROR_NAMES = {0: "FTK_OK", -1: "FTK_E_INVALID_ARGUMENT", -2: "FTK_E_NO_DEVICE", -3: "FTK_E_HIP", -4: "FTK_E_UNSUPPORTED",
               -5: "FTK_E_OUT_OF_MEMORY"}

MODELS = {"basic": 0, "affine": 1, "lssd": 2}
METHODS = {"inverse": 0, "direct": 1, "fast": 2, "sse": 3, "neon": 4}

# every symbol include/ftk.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "ftk_abi_version", "ftk_build_info", "ftk_device_count", "ftk_context_create", "ftk_context_destroy", "ftk_last_error", "ftk_synchronize", "ftk_warmup", "ftk_set_reduction_mode", "ftk_context_refresh_env", "ftk_pyramid_update",
    "ftk_default_klt_options", "ftk_pyramid_upload", "ftk_pyramid_wrap_device", "ftk_pyramid_build", "ftk_pyramid_levels",
    "ftk_pyramid_level", "ftk_pyramid_download_level", "ftk_pyramid_destroy", "ftk_klt_track", "ftk_klt_track_device",
    "ftk_extract_extend_patch", "ftk_hamming_match", "ftk_hamming_match_device", "ftk_cosine_match", "ftk_cosine_match_device", "ftk_ldlt6_solve", "ftk_default_direct_options", "ftk_direct_track", "ftk_direct_track_batch_device", "ftk_fill_matched_pixels",
    "ftk_brief_compute", "ftk_brief_compute_device", "ftk_harris_detect", "ftk_harris_response",
    "ftk_shard_bounds", "ftk_klt_shard_bytes", "ftk_comm_unique_id", "ftk_comm_create", "ftk_comm_destroy", "ftk_comm_rank", "ftk_comm_world",
    "ftk_klt_track_sharded_device", "ftk_klt_track_sharded", "ftk_klt_track_shard_device", "ftk_klt_unpack_shards_device", "ftk_hamming_match_sharded_device",
]
UNIQUE_ID_BYTES = 128


class FtkError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class KltOptions(C.Structure):
    """OpticalFlowOptions (optical_flow.h:20-28)."""
    _fields_ = [
        ("max_track_points", C.c_uint32), ("max_iteration", C.c_uint32), ("max_tolerance_large_step", C.c_uint32),
        ("half_rows", C.c_int32), ("half_cols", C.c_int32), ("max_converge_step", C.c_float), ("method", C.c_int32),
    ]


class DirectOptions(C.Structure):
    """DirectMethodOptions (direct_method_tracker.h:20-28)."""
    _fields_ = [
        ("max_track_points", C.c_uint32), ("max_iteration", C.c_uint32), ("half_rows", C.c_int32), ("half_cols", C.c_int32),
        ("max_converge_step", C.c_float), ("max_converge_residual", C.c_float), ("method", C.c_int32),
    ]


class DirectProblem(C.Structure):
    """ftk_direct_problem: one pose problem of a batched launch (device pointers)."""
    _fields_ = [
        ("ref", C.c_void_p), ("cur", C.c_void_p), ("K", C.c_float * 4), ("d_p_c_in_ref", C.c_void_p), ("d_ref_uv", C.c_void_p),
        ("d_cur_uv", C.c_void_p), ("n", C.c_int32), ("d_pose", C.c_void_p), ("d_status", C.c_void_p), ("status_valid", C.c_int32),
        ("d_iterations", C.c_void_p),
    ]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libftk_hip.so for gfx950 with the committed Makefile (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith((".cpp", ".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_PKG_DIR), "include", "ftk.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        cmd = ["make", "-C", CSRC_DIR, "-j4"] + (["-B"] if force else []) + ["libftk_hip.so"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("building libftk_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
        if verbose:
            print(res.stdout)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own HIP / HSA runtime.  Two runtimes in one process do not share
    # the device, so when torch is installed it must be loaded FIRST: libftk_hip.so then binds to
    # the same libamdhip64.so.7 (same SONAME) and device pointers / streams are interchangeable.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = os.environ.get("FTK_LIB_PATH", LIB_PATH)  # e.g. a diagnostic (-DFTK_STAMPS) build
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension must be built first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C feature_tracker_amd/csrc). "
            "feature_tracker_amd has no CPU fallback.")
    l = C.CDLL(path)
    vp, i32, u32p = C.c_void_p, C.c_int32, C.POINTER(C.c_uint32)
    l.ftk_abi_version.restype = C.c_int
    l.ftk_build_info.restype = C.c_char_p
    l.ftk_device_count.restype = C.c_int
    l.ftk_context_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    l.ftk_context_destroy.argtypes = [vp]
    l.ftk_context_destroy.restype = None
    l.ftk_last_error.argtypes = [vp]
    l.ftk_last_error.restype = C.c_char_p
    l.ftk_synchronize.argtypes = [vp]
    l.ftk_warmup.argtypes = [vp, C.c_uint]
    l.ftk_set_reduction_mode.argtypes = [vp, C.c_int]
    l.ftk_context_refresh_env.argtypes = [vp]
    l.ftk_pyramid_update.argtypes = [vp, vp, vp, C.c_int]
    l.ftk_default_klt_options.argtypes = [C.POINTER(KltOptions)]
    l.ftk_default_klt_options.restype = None
    l.ftk_pyramid_upload.argtypes = [vp, C.POINTER(Image), i32, C.POINTER(vp)]
    l.ftk_pyramid_wrap_device.argtypes = [vp, C.POINTER(Image), i32, C.POINTER(vp)]
    l.ftk_pyramid_build.argtypes = [vp, vp, i32, i32, i32, C.c_int, C.POINTER(vp)]
    l.ftk_pyramid_levels.argtypes = [vp]
    l.ftk_pyramid_level.argtypes = [vp, i32, C.POINTER(Image)]
    l.ftk_pyramid_download_level.argtypes = [vp, vp, i32, vp]
    l.ftk_pyramid_destroy.argtypes = [vp]
    l.ftk_pyramid_destroy.restype = None
    l.ftk_klt_track.argtypes = [vp, C.c_int, C.POINTER(KltOptions), vp, vp, vp, vp, vp, i32, vp, C.c_int, C.c_int, vp]
    l.ftk_klt_track_device.argtypes = [vp, C.c_int, C.POINTER(KltOptions), vp, vp, vp, vp, vp, vp, vp, i32, vp, C.c_int, C.c_int, vp]
    l.ftk_extract_extend_patch.argtypes = [vp, vp, i32, C.c_float, C.c_float, i32, i32, vp, vp, u32p]
    l.ftk_hamming_match.argtypes = [vp, vp, i32, vp, i32, i32, i32, C.c_float, vp, vp, i32, i32, vp, C.POINTER(C.c_int)]
    l.ftk_hamming_match_device.argtypes = [vp, vp, i32, vp, i32, i32, i32, C.c_float, vp, vp, i32, i32, vp, vp]
    l.ftk_cosine_match.argtypes = [vp, vp, i32, vp, i32, i32, C.c_float, vp, vp, i32, i32, vp, C.POINTER(C.c_int)]
    l.ftk_cosine_match_device.argtypes = [vp, vp, i32, vp, i32, i32, C.c_float, vp, vp, i32, i32, vp]
    l.ftk_ldlt6_solve.argtypes = [vp, vp, vp, vp, i32]
    l.ftk_default_direct_options.argtypes = [C.POINTER(DirectOptions)]
    l.ftk_default_direct_options.restype = None
    l.ftk_direct_track.argtypes = [vp, C.POINTER(DirectOptions), vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, C.c_int, u32p]
    l.ftk_direct_track_batch_device.argtypes = [vp, C.POINTER(DirectOptions), C.POINTER(DirectProblem), i32]
    l.ftk_fill_matched_pixels.argtypes = [vp, i32, vp, i32, vp, vp]
    l.ftk_brief_compute.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp]
    l.ftk_brief_compute_device.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp]
    l.ftk_harris_detect.argtypes = [vp, vp, i32, i32, i32, C.c_float, vp, C.POINTER(C.c_int32)]
    l.ftk_harris_response.argtypes = [vp, vp, i32, vp]
    l.ftk_shard_bounds.argtypes = [i32, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    l.ftk_shard_bounds.restype = None
    l.ftk_klt_shard_bytes.argtypes = [i32, i32]
    l.ftk_klt_shard_bytes.restype = C.c_size_t
    l.ftk_comm_unique_id.argtypes = [vp]
    l.ftk_comm_create.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    l.ftk_comm_destroy.argtypes = [vp]
    l.ftk_comm_destroy.restype = None
    l.ftk_comm_rank.argtypes = [vp]
    l.ftk_comm_world.argtypes = [vp]
    l.ftk_klt_track_sharded_device.argtypes = [vp, vp, C.c_int, C.POINTER(KltOptions), vp, vp, vp, vp, vp, vp, vp, i32, vp, C.c_int, C.c_int, vp]
    l.ftk_klt_track_sharded.argtypes = [vp, vp, C.c_int, C.POINTER(KltOptions), vp, vp, vp, vp, vp, i32, vp, C.c_int, C.c_int, vp]
    l.ftk_klt_track_shard_device.argtypes = [vp, i32, i32, C.c_int, C.POINTER(KltOptions), vp, vp, vp, vp, vp, i32, vp, C.c_int, C.c_int, vp, vp]
    l.ftk_klt_unpack_shards_device.argtypes = [vp, vp, i32, i32, vp, vp]
    l.ftk_hamming_match_sharded_device.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, C.c_float, vp, vp, i32, i32, vp]
    _lib = l
    return l


def build_info() -> dict:
    """ftk_build_info() as a dict: source_hash, compiler, arch, mllvm (accepted internal LLVM options), mllvm_rejected."""
    text = lib().ftk_build_info().decode()
    return {k.strip(): v.strip() for k, v in (item.split("=", 1) for item in text.split(";") if "=" in item)}


def check(rc: int, ctx_handle=None) -> None:
    if rc != FTK_OK:
        msg = lib().ftk_last_error(ctx_handle)
        raise FtkError(rc, msg.decode() if msg else "")
