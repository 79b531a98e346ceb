"""ctypes binding of oracle/liboracle.so — the CPU restatement of the reference path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under feature_tracker_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "liboracle.so")

MODELS = {"basic": 0, "affine": 1, "lssd": 2}
METHODS = {"inverse": 0, "direct": 1, "fast": 2}

NOT_TRACKED, TRACKED, LARGE_RESIDUAL, OUTSIDE, NUMERIC_ERROR = range(5)


class _Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class _DirectOptions(C.Structure):
    _fields_ = [
        ("max_track_points", C.c_uint32), ("max_iteration", C.c_uint32), ("half_rows", C.c_int32), ("half_cols", C.c_int32),
        ("max_converge_step", C.c_float), ("max_converge_residual", C.c_float), ("method", C.c_int32),
    ]


class _Options(C.Structure):
    _fields_ = [
        ("max_track_points", C.c_uint32), ("max_iteration", C.c_uint32), ("max_tolerance_large_step", C.c_uint32),
        ("half_rows", C.c_int32), ("half_cols", C.c_int32), ("max_converge_step", C.c_float), ("method", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc -O3 -ffp-contract=off)."""
    srcs = [os.path.join(_ORACLE_DIR, f) for f in os.listdir(_ORACLE_DIR) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _ORACLE_DIR, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_get_pixel_value_nocheck.restype = C.c_float
        _lib.orc_create_pyramid.restype = C.c_int64
        _lib.orc_extract_extend_patch.restype = C.c_uint32
        for name in ("orc_eigen_dot", "orc_eigen_norm", "orc_cosine_distance"):
            getattr(_lib, name).restype = C.c_float
    return _lib


def _images(levels):
    arr = (_Image * len(levels))()
    keep = []
    for i, img in enumerate(levels):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        keep.append(img)
        arr[i].data = img.ctypes.data
        arr[i].rows = img.shape[0]
        arr[i].cols = img.shape[1]
    return arr, keep


def make_options(method="fast", half=6, half_cols=None, max_points=500, max_iteration=15, max_large_step=3, converge=4e-2):
    o = _Options()
    o.max_track_points = max_points
    o.max_iteration = max_iteration
    o.max_tolerance_large_step = max_large_step
    o.half_rows = half
    o.half_cols = half if half_cols is None else half_cols
    o.max_converge_step = converge
    o.method = METHODS[method] if isinstance(method, str) else int(method)
    return o


def _prep_track(ref_uv, cur_uv, status):
    ref_uv = np.ascontiguousarray(ref_uv, dtype=np.float32).reshape(-1, 2)
    n = ref_uv.shape[0]
    # OpticalFlow::TrackFeatures input normalisation, optical_flow.cpp:12-19
    if cur_uv is None or np.asarray(cur_uv).reshape(-1, 2).shape[0] != n:
        cur = ref_uv.copy()
    else:
        cur = np.array(cur_uv, dtype=np.float32).reshape(-1, 2).copy()
    if status is None or np.asarray(status).size != n:
        st = np.zeros(n, dtype=np.uint8)
    else:
        st = np.array(status, dtype=np.uint8).copy()
    return ref_uv, cur, st, n


def klt_track_pyramid(model, ref_levels, cur_levels, ref_uv, cur_uv=None, status=None, prior=None, consider_luminance=False, **opt):
    """Returns (ok, cur_uv, status, iters)."""
    ref_uv, cur, st, n = _prep_track(ref_uv, cur_uv, status)
    if n == 0 or len(ref_levels) != len(cur_levels):
        return False, cur, st, np.zeros(n, np.uint32)
    ra, k1 = _images(ref_levels)
    ca, k2 = _images(cur_levels)
    o = make_options(**opt)
    pr = np.ascontiguousarray(np.eye(2) if prior is None else prior, dtype=np.float32).reshape(4)
    iters = np.zeros(n, dtype=np.uint32)
    ok = lib().orc_klt_track_pyramid(
        MODELS[model], C.byref(o), ra, ca, len(ref_levels), ref_uv.ctypes.data_as(C.c_void_p), cur.ctypes.data_as(C.c_void_p),
        st.ctypes.data_as(C.c_void_p), n, pr.ctypes.data_as(C.c_void_p), int(bool(consider_luminance)), iters.ctypes.data_as(C.c_void_p))
    return bool(ok), cur, st, iters


def klt_track_single(model, ref_image, cur_image, ref_uv, cur_uv=None, status=None, prior=None, consider_luminance=False, **opt):
    ref_uv, cur, st, n = _prep_track(ref_uv, cur_uv, status)
    if n == 0:
        return False, cur, st, np.zeros(n, np.uint32)
    ra, k1 = _images([ref_image])
    ca, k2 = _images([cur_image])
    o = make_options(**opt)
    pr = np.ascontiguousarray(np.eye(2) if prior is None else prior, dtype=np.float32).reshape(4)
    iters = np.zeros(n, dtype=np.uint32)
    ok = lib().orc_klt_track_single(
        MODELS[model], C.byref(o), ra, ca, ref_uv.ctypes.data_as(C.c_void_p), cur.ctypes.data_as(C.c_void_p),
        st.ctypes.data_as(C.c_void_p), n, pr.ctypes.data_as(C.c_void_p), int(bool(consider_luminance)), iters.ctypes.data_as(C.c_void_p))
    return bool(ok), cur, st, iters


def create_pyramid(image, levels):
    image = np.ascontiguousarray(image, dtype=np.uint8)
    rows, cols = image.shape
    buf = np.zeros(rows * cols, dtype=np.uint8)
    lib().orc_create_pyramid(image.ctypes.data_as(C.c_void_p), rows, cols, levels, buf.ctypes.data_as(C.c_void_p))
    out, off = [image], 0
    for _ in range(1, levels):
        rows, cols = rows // 2, cols // 2
        out.append(buf[off:off + rows * cols].reshape(rows, cols).copy())
        off += rows * cols
    return out


def get_pixel_value(image, row, col):
    arr, keep = _images([image])
    v = C.c_float(0.0)
    ok = lib().orc_get_pixel_value(arr, C.c_float(row), C.c_float(col), C.byref(v))
    return bool(ok), v.value


def ldlt_solve(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    n = b.shape[0]
    x = np.zeros(n, dtype=np.float32)
    lib().orc_ldlt_solve(n, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
    return x


def extract_extend_patch(image, u, v, ex_rows, ex_cols):
    arr, keep = _images([image])
    patch = np.zeros(ex_rows * ex_cols, dtype=np.float32)
    valid = np.zeros(ex_rows * ex_cols, dtype=np.uint8)
    cnt = lib().orc_extract_extend_patch(arr, C.c_float(u), C.c_float(v), ex_rows, ex_cols, patch.ctypes.data_as(C.c_void_p),
                                         valid.ctypes.data_as(C.c_void_p))
    return int(cnt), patch.reshape(ex_rows, ex_cols), valid.reshape(ex_rows, ex_cols)


def _prep_index(index_pairs, n_ref):
    # descriptor_matcher.h:60-62 — reset to -1 only when the size differs
    if index_pairs is None or np.asarray(index_pairs).size != n_ref:
        return np.full(n_ref, -1, dtype=np.int32)
    return np.array(index_pairs, dtype=np.int32).copy()


def force_match(ref_bits, cur_bits, max_distance, index_pairs=None):
    ref_bits = np.ascontiguousarray(ref_bits, dtype=np.uint8)
    cur_bits = np.ascontiguousarray(cur_bits, dtype=np.uint8)
    n_ref = ref_bits.shape[0]
    n_cur = cur_bits.shape[0]
    n_bits = ref_bits.shape[1] if ref_bits.ndim == 2 else 0
    idx = _prep_index(index_pairs, n_ref)
    if n_cur == 0:
        return False, idx if index_pairs is not None else np.zeros(0, np.int32)
    ok = lib().orc_force_match_bits(ref_bits.ctypes.data_as(C.c_void_p), n_ref, cur_bits.ctypes.data_as(C.c_void_p), n_cur, n_bits,
                                    C.c_float(max_distance), idx.ctypes.data_as(C.c_void_p))
    return bool(ok), idx


def nearby_match(ref_bits, cur_bits, pred_uv, cur_uv, max_distance, max_col=40, max_row=40, index_pairs=None):
    ref_bits = np.ascontiguousarray(ref_bits, dtype=np.uint8)
    cur_bits = np.ascontiguousarray(cur_bits, dtype=np.uint8)
    pred_uv = np.ascontiguousarray(pred_uv, dtype=np.float32).reshape(-1, 2)
    cur_uv = np.ascontiguousarray(cur_uv, dtype=np.float32).reshape(-1, 2)
    n_ref, n_cur = ref_bits.shape[0], cur_bits.shape[0]
    n_bits = ref_bits.shape[1] if ref_bits.ndim == 2 else 0
    # descriptor_matcher.h:94-96
    if n_cur == 0 or n_ref != pred_uv.shape[0] or n_cur != cur_uv.shape[0]:
        return False, (np.zeros(0, np.int32) if index_pairs is None else np.array(index_pairs, np.int32))
    idx = _prep_index(index_pairs, n_ref)
    ok = lib().orc_nearby_match_bits(ref_bits.ctypes.data_as(C.c_void_p), n_ref, cur_bits.ctypes.data_as(C.c_void_p), n_cur, n_bits,
                                     C.c_float(max_distance), pred_uv.ctypes.data_as(C.c_void_p), cur_uv.ctypes.data_as(C.c_void_p),
                                     int(max_col), int(max_row), idx.ctypes.data_as(C.c_void_p))
    return bool(ok), idx


def eigen_dot(x, y):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    return np.float32(lib().orc_eigen_dot(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size))


def cosine_distance(ref, cur):
    """SuperpointMatcher / DiskMatcher::ComputeDistance of one pair."""
    ref = np.ascontiguousarray(ref, dtype=np.float32)
    cur = np.ascontiguousarray(cur, dtype=np.float32)
    return np.float32(lib().orc_cosine_distance(ref.ctypes.data_as(C.c_void_p), cur.ctypes.data_as(C.c_void_p), ref.size))


def match_float(ref_desc, cur_desc, max_distance, pred_uv=None, cur_uv=None, max_col=40, max_row=40, index_pairs=None):
    """ForceMatch (pred_uv is None) / NearbyMatch over float descriptors (n, dim) with the cosine distance."""
    ref_desc = np.ascontiguousarray(ref_desc, dtype=np.float32)
    cur_desc = np.ascontiguousarray(cur_desc, dtype=np.float32)
    n_ref, n_cur = ref_desc.shape[0], cur_desc.shape[0]
    dim = ref_desc.shape[1] if ref_desc.ndim == 2 else 0
    if n_cur == 0:
        return False, (np.zeros(0, np.int32) if index_pairs is None else np.array(index_pairs, np.int32))
    if pred_uv is None:
        idx = _prep_index(index_pairs, n_ref)
        ok = lib().orc_force_match_float(ref_desc.ctypes.data_as(C.c_void_p), n_ref, cur_desc.ctypes.data_as(C.c_void_p), n_cur, dim,
                                         C.c_float(max_distance), idx.ctypes.data_as(C.c_void_p))
        return bool(ok), idx
    pred_uv = np.ascontiguousarray(pred_uv, dtype=np.float32).reshape(-1, 2)
    cur_uv = np.ascontiguousarray(cur_uv, dtype=np.float32).reshape(-1, 2)
    if n_ref != pred_uv.shape[0] or n_cur != cur_uv.shape[0]:
        return False, (np.zeros(0, np.int32) if index_pairs is None else np.array(index_pairs, np.int32))
    idx = _prep_index(index_pairs, n_ref)
    ok = lib().orc_nearby_match_float(ref_desc.ctypes.data_as(C.c_void_p), n_ref, cur_desc.ctypes.data_as(C.c_void_p), n_cur, dim,
                                      C.c_float(max_distance), pred_uv.ctypes.data_as(C.c_void_p), cur_uv.ctypes.data_as(C.c_void_p),
                                      int(max_col), int(max_row), idx.ctypes.data_as(C.c_void_p))
    return bool(ok), idx


def fill_matched_pixels(index_pairs, cur_uv, status=None):
    index_pairs = np.ascontiguousarray(index_pairs, dtype=np.int32)
    cur_uv = np.ascontiguousarray(cur_uv, dtype=np.float32).reshape(-1, 2)
    n_ref = index_pairs.shape[0]
    if status is None or np.asarray(status).size != n_ref:
        st = np.zeros(n_ref, dtype=np.uint8)
    else:
        st = np.array(status, dtype=np.uint8).copy()
    matched = np.zeros((n_ref, 2), dtype=np.float32)
    lib().orc_fill_matched_pixels(index_pairs.ctypes.data_as(C.c_void_p), n_ref, cur_uv.ctypes.data_as(C.c_void_p), cur_uv.shape[0],
                                  matched.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p))
    return matched, st


def direct_track(ref_levels, cur_levels, K, p_c_in_ref, ref_uv, cur_uv=None, q_rc=(1, 0, 0, 0), p_rc=(0, 0, 0), status=None, method="direct",
                 half=6, half_cols=None, max_points=500, max_iteration=15, converge=1e-6):
    """DirectMethod::TrackFeatures, camera-frame overload (direct_method_tracker.cpp:35-86).
    Returns (ok, cur_uv, q_rc (w, x, y, z), p_rc, status, iterations)."""
    ref_uv = np.ascontiguousarray(ref_uv, dtype=np.float32).reshape(-1, 2)
    n = ref_uv.shape[0]
    pts = np.ascontiguousarray(p_c_in_ref, dtype=np.float32).reshape(-1, 3)
    # :42-44 — sizes differ: no prediction
    cur = ref_uv.copy() if (cur_uv is None or np.asarray(cur_uv).reshape(-1, 2).shape[0] != n) else np.array(cur_uv, np.float32).reshape(-1, 2).copy()
    q = np.array(q_rc, dtype=np.float32).copy()
    p = np.array(p_rc, dtype=np.float32).copy()
    valid = status is not None and np.asarray(status).size == n
    st = np.array(status, dtype=np.uint8).copy() if valid else np.zeros(n, np.uint8)
    if n == 0 or len(ref_levels) != len(cur_levels):
        return False, cur, q, p, st, 0
    ra, k1 = _images(ref_levels)
    ca, k2 = _images(cur_levels)
    o = _DirectOptions()
    o.max_track_points, o.max_iteration, o.half_rows, o.half_cols = max_points, max_iteration, half, half if half_cols is None else half_cols
    o.max_converge_step, o.max_converge_residual, o.method = converge, 2.0, METHODS[method] if isinstance(method, str) else int(method)
    Kf = np.ascontiguousarray(K, dtype=np.float32)
    it = C.c_uint32(0)
    ok = lib().orc_direct_track(C.byref(o), ra, ca, len(ref_levels), Kf.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p),
                                ref_uv.ctypes.data_as(C.c_void_p), cur.ctypes.data_as(C.c_void_p), n, q.ctypes.data_as(C.c_void_p),
                                p.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), int(valid), C.byref(it))
    return bool(ok), cur, q, p, st, int(it.value)


def quat_mul(a, b):
    a, b, out = np.asarray(a, np.float32).copy(), np.asarray(b, np.float32).copy(), np.zeros(4, np.float32)
    lib().orc_quat_mul(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def quat_rotate(q, v):
    q, v, out = np.asarray(q, np.float32).copy(), np.asarray(v, np.float32).copy(), np.zeros(3, np.float32)
    lib().orc_quat_rotate(q.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def quat_inverse(q):
    q, out = np.asarray(q, np.float32).copy(), np.zeros(4, np.float32)
    lib().orc_quat_inverse(q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def brief_compute(image, uv, n_bits=256, half=8):
    """Per-bit BRIEF descriptors (n, n_bits) uint8 of the repo's normative definition (oracle_brief.c)."""
    arr, keep = _images([image])
    uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    bits = np.zeros((uv.shape[0], n_bits), dtype=np.uint8)
    ok = lib().orc_brief_compute(arr, uv.ctypes.data_as(C.c_void_p), uv.shape[0], int(n_bits), int(half), bits.ctypes.data_as(C.c_void_p))
    return bool(ok), bits


def harris_response(image):
    arr, keep = _images([image])
    out = np.zeros(keep[0].shape, dtype=np.float32)
    lib().orc_harris_response(arr, out.ctypes.data_as(C.c_void_p))
    return out


def harris_detect(image, max_count=300, min_distance=25, min_response=40.0):
    arr, keep = _images([image])
    uv = np.zeros((max(1, max_count), 2), dtype=np.float32)
    n = lib().orc_harris_detect(arr, int(max_count), int(min_distance), C.c_float(min_response), uv.ctypes.data_as(C.c_void_p))
    return uv[:n].copy()
