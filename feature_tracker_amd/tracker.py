"""Host-side mirror of the reference's tracker / matcher interface on top of the C ABI.

Class and method names follow the reference (namespace feature_tracker):
``OpticalFlowBasicKlt`` / ``OpticalFlowAffineKlt`` / ``OpticalFlowLssdKlt`` with ``options()``,
``TrackFeatures`` (pyramid and single-image overloads, optical_flow.h:38-42), ``predict_affine``,
``predict_R_cr``, ``consider_patch_luminance``; ``BriefMatcher`` = ``DescriptorMatcher<BriefType>``
with ``ForceMatch`` / ``NearbyMatch`` (descriptor_matcher.h:26-37).  Python cannot mutate its
arguments the way the C++ reference does, so the in/out vectors are returned instead:
``ok, cur_uv, status = klt.TrackFeatures(ref_pyr, cur_pyr, ref_uv, cur_uv, status)``.

All compute goes through libftk_hip.so; nothing here falls back to numpy.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional, Sequence

import numpy as np

from . import _native as N

NOT_TRACKED, TRACKED, LARGE_RESIDUAL, OUTSIDE, NUMERIC_ERROR = range(5)  # feature_tracker.h:8-14


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_LIVE_CONTEXTS = weakref.WeakSet()


def refresh_env_switches() -> None:
    """Every live context re-reads the FTK_* experiment switches (they are read once per context, not per call)."""
    for ctx in list(_LIVE_CONTEXTS):
        ctx.refresh_env()


class Context:
    """One HIP device + stream (ftk_context).  ``stream`` may be a raw hipStream_t handle, e.g.
    ``torch.cuda.Stream().cuda_stream``; by default the context owns a private stream."""

    def __init__(self, device: int = -1, stream: Optional[int] = None):
        self._h = C.c_void_p()
        rc = N.lib().ftk_context_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h))
        N.check(rc, None)
        _LIVE_CONTEXTS.add(self)

    def refresh_env(self):
        """ftk_context_refresh_env: the FTK_* experiment switches are read once per context; read them again (tests, sweeps)."""
        if self._h:
            N.check(N.lib().ftk_context_refresh_env(self._h), self._h)

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        N.check(N.lib().ftk_synchronize(self._h), self._h)

    def warmup(self, what: int = 31):
        """ftk_warmup: code-object load and first staging allocations now instead of inside the first real call
        (mask of FTK_WARM_*: 1 KLT, 2 Hamming, 4 cosine, 8 direct method, 16 BRIEF / Harris)."""
        N.check(N.lib().ftk_warmup(self._h, int(what)), self._h)

    def set_reduction(self, mode: str = "exact"):
        """ftk_set_reduction_mode: "exact" (default, the contract: sums in the reference's order, bit-identical results) or "tree"
        (throughput mode: same products, butterfly sums; reported next to the exact mode, never asserted)."""
        N.check(N.lib().ftk_set_reduction_mode(self._h, {"exact": 0, "tree": 1}[mode]), self._h)

    def close(self):
        if self._h:
            N.lib().ftk_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def _image_array(levels: Sequence[np.ndarray]):
    arr = (N.Image * len(levels))()
    keep = []
    for i, img in enumerate(levels):
        a = np.ascontiguousarray(img, dtype=np.uint8)
        if a.ndim != 2:
            raise ValueError("images must be 2-D uint8 arrays")
        keep.append(a)
        arr[i].data = a.ctypes.data
        arr[i].rows, arr[i].cols = a.shape
    return arr, keep


class ImagePyramid:
    """Device-resident image pyramid (level 0 = full resolution)."""

    def __init__(self, handle, ctx: Context, keepalive=None):
        self._h = handle
        self._ctx = ctx
        self._keep = keepalive

    @classmethod
    def from_host_levels(cls, levels: Sequence[np.ndarray], ctx: Optional[Context] = None) -> "ImagePyramid":
        """Upload an already built pyramid (what ImagePyramid::GetImageConst(i) returns per level)."""
        ctx = ctx or default_context()
        arr, keep = _image_array(levels)
        h = C.c_void_p()
        N.check(N.lib().ftk_pyramid_upload(ctx.handle, arr, len(levels), C.byref(h)), ctx.handle)
        return cls(h, ctx)

    @classmethod
    def from_device_levels(cls, ptrs_rows_cols: Sequence[tuple], ctx: Optional[Context] = None, keepalive=None) -> "ImagePyramid":
        """Borrow levels already in device memory: sequence of (device_ptr, rows, cols)."""
        ctx = ctx or default_context()
        arr = (N.Image * len(ptrs_rows_cols))()
        for i, (ptr, rows, cols) in enumerate(ptrs_rows_cols):
            arr[i].data, arr[i].rows, arr[i].cols = int(ptr), int(rows), int(cols)
        h = C.c_void_p()
        N.check(N.lib().ftk_pyramid_wrap_device(ctx.handle, arr, len(ptrs_rows_cols), C.byref(h)), ctx.handle)
        return cls(h, ctx, keepalive)

    @classmethod
    def build(cls, image: np.ndarray, levels: int, ctx: Optional[Context] = None) -> "ImagePyramid":
        """SetRawImage + CreateImagePyramid(levels) on the device (test_optical_flow.cpp:49-53,70-71)."""
        ctx = ctx or default_context()
        a = np.ascontiguousarray(image, dtype=np.uint8)
        h = C.c_void_p()
        N.check(N.lib().ftk_pyramid_build(ctx.handle, _ptr(a), a.shape[0], a.shape[1], int(levels), 0, C.byref(h)), ctx.handle)
        return cls(h, ctx)

    @classmethod
    def build_from_device(cls, device_ptr: int, rows: int, cols: int, levels: int, ctx: Optional[Context] = None, keepalive=None):
        ctx = ctx or default_context()
        h = C.c_void_p()
        N.check(N.lib().ftk_pyramid_build(ctx.handle, C.c_void_p(int(device_ptr)), rows, cols, int(levels), 1, C.byref(h)), ctx.handle)
        return cls(h, ctx, keepalive)

    def update(self, image, location: str = "host"):
        """The next frame into this pyramid (ftk_pyramid_update): level 0 overwritten, levels >= 1 rebuilt on the device, no
        allocation.  ``image``: a uint8 array of level 0's shape (location "host": synchronous), or a raw pointer (int) with
        location "device" (stream-ordered) / "host_async" (pinned host memory that stays valid until the stream has passed)."""
        loc = {"host": 0, "device": 1, "host_async": 2}[location]
        if isinstance(image, (int, np.integer)):
            ptr = C.c_void_p(int(image))
        else:
            a = np.ascontiguousarray(image, dtype=np.uint8)
            _, rows, cols = self.level_desc(0)
            if a.shape != (rows, cols):
                raise ValueError(f"image shape {a.shape} differs from level 0 ({rows}, {cols})")
            ptr = _ptr(a)
        N.check(N.lib().ftk_pyramid_update(self._ctx.handle, self._h, ptr, loc), self._ctx.handle)

    @property
    def handle(self):
        return self._h

    def level(self) -> int:
        return N.lib().ftk_pyramid_levels(self._h)

    def level_desc(self, i: int):
        im = N.Image()
        rc = N.lib().ftk_pyramid_level(self._h, i, C.byref(im))
        if rc != 0:
            raise IndexError(i)
        return im.data, im.rows, im.cols

    def download_level(self, i: int) -> np.ndarray:
        _, rows, cols = self.level_desc(i)
        out = np.empty((rows, cols), dtype=np.uint8)
        N.check(N.lib().ftk_pyramid_download_level(self._ctx.handle, self._h, i, _ptr(out)), self._ctx.handle)
        return out

    def close(self):
        if self._h:
            N.lib().ftk_pyramid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OpticalFlowOptions:
    """optical_flow.h:20-28 — same field names and defaults."""

    def __init__(self):
        self.kMaxTrackPointsNumber = 500
        self.kMaxIteration = 15
        self.kMaxToleranceLargeStep = 3
        self.kPatchRowHalfSize = 6
        self.kPatchColHalfSize = 6
        self.kMaxConvergeStep = 4e-2
        self.kMethod = "fast"  # OpticalFlowMethod::kFast

    def to_native(self) -> N.KltOptions:
        o = N.KltOptions()
        o.max_track_points = int(self.kMaxTrackPointsNumber)
        o.max_iteration = int(self.kMaxIteration)
        o.max_tolerance_large_step = int(self.kMaxToleranceLargeStep)
        o.half_rows = int(self.kPatchRowHalfSize)
        o.half_cols = int(self.kPatchColHalfSize)
        o.max_converge_step = float(self.kMaxConvergeStep)
        o.method = N.METHODS[self.kMethod] if isinstance(self.kMethod, str) else int(self.kMethod)
        return o


class OpticalFlow:
    """Abstract base (optical_flow.h:30-112): input normalisation + dispatch to the device tracker."""

    _model = None
    _name = "None"

    def __init__(self, ctx: Optional[Context] = None):
        self._ctx = ctx
        self._options = OpticalFlowOptions()
        self.last_iterations: Optional[np.ndarray] = None  # it(f) of the last call (bytes-moved accounting)

    def OpticalFlowMethodName(self) -> str:
        return self._name

    def options(self) -> OpticalFlowOptions:
        return self._options

    # hooks for the subclasses' prediction members
    def _prior(self) -> Optional[np.ndarray]:
        return None

    def _luminance(self) -> bool:
        return False

    def TrackFeatures(self, ref, cur, ref_pixel_uv, cur_pixel_uv=None, status=None):
        """Pyramid overload when ``ref``/``cur`` are ImagePyramid, single-image overload when they are
        2-D uint8 arrays (optical_flow.cpp:6-26 / :28-47).  Returns (ok, cur_pixel_uv, status)."""
        ref_uv = np.ascontiguousarray(ref_pixel_uv, dtype=np.float32).reshape(-1, 2)
        n = ref_uv.shape[0]
        cur_uv = None if cur_pixel_uv is None else np.asarray(cur_pixel_uv, dtype=np.float32).reshape(-1, 2)
        st = None if status is None else np.asarray(status, dtype=np.uint8).reshape(-1)
        single = not isinstance(ref, ImagePyramid)
        # RETURN_FALSE_IF(ref_pixel_uv.empty()) / level mismatch — before any normalisation (optical_flow.cpp:8-9)
        if n == 0 or (not single and cur.level() != ref.level()):
            return False, (np.zeros((0, 2), np.float32) if cur_uv is None else cur_uv), (np.zeros(0, np.uint8) if st is None else st)
        # size mismatch => "no prediction" / "not tracked yet" (optical_flow.cpp:12-19)
        cur_uv = ref_uv.copy() if (cur_uv is None or cur_uv.shape[0] != n) else np.ascontiguousarray(cur_uv).copy()
        st = np.zeros(n, dtype=np.uint8) if (st is None or st.shape[0] != n) else np.ascontiguousarray(st).copy()

        ctx = self._ctx or default_context()
        if single:
            ref_pyr = ImagePyramid.from_host_levels([ref], ctx)
            cur_pyr = ImagePyramid.from_host_levels([cur], ctx)
        else:
            ref_pyr, cur_pyr = ref, cur
        opt = self._options.to_native()
        prior = self._prior()
        prior_arr = None if prior is None else np.ascontiguousarray(prior, dtype=np.float32).reshape(4)
        iters = np.zeros(n, dtype=np.uint32)
        rc = N.lib().ftk_klt_track(ctx.handle, self._model, C.byref(opt), ref_pyr.handle, cur_pyr.handle, _ptr(ref_uv), _ptr(cur_uv), _ptr(st), n,
                                   _ptr(prior_arr), int(self._luminance()), int(single), _ptr(iters))
        N.check(rc, ctx.handle)
        self.last_iterations = iters
        return True, cur_uv, st

    def ExtractExtendPatchInReferenceImage(self, ref_image, ref_pixel_uv, ex_ref_patch_rows: int, ex_ref_patch_cols: int):
        """optical_flow.cpp:49-102.  Returns (valid_count, ex_patch (rows x cols float32), valid (rows x cols bool))."""
        ctx = self._ctx or default_context()
        pyr = ref_image if isinstance(ref_image, ImagePyramid) else ImagePyramid.from_host_levels([ref_image], ctx)
        patch = np.zeros(ex_ref_patch_rows * ex_ref_patch_cols, dtype=np.float32)
        valid = np.zeros(ex_ref_patch_rows * ex_ref_patch_cols, dtype=np.uint8)
        cnt = C.c_uint32(0)
        N.check(N.lib().ftk_extract_extend_patch(ctx.handle, pyr.handle, 0, float(ref_pixel_uv[0]), float(ref_pixel_uv[1]), ex_ref_patch_rows,
                                                 ex_ref_patch_cols, _ptr(patch), _ptr(valid), C.byref(cnt)), ctx.handle)
        return cnt.value, patch.reshape(ex_ref_patch_rows, ex_ref_patch_cols), valid.reshape(ex_ref_patch_rows, ex_ref_patch_cols).astype(bool)


class OpticalFlowBasicKlt(OpticalFlow):
    _model = N.MODELS["basic"]
    _name = "Basic-Klt"  # basic_klt.h:15


class OpticalFlowAffineKlt(OpticalFlow):
    _model = N.MODELS["affine"]
    _name = "Affine-Klt"  # affine_klt.h:15

    def __init__(self, ctx: Optional[Context] = None):
        super().__init__(ctx)
        self.predict_affine = np.eye(2, dtype=np.float32)  # affine_klt.h:50

    def _prior(self):
        return self.predict_affine


class OpticalFlowLssdKlt(OpticalFlow):
    _model = N.MODELS["lssd"]
    _name = "Lssd-Klt"  # lssd_klt.h:15

    def __init__(self, ctx: Optional[Context] = None):
        super().__init__(ctx)
        self.predict_R_cr = np.eye(2, dtype=np.float32)  # lssd_klt.h:53
        self.consider_patch_luminance = False  # lssd_klt.h:54

    def _prior(self):
        return self.predict_R_cr

    def _luminance(self):
        return bool(self.consider_patch_luminance)


def pack_brief(bits: np.ndarray) -> np.ndarray:
    """Per-bit BriefType container (n, n_bits) of 0/1 -> (n, ceil(n_bits / 32)) uint32 words."""
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    if bits.ndim != 2:
        bits = bits.reshape(bits.shape[0], -1) if bits.size else bits.reshape(0, 0)
    n, n_bits = bits.shape
    words = max(1, (n_bits + 31) // 32)
    padded = np.zeros((n, words * 32), dtype=np.uint8)
    padded[:, :n_bits] = bits != 0
    packed = np.packbits(padded.reshape(n, words, 32), axis=-1, bitorder="little")
    return np.ascontiguousarray(packed).view("<u4").reshape(n, words)


def unpack_brief(words: np.ndarray, n_bits: int) -> np.ndarray:
    """(n, n_words) uint32 -> per-bit (n, n_bits) uint8, the inverse of pack_brief."""
    words = np.ascontiguousarray(words, dtype="<u4")
    bits = np.unpackbits(words.view(np.uint8).reshape(words.shape[0], -1), axis=1, bitorder="little")
    return np.ascontiguousarray(bits[:, :n_bits])


class HarrisOptions:
    def __init__(self):
        self.kMinFeatureDistance = 20
        self.kMinValidResponse = 40.0


class FeaturePointHarrisDetector:
    """feature_detector::FeaturePointHarrisDetector (un-vendored Feature_Detector; used by
    test/test_optical_flow.cpp:34-39) on the device, with this repo's definition (oracle/oracle_harris.c)."""

    def __init__(self, ctx: Optional[Context] = None):
        self._ctx = ctx
        self._options = HarrisOptions()

    def options(self) -> HarrisOptions:
        return self._options

    def DetectGoodFeatures(self, image, needed_feature_num: int):
        """Returns (ok, features) with features an (n, 2) float32 array of (u, v), strongest first."""
        ctx = self._ctx or default_context()
        pyr = image if isinstance(image, ImagePyramid) else ImagePyramid.from_host_levels([image], ctx)
        uv = np.zeros((max(1, int(needed_feature_num)), 2), dtype=np.float32)
        n = C.c_int32(0)
        N.check(N.lib().ftk_harris_detect(ctx.handle, pyr.handle, 0, int(needed_feature_num), int(self._options.kMinFeatureDistance),
                                          float(self._options.kMinValidResponse), _ptr(uv), C.byref(n)), ctx.handle)
        return True, uv[: n.value].copy()

    def response(self, image) -> np.ndarray:
        ctx = self._ctx or default_context()
        pyr = image if isinstance(image, ImagePyramid) else ImagePyramid.from_host_levels([image], ctx)
        _, rows, cols = pyr.level_desc(0)
        out = np.zeros((rows, cols), dtype=np.float32)
        N.check(N.lib().ftk_harris_response(ctx.handle, pyr.handle, 0, _ptr(out)), ctx.handle)
        return out


class BriefDescriptorOptions:
    def __init__(self):
        self.kLength = 256
        self.kHalfPatchSize = 8


class BriefDescriptor:
    """feature_detector::BriefDescriptor (un-vendored Feature_Detector; used by
    test/test_descriptor_matcher_brief.cpp:70-76) computed on the device with this repo's sampling
    pattern.  ``Compute`` returns the per-bit container the reference produces; ``compute_packed``
    returns the words ftk_hamming_match reads."""

    def __init__(self, ctx: Optional[Context] = None):
        self._ctx = ctx
        self._options = BriefDescriptorOptions()

    def options(self) -> BriefDescriptorOptions:
        return self._options

    def compute_packed(self, image, pixel_uv) -> np.ndarray:
        ctx = self._ctx or default_context()
        uv = np.ascontiguousarray(pixel_uv, dtype=np.float32).reshape(-1, 2)
        n, n_bits = uv.shape[0], int(self._options.kLength)
        words = np.zeros((n, (n_bits + 31) // 32), dtype=np.uint32)
        if n == 0:
            return words
        pyr = image if isinstance(image, ImagePyramid) else ImagePyramid.from_host_levels([image], ctx)
        N.check(N.lib().ftk_brief_compute(ctx.handle, pyr.handle, 0, _ptr(uv), n, n_bits, int(self._options.kHalfPatchSize), _ptr(words)),
                ctx.handle)
        return words

    def Compute(self, image, pixel_uv):
        """Returns (ok, descriptors) with descriptors a (n, kLength) 0/1 uint8 array."""
        if int(self._options.kLength) <= 0 or int(self._options.kHalfPatchSize) <= 0:
            return False, np.zeros((0, 0), np.uint8)
        return True, unpack_brief(self.compute_packed(image, pixel_uv), int(self._options.kLength))


class DescriptorMatcherOptions:
    """descriptor_matcher.h:16-20."""

    def __init__(self):
        self.kMaxValidPredictRowDistance = 40
        self.kMaxValidPredictColDistance = 40
        self.kMaxValidDescriptorDistance = 0.0


class BriefMatcher:
    """DescriptorMatcher<BriefType> with the Hamming ComputeDistance of
    test/test_descriptor_matcher_brief.cpp:27-46.  Descriptors are per-bit arrays (n, n_bits)."""

    def __init__(self, ctx: Optional[Context] = None):
        self._ctx = ctx
        self._options = DescriptorMatcherOptions()

    def options(self) -> DescriptorMatcherOptions:
        return self._options

    def _match(self, descriptors_ref, descriptors_cur, pred_uv, cur_uv, index_pairs):
        ref_bits = np.asarray(descriptors_ref, dtype=np.uint8)
        cur_bits = np.asarray(descriptors_cur, dtype=np.uint8)
        n_ref, n_cur = ref_bits.shape[0], cur_bits.shape[0]
        n_bits = ref_bits.shape[1] if ref_bits.ndim == 2 else 0
        if n_cur == 0:
            return False, index_pairs  # descriptor_matcher.h:58
        # index_pairs reset only on size mismatch (descriptor_matcher.h:60-62)
        if index_pairs is None or np.asarray(index_pairs).size != n_ref:
            idx = np.full(n_ref, -1, dtype=np.int32)
        else:
            idx = np.array(index_pairs, dtype=np.int32).copy()
        ctx = self._ctx or default_context()
        ref_words = pack_brief(ref_bits.reshape(n_ref, n_bits))
        cur_words = pack_brief(cur_bits.reshape(n_cur, n_bits))
        ok = C.c_int(0)
        o = self._options
        rc = N.lib().ftk_hamming_match(ctx.handle, _ptr(ref_words), n_ref, _ptr(cur_words), n_cur, ref_words.shape[1], n_bits,
                                       float(o.kMaxValidDescriptorDistance), _ptr(pred_uv), _ptr(cur_uv), int(o.kMaxValidPredictColDistance),
                                       int(o.kMaxValidPredictRowDistance), _ptr(idx), C.byref(ok))
        N.check(rc, ctx.handle)
        return bool(ok.value), idx

    def ForceMatch(self, descriptors_ref, descriptors_cur, index_pairs_in_cur=None):
        """descriptor_matcher.h:55-79.  Returns (ok, index_pairs_in_cur)."""
        return self._match(descriptors_ref, descriptors_cur, None, None, index_pairs_in_cur)

    def NearbyMatch(self, descriptors_ref, descriptors_cur, pixel_uv_pred_in_cur, pixel_uv_cur, index_pairs_in_cur=None):
        """descriptor_matcher.h:90-124.  Returns (ok, index_pairs_in_cur)."""
        pred = np.ascontiguousarray(pixel_uv_pred_in_cur, dtype=np.float32).reshape(-1, 2)
        cur = np.ascontiguousarray(pixel_uv_cur, dtype=np.float32).reshape(-1, 2)
        n_ref, n_cur = len(descriptors_ref), len(descriptors_cur)
        if n_cur == 0 or n_ref != pred.shape[0] or n_cur != cur.shape[0]:
            return False, index_pairs_in_cur  # descriptor_matcher.h:94-96
        return self._match(descriptors_ref, descriptors_cur, pred, cur, index_pairs_in_cur)

    def FillMatchedPixelByPairIndices(self, index_pairs_in_cur, pixel_uv_cur, status=None):
        """descriptor_matcher.h:135-157.  Returns (matched_pixel_uv_cur, status)."""
        idx = np.ascontiguousarray(index_pairs_in_cur, dtype=np.int32)
        cur = np.ascontiguousarray(pixel_uv_cur, dtype=np.float32).reshape(-1, 2)
        n_ref = idx.shape[0]
        st = np.zeros(n_ref, dtype=np.uint8) if (status is None or np.asarray(status).size != n_ref) else np.array(status, dtype=np.uint8).copy()
        matched = np.zeros((n_ref, 2), dtype=np.float32)
        rc = N.lib().ftk_fill_matched_pixels(_ptr(idx), n_ref, _ptr(cur), cur.shape[0], _ptr(matched), _ptr(st))
        N.check(rc, None)
        return matched, st

    def ForceMatchPixels(self, descriptors_ref, descriptors_cur, pixel_uv_cur, status=None):
        """The pixel-returning ForceMatch overload (descriptor_matcher.h:81-88)."""
        ok, idx = self.ForceMatch(descriptors_ref, descriptors_cur, None)
        if not ok:
            return False, None, status
        matched, st = self.FillMatchedPixelByPairIndices(idx, pixel_uv_cur, status)
        return True, matched, st

    def NearbyMatchPixels(self, descriptors_ref, descriptors_cur, pixel_uv_pred_in_cur, pixel_uv_cur, status=None):
        """The pixel-returning NearbyMatch overload (descriptor_matcher.h:126-133)."""
        ok, idx = self.NearbyMatch(descriptors_ref, descriptors_cur, pixel_uv_pred_in_cur, pixel_uv_cur, None)
        if not ok:
            return False, None, status
        matched, st = self.FillMatchedPixelByPairIndices(idx, pixel_uv_cur, status)
        return True, matched, st


class CosineMatcher(BriefMatcher):
    """DescriptorMatcher<FloatDescriptor> with the cosine ComputeDistance of the reference's SuperPoint /
    DISK callers (test/test_descriptor_matcher_superpoint.cpp:26-35, test_descriptor_matcher_disk.cpp:26-35):
    0.5 - ref.dot(cur) / ref.norm() / cur.norm() * 0.5.  Descriptors are float arrays (n, dim).  Shares the
    ForceMatch / NearbyMatch / FillMatchedPixelByPairIndices surface with BriefMatcher; only the device
    entry point differs (ftk_cosine_match: fp16 MFMA shortlist + exact fp32 decision)."""

    def _match(self, descriptors_ref, descriptors_cur, pred_uv, cur_uv, index_pairs):
        ref = np.ascontiguousarray(descriptors_ref, dtype=np.float32)
        cur = np.ascontiguousarray(descriptors_cur, dtype=np.float32)
        n_ref, n_cur = ref.shape[0], cur.shape[0]
        if n_cur == 0:
            return False, index_pairs  # descriptor_matcher.h:58
        dim = cur.shape[1]
        if n_ref and ref.shape[1] != dim:
            raise ValueError(f"descriptor sizes differ: ref {ref.shape[1]}, cur {dim}")
        if index_pairs is None or np.asarray(index_pairs).size != n_ref:
            idx = np.full(n_ref, -1, dtype=np.int32)
        else:
            idx = np.array(index_pairs, dtype=np.int32).copy()
        ctx = self._ctx or default_context()
        ok = C.c_int(0)
        o = self._options
        rc = N.lib().ftk_cosine_match(ctx.handle, _ptr(ref), n_ref, _ptr(cur), n_cur, dim, float(o.kMaxValidDescriptorDistance),
                                      _ptr(pred_uv), _ptr(cur_uv), int(o.kMaxValidPredictColDistance), int(o.kMaxValidPredictRowDistance),
                                      _ptr(idx), C.byref(ok))
        N.check(rc, ctx.handle)
        return bool(ok.value), idx


SuperpointMatcher = CosineMatcher  # test/test_descriptor_matcher_superpoint.cpp:26
DiskMatcher = CosineMatcher        # test/test_descriptor_matcher_disk.cpp:26


class DirectMethodOptions:
    """direct_method_tracker.h:20-28 — same field names and defaults."""

    def __init__(self):
        self.kMaxTrackPointsNumber = 500
        self.kMaxIteration = 15
        self.kPatchRowHalfSize = 6
        self.kPatchColHalfSize = 6
        self.kMaxConvergeStep = 1e-6
        self.kMaxConvergeResidual = 2.0
        self.kMethod = "direct"  # DirectMethodMethod::kDirect (kInverse / kFast are empty stubs in the reference)

    def to_native(self) -> N.DirectOptions:
        o = N.DirectOptions()
        o.max_track_points = int(self.kMaxTrackPointsNumber)
        o.max_iteration = int(self.kMaxIteration)
        o.half_rows = int(self.kPatchRowHalfSize)
        o.half_cols = int(self.kPatchColHalfSize)
        o.max_converge_step = float(self.kMaxConvergeStep)
        o.max_converge_residual = float(self.kMaxConvergeResidual)
        o.method = N.METHODS[self.kMethod] if isinstance(self.kMethod, str) else int(self.kMethod)
        return o


def _f32(x):
    return np.float32(x)


def _quat_mul(a, b):
    """Eigen::Quaternionf product for the reference's SSE2 build, (w, x, y, z) fp32 (oracle/oracle_direct_method.c)."""
    aw, ax, ay, az = (_f32(v) for v in a)
    bw, bx, by, bz = (_f32(v) for v in b)
    x = _f32(_f32(_f32(ax * bw) - _f32(az * by)) + _f32(_f32(ay * bz) + _f32(aw * bx)))
    y = _f32(_f32(_f32(ay * bw) - _f32(ax * bz)) + _f32(_f32(az * bx) + _f32(aw * by)))
    z = _f32(_f32(_f32(az * bw) - _f32(ay * bx)) + _f32(_f32(ax * by) + _f32(aw * bz)))
    w = _f32(_f32(_f32(aw * bw) - _f32(ax * bx)) - _f32(_f32(az * bz) + _f32(ay * by)))
    return np.array([w, x, y, z], dtype=np.float32)


def _quat_inverse(q):
    w, x, y, z = (_f32(v) for v in q)
    n2 = _f32(_f32(_f32(x * x) + _f32(z * z)) + _f32(_f32(y * y) + _f32(w * w)))
    if not n2 > 0:
        return np.zeros(4, dtype=np.float32)
    return np.array([_f32(w / n2), _f32(-x / n2), _f32(-y / n2), _f32(-z / n2)], dtype=np.float32)


def _quat_rotate(q, v):
    w, x, y, z = (_f32(c) for c in q)
    v = np.asarray(v, dtype=np.float32)
    qv = (x, y, z)

    def cross(a, b):
        return [_f32(_f32(a[1] * b[2]) - _f32(a[2] * b[1])), _f32(_f32(a[2] * b[0]) - _f32(a[0] * b[2])), _f32(_f32(a[0] * b[1]) - _f32(a[1] * b[0]))]

    uv = cross(qv, v)
    uv = [_f32(c + c) for c in uv]
    c2 = cross(qv, uv)
    return np.array([_f32(_f32(v[k] + _f32(w * uv[k])) + c2[k]) for k in range(3)], dtype=np.float32)


class DirectMethod:
    """feature_tracker::DirectMethod (direct_method_tracker.h:30-78): photometric 6-DoF pose alignment of one
    frame against a reference frame over all features jointly.  The camera-frame overload runs on the device
    (ftk_direct_track); the world-frame overload (direct_method_tracker.cpp:8-33) is the same quaternion algebra
    around it, evaluated in fp32 on the host.  Quaternions are (w, x, y, z)."""

    def __init__(self, ctx: Optional[Context] = None):
        self._ctx = ctx
        self._options = DirectMethodOptions()
        self.last_iterations = 0

    def options(self) -> DirectMethodOptions:
        return self._options

    def TrackFeatures(self, ref_pyramid: ImagePyramid, cur_pyramid: ImagePyramid, K, p_c_in_ref, ref_pixel_uv, cur_pixel_uv=None,
                      q_rc=(1.0, 0.0, 0.0, 0.0), p_rc=(0.0, 0.0, 0.0), status=None):
        """Camera-frame overload (direct_method_tracker.cpp:35-86).  Returns (ok, cur_pixel_uv, q_rc, p_rc, status)."""
        ref_uv = np.ascontiguousarray(ref_pixel_uv, dtype=np.float32).reshape(-1, 2)
        n = ref_uv.shape[0]
        q = np.array(q_rc, dtype=np.float32).reshape(4).copy()
        p = np.array(p_rc, dtype=np.float32).reshape(3).copy()
        cur_uv = None if cur_pixel_uv is None else np.asarray(cur_pixel_uv, dtype=np.float32).reshape(-1, 2)
        st = None if status is None else np.asarray(status, dtype=np.uint8).reshape(-1)
        # RETURN_FALSE_IF(ref_pixel_uv.empty()) / level mismatch (:38-39)
        if n == 0 or cur_pyramid.level() != ref_pyramid.level():
            return False, (np.zeros((0, 2), np.float32) if cur_uv is None else cur_uv), q, p, (np.zeros(0, np.uint8) if st is None else st)
        pts = np.ascontiguousarray(p_c_in_ref, dtype=np.float32).reshape(-1, 3)
        if pts.shape[0] < min(n, int(self._options.kMaxTrackPointsNumber)):
            raise ValueError("p_c_in_ref has fewer points than features to track")
        if pts.shape[0] < n:
            pts = np.concatenate([pts, np.zeros((n - pts.shape[0], 3), np.float32)], axis=0)
        cur_uv = ref_uv.copy() if (cur_uv is None or cur_uv.shape[0] != n) else np.ascontiguousarray(cur_uv).copy()  # :42-44
        valid = st is not None and st.shape[0] == n
        st = np.ascontiguousarray(st).copy() if valid else np.zeros(n, dtype=np.uint8)
        ctx = self._ctx or default_context()
        opt = self._options.to_native()
        Kf = np.ascontiguousarray(K, dtype=np.float32).reshape(4)
        it = C.c_uint32(0)
        rc = N.lib().ftk_direct_track(ctx.handle, C.byref(opt), ref_pyramid.handle, cur_pyramid.handle, _ptr(Kf), _ptr(pts), _ptr(ref_uv), _ptr(cur_uv),
                                      n, _ptr(q), _ptr(p), _ptr(st), int(valid), C.byref(it))
        N.check(rc, ctx.handle)
        self.last_iterations = int(it.value)
        return True, cur_uv, q, p, st

    def TrackFeaturesWorld(self, ref_pyramid: ImagePyramid, cur_pyramid: ImagePyramid, K, ref_q_wc, ref_p_wc, p_w, ref_pixel_uv, cur_pixel_uv=None,
                           cur_q_wc=(1.0, 0.0, 0.0, 0.0), cur_p_wc=(0.0, 0.0, 0.0), status=None):
        """World-frame overload (direct_method_tracker.cpp:8-33).  Returns (ok, cur_pixel_uv, cur_q_wc, cur_p_wc, status)."""
        ref_q_wc = np.asarray(ref_q_wc, dtype=np.float32)
        ref_p_wc = np.asarray(ref_p_wc, dtype=np.float32)
        cur_q_wc = np.asarray(cur_q_wc, dtype=np.float32)
        cur_p_wc = np.asarray(cur_p_wc, dtype=np.float32)
        ref_q_cw = _quat_inverse(ref_q_wc)
        p_w = np.asarray(p_w, dtype=np.float32).reshape(-1, 3)
        p_c_in_ref = np.stack([_quat_rotate(ref_q_cw, (pw - ref_p_wc).astype(np.float32)) for pw in p_w], axis=0) if len(p_w) else np.zeros((0, 3), np.float32)
        q_rc = _quat_mul(ref_q_cw, cur_q_wc)
        p_rc = _quat_rotate(ref_q_cw, (cur_p_wc - ref_p_wc).astype(np.float32))
        ok, cur_uv, q_rc, p_rc, st = self.TrackFeatures(ref_pyramid, cur_pyramid, K, p_c_in_ref, ref_pixel_uv, cur_pixel_uv, q_rc, p_rc, status)
        if not ok:
            return False, cur_uv, cur_q_wc, cur_p_wc, st
        out_q = _quat_mul(ref_q_wc, q_rc)
        out_p = (_quat_rotate(ref_q_wc, p_rc) + ref_p_wc).astype(np.float32)
        return True, cur_uv, out_q, out_p, st
