"""Device-resident entry points for callers that already hold their data in HBM (torch tensors).

PyTorch is plumbing here: it owns the device memory and the HIP stream; the computation is the
``ftk_*_device`` part of the C ABI (hand-written HIP kernels).  Used by bench.py, the smoke test
and the multi-GPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _native as N
from .tracker import Context, ImagePyramid, OpticalFlowOptions


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("feature_tracker_amd.device needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch


def context_on_stream(stream, device_index: Optional[int] = None) -> Context:
    """A context that launches on the given ``torch.cuda.Stream`` (not the legacy null stream), so
    torch events, collectives and allocations made under ``with torch.cuda.stream(stream)`` order
    correctly with the kernels."""
    torch = _torch()
    if device_index is None:
        device_index = stream.device.index if stream.device.index is not None else torch.cuda.current_device()
    handle = stream.cuda_stream
    if not handle:
        raise ValueError("pass a non-default torch.cuda.Stream (the null stream has no handle to borrow)")
    return Context(device_index, handle)


def pyramid_from_tensors(levels: Sequence, ctx: Context) -> ImagePyramid:
    """levels: uint8 CUDA tensors [rows, cols], contiguous.  Borrowed, not copied."""
    desc = []
    for t in levels:
        if t.dtype.__str__() != "torch.uint8" or not t.is_cuda or not t.is_contiguous() or t.dim() != 2:
            raise ValueError("pyramid levels must be contiguous 2-D uint8 CUDA tensors")
        desc.append((t.data_ptr(), t.shape[0], t.shape[1]))
    return ImagePyramid.from_device_levels(desc, ctx, keepalive=list(levels))


def upload_pyramid(host_levels: Sequence[np.ndarray], ctx: Context, device) -> ImagePyramid:
    torch = _torch()
    tensors = [torch.from_numpy(np.ascontiguousarray(l)).to(device) for l in host_levels]
    return pyramid_from_tensors(tensors, ctx)


class DeviceKlt:
    """Runs one tracker variant on device-resident feature buffers.

    ``track(ref_uv, cur_uv_in, status_in, cur_uv_out, status_out, iters=None)`` enqueues ONE kernel on
    the context's stream and returns immediately; the out tensors may alias the in tensors.
    """

    def __init__(self, model: str, options: OpticalFlowOptions, ref_pyr: ImagePyramid, cur_pyr: ImagePyramid, ctx: Context,
                 prior: Optional[np.ndarray] = None, consider_luminance: bool = False, single_level: bool = False):
        self.model = N.MODELS[model]
        self.opt = options.to_native()
        self.ref_pyr, self.cur_pyr, self.ctx = ref_pyr, cur_pyr, ctx
        self.prior = None if prior is None else np.ascontiguousarray(prior, dtype=np.float32).reshape(4)
        self.lum = int(bool(consider_luminance))
        self.single = int(bool(single_level))

    def bind(self, ref_uv, cur_uv_in, status_in, cur_uv_out, status_out, iters=None):
        """Pre-marshals one launch on fixed buffers; the returned callable enqueues it with minimal
        host work (for launch-rate-sensitive loops).  The tensors must outlive the callable."""
        fn = N.lib().ftk_klt_track_device
        args = (self.ctx.handle, self.model, C.byref(self.opt), self.ref_pyr.handle, self.cur_pyr.handle, C.c_void_p(ref_uv.data_ptr()),
                C.c_void_p(cur_uv_in.data_ptr()), C.c_void_p(cur_uv_out.data_ptr()), C.c_void_p(status_in.data_ptr()),
                C.c_void_p(status_out.data_ptr()), ref_uv.shape[0], None if self.prior is None else self.prior.ctypes.data_as(C.c_void_p),
                self.lum, self.single, None if iters is None else C.c_void_p(iters.data_ptr()))
        keep = (ref_uv, cur_uv_in, status_in, cur_uv_out, status_out, iters)
        handle = self.ctx.handle

        def launch(_fn=fn, _args=args, _keep=keep):
            rc = _fn(*_args)
            if rc != 0:
                N.check(rc, handle)

        return launch

    @property
    def max_track_points(self) -> int:
        """kMaxTrackPointsNumber of this tracker's options (the GLOBAL cap when the feature list is sharded)."""
        return int(self.opt.max_track_points)

    def _args(self, ref_uv, cur_uv_in, status_in):
        return (C.c_void_p(ref_uv.data_ptr()), C.c_void_p(cur_uv_in.data_ptr()), C.c_void_p(status_in.data_ptr()))

    def track_sharded(self, comm: "Comm", ref_uv, cur_uv_in, status_in, cur_uv_out, status_out, iters=None):
        """ftk_klt_track_sharded_device: this rank's block of the (full-length) buffers, RCCL all-gather, scatter — every rank's
        out tensors hold all n results afterwards (stream-ordered)."""
        r, c, s = self._args(ref_uv, cur_uv_in, status_in)
        rc = N.lib().ftk_klt_track_sharded_device(
            self.ctx.handle, comm.handle, self.model, C.byref(self.opt), self.ref_pyr.handle, self.cur_pyr.handle, r, c, C.c_void_p(cur_uv_out.data_ptr()),
            s, C.c_void_p(status_out.data_ptr()), ref_uv.shape[0], None if self.prior is None else self.prior.ctypes.data_as(C.c_void_p), self.lum, self.single,
            None if iters is None else C.c_void_p(iters.data_ptr()))
        N.check(rc, self.ctx.handle)

    def bind_sharded(self, comm: "Comm", ref_uv, cur_uv_in, status_in, cur_uv_out, status_out):
        """Pre-marshalled track_sharded on fixed buffers (launch-rate-sensitive loops, HIP-graph capture)."""
        r, c, s = self._args(ref_uv, cur_uv_in, status_in)
        fn = N.lib().ftk_klt_track_sharded_device
        args = (self.ctx.handle, comm.handle, self.model, C.byref(self.opt), self.ref_pyr.handle, self.cur_pyr.handle, r, c, C.c_void_p(cur_uv_out.data_ptr()), s,
                C.c_void_p(status_out.data_ptr()), ref_uv.shape[0], None if self.prior is None else self.prior.ctypes.data_as(C.c_void_p), self.lum, self.single, None)
        keep = (comm, ref_uv, cur_uv_in, status_in, cur_uv_out, status_out)
        handle = self.ctx.handle

        def launch(_fn=fn, _args=args, _keep=keep):
            rc = _fn(*_args)
            if rc != 0:
                N.check(rc, handle)

        return launch

    def track_shard(self, rank: int, world: int, ref_uv, cur_uv_in, status_in, packed_shard, iters=None):
        """ftk_klt_track_shard_device: rank's block tracked into its packed shard (for callers with their own collective)."""
        r, c, s = self._args(ref_uv, cur_uv_in, status_in)
        rc = N.lib().ftk_klt_track_shard_device(
            self.ctx.handle, int(rank), int(world), self.model, C.byref(self.opt), self.ref_pyr.handle, self.cur_pyr.handle, r, c, s, ref_uv.shape[0],
            None if self.prior is None else self.prior.ctypes.data_as(C.c_void_p), self.lum, self.single, C.c_void_p(packed_shard.data_ptr()),
            None if iters is None else C.c_void_p(iters.data_ptr()))
        N.check(rc, self.ctx.handle)

    def unpack_shards(self, gathered, n: int, world: int, cur_uv_out, status_out):
        N.check(N.lib().ftk_klt_unpack_shards_device(self.ctx.handle, C.c_void_p(gathered.data_ptr()), int(n), int(world), C.c_void_p(cur_uv_out.data_ptr()),
                                                     C.c_void_p(status_out.data_ptr())), self.ctx.handle)

    def track(self, ref_uv, cur_uv_in, status_in, cur_uv_out, status_out, iters=None, max_track_points=None):
        """``max_track_points`` overrides the options' cap for this launch (a shard's share of the global cap)."""
        n = ref_uv.shape[0]
        opt = self.opt
        if max_track_points is not None and int(max_track_points) != int(opt.max_track_points):
            opt = N.KltOptions.from_buffer_copy(self.opt)
            opt.max_track_points = int(max_track_points)
        rc = N.lib().ftk_klt_track_device(
            self.ctx.handle, self.model, C.byref(opt), self.ref_pyr.handle, self.cur_pyr.handle, C.c_void_p(ref_uv.data_ptr()),
            C.c_void_p(cur_uv_in.data_ptr()), C.c_void_p(cur_uv_out.data_ptr()), C.c_void_p(status_in.data_ptr()),
            C.c_void_p(status_out.data_ptr()), n, None if self.prior is None else self.prior.ctypes.data_as(C.c_void_p), self.lum, self.single,
            None if iters is None else C.c_void_p(iters.data_ptr()))
        N.check(rc, self.ctx.handle)


class Comm:
    """ftk_comm: the communicator of the native multi-GPU path (one process per GPU; RCCL all-gather issued by libftk_hip.so on
    the context's stream).  ``unique_id`` (128 bytes from ``Comm.unique_id()`` on rank 0, handed to the other ranks by any
    means) is required for world > 1; world == 1 with no id needs no RCCL."""

    def __init__(self, ctx: Context, rank: int = 0, world: int = 1, unique_id: Optional[bytes] = None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        out = C.c_void_p()
        buf = None if unique_id is None else C.create_string_buffer(bytes(unique_id), N.UNIQUE_ID_BYTES)
        N.check(N.lib().ftk_comm_create(ctx.handle, self.rank, self.world, buf, C.byref(out)), ctx.handle)
        self._handle = out

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(N.UNIQUE_ID_BYTES)
        N.check(N.lib().ftk_comm_unique_id(buf), None)
        return buf.raw

    @property
    def handle(self):
        return self._handle

    def close(self):
        if self._handle:
            N.lib().ftk_comm_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(n: int, world: int, rank: int):
    """ftk_shard_bounds: the native twin of feature_tracker_amd.dist.shard_bounds."""
    b, e = C.c_int32(), C.c_int32()
    N.lib().ftk_shard_bounds(n, world, rank, C.byref(b), C.byref(e))
    return b.value, e.value


def hamming_match_sharded_device(ctx: Context, comm: Comm, ref_words, cur_words, n_bits: int, max_distance: float, index_pairs, pred_uv=None,
                                 cur_uv=None, max_col: int = 40, max_row: int = 40):
    """ftk_hamming_match_sharded_device on torch tensors: every rank passes ALL reference rows; index_pairs is complete everywhere."""
    rc = N.lib().ftk_hamming_match_sharded_device(
        ctx.handle, comm.handle, C.c_void_p(ref_words.data_ptr()), ref_words.shape[0], C.c_void_p(cur_words.data_ptr()), cur_words.shape[0],
        ref_words.shape[1], int(n_bits), float(max_distance), None if pred_uv is None else C.c_void_p(pred_uv.data_ptr()),
        None if cur_uv is None else C.c_void_p(cur_uv.data_ptr()), int(max_col), int(max_row), C.c_void_p(index_pairs.data_ptr()))
    N.check(rc, ctx.handle)


def hamming_match_device(ctx: Context, ref_words, cur_words, n_bits: int, max_distance: float, index_pairs, pred_uv=None, cur_uv=None,
                         max_col: int = 40, max_row: int = 40, workspace=None):
    """ForceMatch (pred_uv None) / NearbyMatch on packed descriptors held in CUDA tensors ([n, words] int32/uint32)."""
    rc = N.lib().ftk_hamming_match_device(
        ctx.handle, C.c_void_p(ref_words.data_ptr()), ref_words.shape[0], C.c_void_p(cur_words.data_ptr()), cur_words.shape[0],
        ref_words.shape[1], int(n_bits), float(max_distance), None if pred_uv is None else C.c_void_p(pred_uv.data_ptr()),
        None if cur_uv is None else C.c_void_p(cur_uv.data_ptr()), int(max_col), int(max_row), C.c_void_p(index_pairs.data_ptr()),
        None if workspace is None else C.c_void_p(workspace.data_ptr()))
    N.check(rc, ctx.handle)


def cosine_match_device(ctx: Context, ref_desc, cur_desc, max_distance: float, index_pairs, pred_uv=None, cur_uv=None, max_col: int = 40,
                        max_row: int = 40):
    """ForceMatch (pred_uv None) / NearbyMatch on float descriptors held in CUDA tensors ([n, dim] float32, contiguous)."""
    rc = N.lib().ftk_cosine_match_device(
        ctx.handle, C.c_void_p(ref_desc.data_ptr()), ref_desc.shape[0], C.c_void_p(cur_desc.data_ptr()), cur_desc.shape[0],
        cur_desc.shape[1], float(max_distance), None if pred_uv is None else C.c_void_p(pred_uv.data_ptr()),
        None if cur_uv is None else C.c_void_p(cur_uv.data_ptr()), int(max_col), int(max_row), C.c_void_p(index_pairs.data_ptr()))
    N.check(rc, ctx.handle)


def brief_compute_device(ctx: Context, image_pyr: ImagePyramid, uv, n_bits: int, half_patch: int, words_out, level: int = 0):
    """BRIEF descriptors of CUDA-resident features straight into packed CUDA words ([n, ceil(n_bits/32)] int32)."""
    rc = N.lib().ftk_brief_compute_device(ctx.handle, image_pyr.handle, int(level), C.c_void_p(uv.data_ptr()), uv.shape[0], int(n_bits),
                                          int(half_patch), C.c_void_p(words_out.data_ptr()))
    N.check(rc, ctx.handle)


class DeviceDirectBatch:
    """A batch of DirectMethod pose problems (ftk_direct_track_batch_device): one workgroup per problem, ONE launch.
    Every tensor stays in HBM; ``problems`` is a list of dicts with keys ref, cur (ImagePyramid), K (4 floats),
    p_c_in_ref [n, 3], ref_uv [n, 2], cur_uv [n, 2] (in/out), pose [7] (q w,x,y,z then p; in/out), status [n] uint8,
    status_valid (bool) and optionally iterations (int32 [1])."""

    def __init__(self, options, problems, ctx: Context):
        self.ctx = ctx
        self.opt = options.to_native()
        self._keep = problems
        self.n = len(problems)
        self.table = (N.DirectProblem * self.n)()
        for k, pr in enumerate(problems):
            t = self.table[k]
            t.ref, t.cur = pr["ref"].handle, pr["cur"].handle
            for i in range(4):
                t.K[i] = float(pr["K"][i])
            t.d_p_c_in_ref = pr["p_c_in_ref"].data_ptr()
            t.d_ref_uv = pr["ref_uv"].data_ptr()
            t.d_cur_uv = pr["cur_uv"].data_ptr()
            t.n = int(pr["ref_uv"].shape[0])
            t.d_pose = pr["pose"].data_ptr()
            t.d_status = pr["status"].data_ptr()
            t.status_valid = int(bool(pr.get("status_valid", False)))
            it = pr.get("iterations")
            t.d_iterations = None if it is None else it.data_ptr()

    def track(self):
        N.check(N.lib().ftk_direct_track_batch_device(self.ctx.handle, C.byref(self.opt), self.table, self.n), self.ctx.handle)
