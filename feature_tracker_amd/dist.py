"""Feature / descriptor-row sharding across the GPUs of one node.

Features are independent units (basic_klt.cpp:13-54 keeps no cross-feature state), so the path
shards by block-partitioning the feature list over ranks, replicating both pyramids, and joining
the per-rank results with ONE all-gather of the packed result shard ([u, v] float32 pairs followed
by status bytes).  ``torch.distributed`` carries the collective: backend "nccl" (= RCCL over xGMI)
on GPUs, "gloo" in the CPU tests.  No other exchange step exists on this path.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world_size: int, rank: int):
    """Contiguous block partition; the first (n % world_size) ranks get one extra feature."""
    base, extra = divmod(n, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_capacity(n: int, world_size: int) -> int:
    """Per-rank slot count of the gathered buffer (ranks with fewer features pad)."""
    return (n + world_size - 1) // world_size


def packed_bytes(capacity: int) -> int:
    """Bytes of one rank's packed result shard: capacity * (8 B uv + 1 B status), rounded up to 16 B
    so that every rank's slice of the gathered buffer keeps the uv pairs 4-byte aligned."""
    return (capacity * 9 + 15) // 16 * 16


def pack_views(buffer, capacity: int):
    """Split one rank's packed byte buffer (torch uint8 tensor, packed_bytes(capacity) bytes) into
    (uv float32 [capacity, 2], status uint8 [capacity]) views that alias it."""
    import torch

    uv = buffer[: capacity * 8].view(torch.float32).view(capacity, 2)
    status = buffer[capacity * 8: capacity * 9]
    return uv, status


def unpack_gathered(gathered, n: int, world_size: int):
    """gathered: uint8 tensor [world_size * capacity * 9] -> (uv [n, 2], status [n]) in global feature order."""
    import torch

    cap = shard_capacity(n, world_size)
    per = packed_bytes(cap)
    uvs, sts = [], []
    for r in range(world_size):
        b, e = shard_bounds(n, world_size, r)
        chunk = gathered[r * per:(r + 1) * per]
        uv, st = pack_views(chunk, cap)
        uvs.append(uv[: e - b])
        sts.append(st[: e - b])
    return torch.cat(uvs, dim=0), torch.cat(sts, dim=0)


def all_gather_results(local_packed, world_size: int, group=None, force_collective: bool = False, out=None, async_op: bool = False):
    """One collective per call: every rank contributes its packed shard and receives all of them.
    Returns the gathered buffer, or (buffer, work) when async_op is set — the caller then overlaps the
    collective with the next launch and calls work.wait() before reusing `local_packed`."""
    import torch
    import torch.distributed as dist

    if out is None:
        out = torch.empty(local_packed.numel() * world_size, dtype=torch.uint8, device=local_packed.device)
    if world_size == 1 and not force_collective:
        out.copy_(local_packed)
        return (out, None) if async_op else out
    work = dist.all_gather_into_tensor(out, local_packed, group=group, async_op=async_op)
    return (out, work) if async_op else out


def split_numpy(arr: np.ndarray, world_size: int, rank: int) -> np.ndarray:
    b, e = shard_bounds(arr.shape[0], world_size, rank)
    return arr[b:e]


class ShardedKlt:
    """TrackFeatures over the GPUs of one node (BASELINE.json configs[4]).

    Every rank holds both pyramids (replicated) and the full feature list; rank r tracks the
    contiguous block ``shard_bounds(n, world, r)`` with ONE kernel launch, writes (u, v) + status into
    its packed shard, and one all-gather gives every rank the complete result in the original feature
    order — identical to the single-GPU result because features do not interact.

    ``kMaxTrackPointsNumber`` is a GLOBAL cap in the reference (only features [0, cap) are tracked,
    basic_klt.cpp:9), so rank r tracks the first ``clamp(cap - begin, 0, m)`` features of its block and
    passes the rest through, exactly as the unsharded call does.

    ``tracker`` is any object with ``track(ref_uv, cur_uv_in, status_in, cur_uv_out, status_out, iters,
    max_track_points=None)`` on torch tensors (``feature_tracker_amd.device.DeviceKlt`` on GPUs);
    ``max_track_points`` (default: the tracker's own ``max_track_points`` attribute, i.e. its options) is the
    global cap.
    """

    def __init__(self, tracker, n: int, device, world_size: int = 1, rank: int = 0, group=None, max_track_points=None):
        import torch

        self.tracker, self.n, self.world, self.rank, self.group = tracker, int(n), int(world_size), int(rank), group
        cap = max_track_points if max_track_points is not None else getattr(tracker, "max_track_points", None)
        self.global_cap = None if cap is None else int(cap)
        self.begin, self.end = shard_bounds(self.n, self.world, self.rank)
        self.cap = shard_capacity(self.n, self.world)
        self.packed = torch.zeros(packed_bytes(self.cap), dtype=torch.uint8, device=device)
        self.gathered = torch.empty(packed_bytes(self.cap) * self.world, dtype=torch.uint8, device=device)
        self.uv_view, self.status_view = pack_views(self.packed, self.cap)

    def launch_local(self, ref_uv, cur_uv_in, status_in, iters=None):
        """Enqueue only the local shard's kernel (its packed shard is complete when the stream reaches this point): for callers
        that place the all-gather themselves, e.g. on a side stream beside the next launch."""
        m = self.end - self.begin
        if m > 0:
            kw = {}
            if self.global_cap is not None:
                kw["max_track_points"] = max(0, min(self.global_cap - self.begin, m))  # this block's share of the global cap
            self.tracker.track(ref_uv[self.begin:self.end], cur_uv_in[self.begin:self.end], status_in[self.begin:self.end],
                               self.uv_view[:m], self.status_view[:m], None if iters is None else iters[self.begin:self.end], **kw)

    def gather(self, force_collective=None):
        """The all-gather of the packed shards on the current stream; returns the gathered byte buffer."""
        force = self.world > 1 if force_collective is None else force_collective
        return all_gather_results(self.packed, self.world, group=self.group, force_collective=force, out=self.gathered)

    def launch(self, ref_uv, cur_uv_in, status_in, iters=None):
        """Enqueue the local shard's kernel and the all-gather; returns the gathered byte buffer (asynchronous on GPUs)."""
        self.launch_local(ref_uv, cur_uv_in, status_in, iters)
        return self.gather()

    def track(self, ref_uv, cur_uv_in, status_in, iters=None):
        """Returns (cur_uv [n, 2], status [n]) for ALL features, in order."""
        return unpack_gathered(self.launch(ref_uv, cur_uv_in, status_in, iters), self.n, self.world)


class ShardedMatcher:
    """ForceMatch / NearbyMatch over the GPUs of one node ("shard ref rows, replicate cur").

    Rows of the reference set are independent (descriptor_matcher.h:67-76), so rank r matches the contiguous
    block ``shard_bounds(n_ref, world, r)`` of ref descriptors against ALL candidates and one all-gather of the
    int32 index shards gives every rank the complete ``index_pairs_in_cur`` — identical to the single-GPU result.

    ``match`` is any callable ``match(ref_rows, cur_rows, pred_rows_or_None, cur_uv_or_None, index_inout)`` on torch
    tensors that writes the shard's indices in place (``feature_tracker_amd.device.hamming_match_device`` /
    ``cosine_match_device`` behind a lambda on GPUs)."""

    def __init__(self, match, n_ref: int, device, world_size: int = 1, rank: int = 0, group=None):
        import torch

        self.match, self.n, self.world, self.rank, self.group = match, int(n_ref), int(world_size), int(rank), group
        self.begin, self.end = shard_bounds(self.n, self.world, self.rank)
        self.cap = shard_capacity(self.n, self.world)
        self.local = torch.full((self.cap,), -1, dtype=torch.int32, device=device)
        self.gathered = torch.empty(self.cap * self.world, dtype=torch.int32, device=device)

    def launch(self, ref_desc, cur_desc, pred_uv=None, cur_uv=None, index_pairs=None):
        """Enqueue the shard's match and the all-gather; returns the gathered [world * cap] int32 buffer."""
        import torch
        import torch.distributed as dist

        m = self.end - self.begin
        # index_pairs is in/out in the reference (stale entries survive, descriptor_matcher.h:60-62): seed the shard
        if index_pairs is not None:
            self.local[:m].copy_(index_pairs[self.begin:self.end])
        else:
            self.local.fill_(-1)
        if m > 0:
            self.match(ref_desc[self.begin:self.end], cur_desc, None if pred_uv is None else pred_uv[self.begin:self.end], cur_uv, self.local[:m])
        if self.world == 1:
            self.gathered.copy_(self.local)
        else:
            dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
        return self.gathered

    def match_all(self, ref_desc, cur_desc, pred_uv=None, cur_uv=None, index_pairs=None):
        """Returns index_pairs_in_cur [n_ref] for ALL reference descriptors, in order."""
        import torch

        g = self.launch(ref_desc, cur_desc, pred_uv, cur_uv, index_pairs)
        parts = []
        for r in range(self.world):
            b, e = shard_bounds(self.n, self.world, r)
            parts.append(g[r * self.cap: r * self.cap + (e - b)])
        return torch.cat(parts, dim=0)
