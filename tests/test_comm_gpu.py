"""GPU tests of the native multi-GPU entry points (include/ftk.h, "features sharded over the GPUs of one node").

One MI355X is all a test box has, so:
  * the collective path runs at world size 1 THROUGH RCCL (ftk_comm_unique_id + ncclCommInitRank(nranks = 1) +
    ncclAllGather issued by libftk_hip.so on the context stream);
  * world sizes 2..8 are covered by the two halves of the sharded call — every rank's block tracked into its packed
    shard (ftk_klt_track_shard_device), the shards laid side by side exactly as ncclAllGather would leave them, and
    the scatter (ftk_klt_unpack_shards_device) — on one device, compared bit for bit with the unsharded launch and the oracle.
The torch.distributed / gloo twin of this logic is tests/test_host_logic_cpu.py::test_two_rank_gloo_exchange."""
import numpy as np
import pytest

from feature_tracker_amd import dist as FD
from feature_tracker_amd import synth
from tests import scenes

pytestmark = pytest.mark.gpu


def _device_tracker(ftk, model, method, half, ref_levels, cur_levels, n, cap=None):
    import torch
    from feature_tracker_amd import device as D
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = D.context_on_stream(stream, 0)
    opt = ftk.OpticalFlowOptions()
    opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, half, half, n if cap is None else cap
    klt = D.DeviceKlt(model, opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
    return torch, D, dev, stream, ctx, klt


def test_shard_bounds_match_the_python_layer(ftk):
    from feature_tracker_amd import device as D
    from feature_tracker_amd import _native
    for n in (0, 1, 7, 2000, 200000, 200003):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                assert D.shard_bounds(n, world, rank) == FD.shard_bounds(n, world, rank)
            assert _native.lib().ftk_klt_shard_bytes(n, world) == (FD.packed_bytes(FD.shard_capacity(n, world)) if n else 0)


@pytest.mark.parametrize("with_rccl", [False, True])
def test_world_size_1_through_the_native_collective(ftk, oracle, with_rccl):
    ref_levels, cur_levels = scenes.scene(320, 240, 3)
    n = 777
    uv = scenes.features(n, 320, 240, half=5)
    torch, D, dev, stream, ctx, klt = _device_tracker(ftk, "basic", "inverse", 5, ref_levels, cur_levels, n)
    with torch.cuda.stream(stream):
        comm = D.Comm(ctx, 0, 1, D.Comm.unique_id() if with_rccl else None)
        d_ref = torch.from_numpy(uv).to(dev)
        d_out, d_st = torch.empty_like(d_ref), torch.empty(n, dtype=torch.uint8, device=dev)
        for _ in range(3):  # repeated calls reuse the communicator's buffers
            klt.track_sharded(comm, d_ref, d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev), d_out, d_st)
        stream.synchronize()
        comm.close()
    ok, c, s, _ = oracle.klt_track_pyramid("basic", ref_levels, cur_levels, uv, method="inverse", half=5, max_points=n)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32), c.view(np.uint32)) and np.array_equal(d_st.cpu().numpy(), s)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("model,method", [("basic", "inverse"), ("lssd", "fast"), ("affine", "inverse")])
def test_emulated_ranks_shard_and_unpack(ftk, oracle, world, model, method):
    """Every rank's block -> packed shard -> (what the all-gather leaves) -> scatter == the unsharded launch == the oracle; with a
    global kMaxTrackPointsNumber that cuts through the middle of a block."""
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "similarity")
    n = 1003  # not a multiple of any world size
    uv = scenes.features(n, 320, 240, half=5)
    pred = uv + np.float32([1.5, -1.0])
    status = (np.arange(n) % 7 == 0).astype(np.uint8) * 3  # some features arrive as kOutside and are passed through
    for cap in (n, 611):
        torch, D, dev, stream, ctx, klt = _device_tracker(ftk, model, method, 5, ref_levels, cur_levels, n, cap)
        with torch.cuda.stream(stream):
            d_ref, d_in, d_sin = torch.from_numpy(uv).to(dev), torch.from_numpy(pred).to(dev), torch.from_numpy(status).to(dev)
            shard = int(ftk_shard_bytes(n, world))
            gathered = torch.zeros(shard * world, dtype=torch.uint8, device=dev)
            for rank in range(world):
                klt.track_shard(rank, world, d_ref, d_in, d_sin, gathered[rank * shard:(rank + 1) * shard])
            d_out, d_st = torch.empty_like(d_ref), torch.empty(n, dtype=torch.uint8, device=dev)
            klt.unpack_shards(gathered, n, world, d_out, d_st)
            d_all, d_all_st = torch.empty_like(d_ref), torch.empty(n, dtype=torch.uint8, device=dev)
            klt.track(d_ref, d_in, d_sin, d_all, d_all_st)
            stream.synchronize()
        assert torch.equal(d_out.view(torch.int32), d_all.view(torch.int32)) and torch.equal(d_st, d_all_st), (world, cap)
        ok, c, s, _ = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, pred, status, method=method, half=5, max_points=cap)
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), c.view(np.uint32)) and np.array_equal(d_st.cpu().numpy(), s), (world, cap)
        # the python-level unpack (torch.distributed path) reads the same bytes the same way
        guv, gst = FD.unpack_gathered(gathered, n, world)
        assert torch.equal(guv.view(torch.int32), d_out.view(torch.int32)) and torch.equal(gst, d_st)


def ftk_shard_bytes(n, world):
    from feature_tracker_amd import _native
    return _native.lib().ftk_klt_shard_bytes(n, world)


@pytest.mark.parametrize("nearby", [False, True])
def test_hamming_match_sharded_world_1(ftk, oracle, nearby):
    import torch
    from feature_tracker_amd import device as D
    ref, cur, _ = synth.make_descriptors(900, 700, flips=20)
    rs = np.random.RandomState(2)
    cur_uv = rs.uniform(0, 300, (700, 2)).astype(np.float32)
    pred_uv = rs.uniform(0, 300, (900, 2)).astype(np.float32)
    stale = (np.arange(900, dtype=np.int32) + 5000)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        comm = D.Comm(ctx, 0, 1, D.Comm.unique_id())
        d_ref = torch.from_numpy(ftk.pack_brief(ref).view(np.int32)).to(dev)
        d_cur = torch.from_numpy(ftk.pack_brief(cur).view(np.int32)).to(dev)
        d_idx = torch.from_numpy(stale.copy()).to(dev)
        D.hamming_match_sharded_device(ctx, comm, d_ref, d_cur, 256, 40.0, d_idx, torch.from_numpy(pred_uv).to(dev) if nearby else None,
                                       torch.from_numpy(cur_uv).to(dev) if nearby else None, 80, 80)
        stream.synchronize()
        comm.close()
    if nearby:
        ok, want = oracle.nearby_match(ref, cur, pred_uv, cur_uv, 40.0, 80, 80, stale)
    else:
        ok, want = oracle.force_match(ref, cur, 40.0, stale)
    assert np.array_equal(d_idx.cpu().numpy(), want)


def test_a_failed_local_launch_poisons_its_shard_instead_of_leaving_the_collective(ftk):
    """when this rank's tracker launch fails the call must still take part in the all-gather (the peers would block in
    it for ever) — with a shard of 0xFF bytes — and report the local error; the host-buffer form turns a poisoned block into an
    error.  Forced here with a patch size the kernels refuse (FTK_E_UNSUPPORTED), world size 1 through RCCL."""
    import ctypes as C
    from feature_tracker_amd import _native
    ref_levels, cur_levels = scenes.scene(320, 240, 3)
    n = 300
    uv = scenes.features(n, 320, 240, half=5)
    torch, D, dev, stream, ctx, klt = _device_tracker(ftk, "basic", "inverse", 5, ref_levels, cur_levels, n)
    with torch.cuda.stream(stream):
        comm = D.Comm(ctx, 0, 1, D.Comm.unique_id())
        d_ref = torch.from_numpy(uv).to(dev)
        d_out, d_st = torch.zeros_like(d_ref), torch.zeros(n, dtype=torch.uint8, device=dev)
        klt.track_sharded(comm, d_ref, d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev), d_out, d_st)  # sizes the exchange buffers
        stream.synchronize()
        try:  # (a communicator left open by a failed assertion keeps the interpreter from exiting)
            klt.opt.half_rows = 1024  # outside [0, 1023]: fill_klt_params refuses it
            with pytest.raises(_native.FtkError) as e:
                klt.track_sharded(comm, d_ref, d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev), d_out, d_st)
            assert e.value.code == -4 and "half patch" in str(e.value)
            stream.synchronize()
            assert (d_st.cpu().numpy() == 0xFF).all() and (d_out.cpu().numpy().view(np.uint32) == 0xFFFFFFFF).all()
            # host-buffer form
            cur, st = uv.copy(), np.zeros(n, np.uint8)
            rc = _native.lib().ftk_klt_track_sharded(ctx.handle, comm.handle, klt.model, C.byref(klt.opt), klt.ref_pyr.handle, klt.cur_pyr.handle,
                                                     uv.ctypes.data_as(C.c_void_p), cur.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), n, None, 0, 0, None)
            assert rc != 0
            klt.opt.half_rows = 5  # and the communicator is still usable afterwards
            klt.track_sharded(comm, d_ref, d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev), d_out, d_st)
            stream.synchronize()
            assert (d_st.cpu().numpy() <= 4).all()
        finally:
            comm.close()


def test_comm_argument_errors(ftk):
    import torch
    from feature_tracker_amd import _native
    from feature_tracker_amd import device as D
    stream = torch.cuda.Stream(device=torch.device("cuda", 0))
    ctx = D.context_on_stream(stream, 0)
    with pytest.raises(_native.FtkError):
        D.Comm(ctx, 2, 2, D.Comm.unique_id())  # rank out of range
    with pytest.raises(_native.FtkError):
        D.Comm(ctx, 0, 2, None)  # world > 1 needs the id
