"""GPU parity tests: Hamming ForceMatch / NearbyMatch against the CPU oracle (indices bit-exact)."""
import numpy as np
import pytest

from feature_tracker_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["default", "launches"], autouse=True)
def matcher_form(request, switch):
    """Every test runs twice: with the default dispatch — the one-launch form (hamming_match_small_kernel: small calls) and the boxes + scan + epilogue launches by size — and with the one-launch form
    switched off (FTK_MATCH_SMALL=0, read per call), so that small inputs also reach the kernels that serve the large ones."""
    if request.param == "launches":
        switch("FTK_MATCH_SMALL", "0")


def matcher(ftk, max_dist, col=40, row=40):
    m = ftk.BriefMatcher()
    m.options().kMaxValidDescriptorDistance = max_dist
    m.options().kMaxValidPredictColDistance = col
    m.options().kMaxValidPredictRowDistance = row
    return m


@pytest.mark.parametrize("n_ref,n_cur,n_bits", [(1000, 1000, 256), (777, 1300, 256), (300, 257, 128), (100, 5000, 64), (64, 64, 200), (50, 70, 512)])
def test_force_match(ftk, oracle, n_ref, n_cur, n_bits):
    ref, cur, perm = synth.make_descriptors(n_ref, n_cur, n_bits=n_bits, flips=max(1, n_bits // 13))
    ok_g, idx_g = matcher(ftk, 60.0).ForceMatch(ref, cur)
    ok_c, idx_c = oracle.force_match(ref, cur, 60.0)
    assert ok_g and ok_c
    assert np.array_equal(idx_g, idx_c)
    assert (idx_c >= 0).sum() > 0


def test_force_match_ties_and_threshold(ftk, oracle):
    """Duplicate candidates: the lowest j wins; distance == threshold never matches; default threshold 0 matches nothing."""
    rs = np.random.RandomState(3)
    ref = rs.randint(0, 2, size=(200, 256)).astype(np.uint8)
    cur = np.concatenate([ref[::-1], ref, ref[::2]], axis=0).copy()
    cur[5, :10] ^= 1  # distance exactly 10 for one pair
    for thr in (0.0, 1.0, 10.0, 10.5, 60.0, 300.0):
        ok_g, idx_g = matcher(ftk, thr).ForceMatch(ref, cur)
        ok_c, idx_c = oracle.force_match(ref, cur, thr)
        assert np.array_equal(idx_g, idx_c), thr
    ok_g, idx_g = ftk.BriefMatcher().ForceMatch(ref, cur)  # kMaxValidDescriptorDistance = 0
    assert (idx_g == -1).all()


def test_stale_index_pairs_survive(ftk, oracle):
    """index_pairs is reset only when its size differs (descriptor_matcher.h:60-62)."""
    ref, cur, _ = synth.make_descriptors(120, 90, n_bits=256, flips=20)
    stale = np.arange(120, dtype=np.int32) + 1000
    ok_g, idx_g = matcher(ftk, 25.0).ForceMatch(ref, cur, stale)
    ok_c, idx_c = oracle.force_match(ref, cur, 25.0, stale)
    assert np.array_equal(idx_g, idx_c)
    assert (idx_g >= 1000).any()
    ok_g, idx_g = matcher(ftk, 25.0).ForceMatch(ref, cur, stale[:7])
    ok_c, idx_c = oracle.force_match(ref, cur, 25.0, stale[:7])
    assert np.array_equal(idx_g, idx_c)


def test_empty_inputs(ftk, oracle):
    ref, cur, _ = synth.make_descriptors(10, 10)
    ok, _ = matcher(ftk, 60.0).ForceMatch(ref, cur[:0])
    assert ok is False  # descriptor_matcher.h:58
    ok, idx = matcher(ftk, 60.0).ForceMatch(ref[:0], cur)
    assert ok is True and idx.size == 0
    # empty descriptors: ComputeDistance = kMaxInt32 (test_descriptor_matcher_brief.cpp:34-36)
    e_ref = np.zeros((4, 0), np.uint8)
    e_cur = np.zeros((6, 0), np.uint8)
    for thr in (60.0, 3e9):
        ok_g, idx_g = matcher(ftk, thr).ForceMatch(e_ref, e_cur)
        ok_c, idx_c = oracle.force_match(e_ref, e_cur, thr)
        assert ok_g == ok_c and np.array_equal(idx_g, idx_c), thr


@pytest.mark.parametrize("window", [(50, 50), (5, 80), (0, 0), (1000, 1000)])
def test_nearby_match(ftk, oracle, window):
    n = 1500
    ref, cur, perm = synth.make_descriptors(n, n, flips=20)
    rs = np.random.RandomState(11)
    cur_uv = np.stack([rs.uniform(0, 640, n), rs.uniform(0, 480, n)], axis=1).astype(np.float32)
    pred_uv = np.empty_like(cur_uv)
    pred_uv[perm] = cur_uv + rs.uniform(-30, 30, size=(n, 2)).astype(np.float32)  # ref perm[j] is predicted near cur j
    m = matcher(ftk, 60.0, col=window[0], row=window[1])
    ok_g, idx_g = m.NearbyMatch(ref, cur, pred_uv, cur_uv)
    ok_c, idx_c = oracle.nearby_match(ref, cur, pred_uv, cur_uv, 60.0, max_col=window[0], max_row=window[1])
    assert ok_g == ok_c
    assert np.array_equal(idx_g, idx_c)
    # pixel-returning overload (descriptor_matcher.h:126-157)
    ok, matched, st = m.NearbyMatchPixels(ref, cur, pred_uv, cur_uv)
    omatched, ost = oracle.fill_matched_pixels(idx_c, cur_uv)
    assert np.array_equal(st, ost) and np.array_equal(matched, omatched)


def test_nearby_size_checks(ftk):
    ref, cur, _ = synth.make_descriptors(10, 10)
    uv = np.zeros((10, 2), np.float32)
    ok, _ = matcher(ftk, 60.0).NearbyMatch(ref, cur, uv[:9], uv)
    assert ok is False  # descriptor_matcher.h:95
    ok, _ = matcher(ftk, 60.0).NearbyMatch(ref, cur, uv, uv[:9])
    assert ok is False  # :96


def test_config4_brute_force_property(ftk):
    """BASELINE.json configs[3] size (10 000 x 10 000): cur[j] is ref[perm[j]] with 20 flips, random pairs sit near 128,
    so the match of ref i must be the inverse permutation — a size-independent check that needs no CPU oracle."""
    n = 10000
    ref, cur, perm = synth.make_descriptors(n, n, flips=20)
    ok, idx = matcher(ftk, 60.0).ForceMatch(ref, cur)
    assert ok
    inv = np.empty(n, np.int32)
    inv[perm] = np.arange(n, dtype=np.int32)
    assert np.array_equal(idx, inv)


def test_sharded_matcher_on_device_world_size_1(ftk, oracle):
    """dist.ShardedMatcher driving the device entry points (world size 1: the shard is everything, no collective)."""
    import torch
    from feature_tracker_amd import device as D
    from feature_tracker_amd import dist as FD
    ref, cur, _ = synth.make_descriptors(700, 900, flips=20)
    fref, fcur, _ = synth.make_float_descriptors(700, 900, dim=128)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        d_ref = torch.from_numpy(ftk.pack_brief(ref).view(np.int32)).to(dev)
        d_cur = torch.from_numpy(ftk.pack_brief(cur).view(np.int32)).to(dev)
        sm = FD.ShardedMatcher(lambda r, c, p, u, idx: D.hamming_match_device(ctx, r, c, 256, 60.0, idx, p, u), 700, dev)
        got = sm.match_all(d_ref, d_cur).cpu().numpy()
        d_fr, d_fc = torch.from_numpy(fref).to(dev), torch.from_numpy(fcur).to(dev)
        sc = FD.ShardedMatcher(lambda r, c, p, u, idx: D.cosine_match_device(ctx, r, c, 0.1, idx, p, u), 700, dev)
        gotf = sc.match_all(d_fr, d_fc).cpu().numpy()
    assert np.array_equal(got, oracle.force_match(ref, cur, 60.0)[1])
    assert np.array_equal(gotf, oracle.match_float(fref, fcur, 0.1)[1])


@pytest.mark.parametrize("n,n_bits,window", [(3000, 256, 30), (2500, 128, 8), (1500, 512, 200)])
def test_nearby_match_in_spatial_order(ftk, oracle, n, n_bits, window):
    """Features in raster order: workgroups whose candidates cannot reach any window of their rows leave before they
    load a descriptor.  Same indices as the scalar loop, including NaN coordinates (which pass every window test) and
    a window of zero."""
    rs = np.random.RandomState(23)
    ref, cur, perm = synth.make_descriptors(n, n, n_bits=n_bits, flips=max(1, n_bits // 13))
    cur_uv = rs.uniform(0, 752, size=(n, 2)).astype(np.float32)
    order = np.lexsort((cur_uv[:, 0], np.floor(cur_uv[:, 1] / 4)))
    cur, cur_uv = np.ascontiguousarray(cur[order]), np.ascontiguousarray(cur_uv[order])
    inv = np.empty(n, np.int64)
    inv[order] = np.arange(n)
    pred_uv = rs.uniform(0, 752, size=(n, 2)).astype(np.float32)
    partner = np.full(n, -1, np.int64)
    partner[perm] = inv
    near = (rs.rand(n) < 0.5) & (partner >= 0)
    pred_uv[near] = cur_uv[partner[near]] + rs.uniform(-window, window, size=(int(near.sum()), 2)).astype(np.float32) * np.float32(0.8)
    rorder = np.lexsort((pred_uv[:, 0], np.floor(pred_uv[:, 1] / 4)))
    ref, pred_uv = np.ascontiguousarray(ref[rorder]), np.ascontiguousarray(pred_uv[rorder])
    pred_uv[5] = np.nan
    pred_uv[n // 2, 1] = np.nan
    cur_uv[7, 0] = np.nan
    cur_uv[n - 3] = np.nan
    for col, row in ((window, window // 2 + 1), (0, 0)):
        m = matcher(ftk, 70.0, col=col, row=row)
        with np.errstate(all="ignore"):
            ok_g, g = m.NearbyMatch(ref, cur, pred_uv, cur_uv)
            ok_c, c = oracle.nearby_match(ref, cur, pred_uv, cur_uv, 70.0, max_col=col, max_row=row)
        assert ok_g and ok_c
        assert np.array_equal(g, c), (col, row, np.flatnonzero(g != c)[:10])


@pytest.mark.parametrize("kernel", ["mfma", "scalar", "lds"])
def test_every_scan_kernel_gives_the_oracle_indices(ftk, oracle, kernel, switch):
    """The three Hamming scans behind ftk_hamming_match (matrix cores for 256 / 512 bits, popcount with the candidates on the
    scalar path, popcount with LDS tiles; FTK_MATCH_KERNEL picks one, read per call) on the same inputs: thresholds below,
    at and far above the distances that occur (the early exits key on the threshold), duplicates (lowest j wins), a
    candidate count that is not a multiple of any tile, reference counts around the 64-row and 512-row blocks, NearbyMatch
    windows with NaN coordinates, and stale indices that must survive."""
    switch("FTK_MATCH_KERNEL", kernel)
    rs = np.random.RandomState(17)
    for n_ref, n_cur, n_bits in ((65, 33, 256), (513, 1001, 256), (1500, 2100, 512), (200, 777, 256), (130, 95, 512)):
        ref, cur, _ = synth.make_descriptors(n_ref, n_cur, n_bits=n_bits, flips=n_bits // 12)
        cur[n_cur // 2] = cur[3]  # a duplicate candidate further down: the lower index must win
        ref[7] = 0
        cur[11] = 0              # all-zero descriptors on both sides
        cur_uv = rs.uniform(0, 300, (n_cur, 2)).astype(np.float32)
        pred_uv = rs.uniform(0, 300, (n_ref, 2)).astype(np.float32)
        pred_uv[5, 0] = np.nan   # passes every window test (descriptor_matcher.h:108-111)
        cur_uv[9, 1] = np.nan
        stale = np.arange(n_ref, dtype=np.int32) + 7000
        for thr in (0.0, 12.0, float(n_bits // 12), 60.0, 1000.0):
            ok_g, idx_g = matcher(ftk, thr).ForceMatch(ref, cur, stale.copy())
            ok_c, idx_c = oracle.force_match(ref, cur, thr, stale.copy())
            assert np.array_equal(idx_g, idx_c), (kernel, n_ref, n_cur, n_bits, thr)
            ok_g, idx_g = matcher(ftk, thr, 35, 80).NearbyMatch(ref, cur, pred_uv, cur_uv, stale.copy())
            ok_c, idx_c = oracle.nearby_match(ref, cur, pred_uv, cur_uv, thr, 35, 80, stale.copy())
            assert np.array_equal(idx_g, idx_c), (kernel, n_ref, n_cur, n_bits, thr, "nearby")
