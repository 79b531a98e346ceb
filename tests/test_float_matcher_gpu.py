"""GPU parity tests: float-descriptor (cosine) ForceMatch / NearbyMatch against the CPU oracle — indices bit-exact.

The device path shortlists pairs with an fp16 MFMA contraction and decides on the distance evaluated in fp32 in
Eigen's reduction order (csrc/float_matcher_kernels.hip); these tests stress exactly the places where a shortlist
could lose the deciding pair: near-ties, exact ties, thresholds at the minimum, irregular (zero / NaN / huge) rows,
candidate-list overflow, ragged sizes and odd descriptor lengths."""
import numpy as np
import pytest

from feature_tracker_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["default", "launches"], autouse=True)
def matcher_form(request, switch):
    """Every test runs twice: with the default dispatch — the one-launch exact form (cosine_match_small_kernel: calls of few candidates) and the clear + prep + contraction + recheck pipeline by size — and with the one-launch form
    switched off (FTK_COSINE_SMALL=0, read per call), so that small inputs also reach the kernels that serve the large ones."""
    if request.param == "launches":
        switch("FTK_COSINE_SMALL", "0")


def matcher(ftk, max_dist, col=40, row=40):
    m = ftk.CosineMatcher()
    m.options().kMaxValidDescriptorDistance = max_dist
    m.options().kMaxValidPredictColDistance = col
    m.options().kMaxValidPredictRowDistance = row
    return m


@pytest.mark.parametrize("n_ref,n_cur,dim", [(1000, 1000, 256), (777, 1300, 128), (300, 257, 256), (100, 3000, 64), (129, 127, 96), (50, 70, 250), (60, 90, 100), (33, 40, 7), (20, 20, 3)])
def test_force_match(ftk, oracle, n_ref, n_cur, dim):
    ref, cur, _ = synth.make_float_descriptors(n_ref, n_cur, dim=dim)
    for thr in (0.1, 0.6):
        ok_g, idx_g = matcher(ftk, thr).ForceMatch(ref, cur)
        ok_c, idx_c = oracle.match_float(ref, cur, thr)
        assert ok_g and ok_c
        assert np.array_equal(idx_g, idx_c), (thr, np.flatnonzero(idx_g != idx_c)[:10])
    assert (idx_c >= 0).sum() > 0


def test_unnormalised_descriptors(ftk, oracle):
    """The distance normalises by the norms, so raw (un-normalised, differently scaled) rows must match the same way."""
    ref, cur, _ = synth.make_float_descriptors(400, 500, dim=128, normalize=False)
    rs = np.random.RandomState(1)
    ref *= rs.uniform(1e-3, 1e3, size=(400, 1)).astype(np.float32)
    cur *= rs.uniform(1e-3, 1e3, size=(500, 1)).astype(np.float32)
    ok_g, idx_g = matcher(ftk, 0.3).ForceMatch(ref, cur)
    ok_c, idx_c = oracle.match_float(ref, cur, 0.3)
    assert np.array_equal(idx_g, idx_c)


def test_random_pairs_no_true_match(ftk, oracle):
    """Independent random descriptors: all distances ~0.5, many near-ties inside the shortlist margin."""
    rs = np.random.RandomState(5)
    ref = rs.standard_normal((600, 256)).astype(np.float32)
    cur = rs.standard_normal((2500, 256)).astype(np.float32)
    ok_g, idx_g = matcher(ftk, 1.0).ForceMatch(ref, cur)
    ok_c, idx_c = oracle.match_float(ref, cur, 1.0)
    assert np.array_equal(idx_g, idx_c)
    assert (idx_c >= 0).all()


def test_exact_ties_lowest_index_and_strict_threshold(ftk, oracle):
    """Duplicated candidates tie exactly: the lowest j wins; a threshold equal to the minimum never matches."""
    ref, cur, _ = synth.make_float_descriptors(150, 150, dim=256, noise=0.3)
    cur = np.concatenate([cur[::-1], cur, cur[::3]], axis=0).copy()
    ok_g, idx_g = matcher(ftk, 0.5).ForceMatch(ref, cur)
    ok_c, idx_c = oracle.match_float(ref, cur, 0.5)
    assert np.array_equal(idx_g, idx_c)
    d_min = np.array([oracle.cosine_distance(ref[i], cur[idx_c[i]]) for i in range(5)], dtype=np.float32)
    for i in range(5):
        for thr in (d_min[i], np.nextafter(d_min[i], np.float32(1.0))):
            _, g = matcher(ftk, float(thr)).ForceMatch(ref[i:i + 1], cur)
            _, c = oracle.match_float(ref[i:i + 1], cur, float(thr))
            assert np.array_equal(g, c), (i, thr)
    _, idx0 = ftk.CosineMatcher().ForceMatch(ref, cur)  # kMaxValidDescriptorDistance = 0: only a distance < 0 could match
    _, idx0_c = oracle.match_float(ref, cur, 0.0)
    assert np.array_equal(idx0, idx0_c)


def test_candidate_overflow_falls_back_to_exact_scan(ftk, oracle):
    """More near-minimal candidates than the per-row list holds (100 copies of every cur row)."""
    ref, cur, _ = synth.make_float_descriptors(40, 12, dim=128)
    cur = np.tile(cur, (100, 1))
    ok_g, idx_g = matcher(ftk, 0.9).ForceMatch(ref, cur)
    ok_c, idx_c = oracle.match_float(ref, cur, 0.9)
    assert np.array_equal(idx_g, idx_c)
    assert (idx_c < 12).all()  # lowest index of each group of copies


def test_irregular_descriptors(ftk, oracle):
    """Zero, NaN, inf, huge and tiny rows never enter the fp16 shortlist; the exact scan reproduces the scalar result."""
    ref, cur, _ = synth.make_float_descriptors(200, 260, dim=256)
    ref[3] = 0.0
    ref[7, 5] = np.nan
    ref[11] *= np.float32(1e25)   # squares overflow -> norm inf
    ref[13] *= np.float32(1e-30)  # squares underflow -> norm 0
    ref[17] *= np.float32(1e15)   # norm finite but outside the regular range
    cur[2] = 0.0
    cur[9, 100] = np.inf
    cur[21] *= np.float32(1e-25)
    cur[40] *= np.float32(3e14)
    with np.errstate(all="ignore"):
        ok_g, idx_g = matcher(ftk, 0.7).ForceMatch(ref, cur)
        ok_c, idx_c = oracle.match_float(ref, cur, 0.7)
    assert np.array_equal(idx_g, idx_c), np.flatnonzero(idx_g != idx_c)
    # more irregular candidates than the side list holds: every row takes the exact scan
    cur2 = cur.copy()
    cur2[100:200] = 0.0
    with np.errstate(all="ignore"):
        _, g = matcher(ftk, 0.7).ForceMatch(ref, cur2)
        _, c = oracle.match_float(ref, cur2, 0.7)
    assert np.array_equal(g, c)


@pytest.mark.parametrize("n_ref,n_cur,dim,window", [(1000, 1000, 256, 50), (500, 800, 128, 20), (300, 300, 256, 1000)])
def test_nearby_match(ftk, oracle, n_ref, n_cur, dim, window):
    ref, cur, perm = synth.make_float_descriptors(n_ref, n_cur, dim=dim)
    rs = np.random.RandomState(9)
    cur_uv = rs.uniform(0, 752, size=(n_cur, 2)).astype(np.float32)
    pred_uv = rs.uniform(0, 752, size=(n_ref, 2)).astype(np.float32)
    hit = rs.rand(n_cur) < 0.5  # half of the true matches lie inside the window
    pred_uv[perm[hit]] = cur_uv[hit] + rs.uniform(-window, window, size=(int(hit.sum()), 2)).astype(np.float32) * np.float32(0.9)
    m = matcher(ftk, 0.2, col=window, row=window // 2 + 1)
    ok_g, idx_g = m.NearbyMatch(ref, cur, pred_uv, cur_uv)
    ok_c, idx_c = oracle.match_float(ref, cur, 0.2, pred_uv, cur_uv, max_col=window, max_row=window // 2 + 1)
    assert ok_g and ok_c
    assert np.array_equal(idx_g, idx_c)
    ok, matched, st = m.NearbyMatchPixels(ref, cur, pred_uv, cur_uv)
    m_c, st_c = oracle.fill_matched_pixels(idx_c, cur_uv)
    assert ok and np.array_equal(st, st_c) and np.array_equal(matched[st == 1], m_c[st_c == 1])


def test_stale_indices_and_empty_inputs(ftk, oracle):
    ref, cur, _ = synth.make_float_descriptors(120, 90, dim=128)
    stale = np.arange(120, dtype=np.int32) + 1000
    _, idx_g = matcher(ftk, 0.02).ForceMatch(ref, cur, stale)
    _, idx_c = oracle.match_float(ref, cur, 0.02, index_pairs=stale)
    assert np.array_equal(idx_g, idx_c) and (idx_g >= 1000).any()
    ok, _ = matcher(ftk, 0.5).ForceMatch(ref, cur[:0])
    assert ok is False  # descriptor_matcher.h:58
    ok, idx = matcher(ftk, 0.5).ForceMatch(ref[:0], cur)
    assert ok is True and idx.size == 0
    ok, _ = matcher(ftk, 0.5).NearbyMatch(ref, cur, np.zeros((5, 2), np.float32), np.zeros((90, 2), np.float32))
    assert ok is False  # descriptor_matcher.h:95


def test_full_size_properties(ftk):
    """10 000 x 10 000 x 256 (BASELINE config 4's matcher shape, float variant): size-independent properties —
    every row finds its planted partner, and the result is invariant under a permutation of the candidates."""
    n = 10000
    ref, cur, perm = synth.make_float_descriptors(n, n, dim=256, noise=0.2)
    ok, idx = matcher(ftk, 0.1).ForceMatch(ref, cur)
    assert ok
    planted = np.full(n, -1, dtype=np.int64)
    planted[perm[::-1]] = np.arange(n)[::-1]  # lowest j among duplicates of a ref row
    assert np.array_equal(idx, planted.astype(np.int32))
    rs = np.random.RandomState(2)
    shuffle = rs.permutation(n)
    ok, idx2 = matcher(ftk, 0.1).ForceMatch(ref, cur[shuffle])
    assert np.array_equal(shuffle[idx2[idx2 >= 0]], idx[idx >= 0])


@pytest.mark.parametrize("env", [{}, {"FTK_COSINE_TWO_PASS": "1"}, {"FTK_COSINE_CHUNKED": "1"}, {"FTK_COSINE_SPLITS": "1"}, {"FTK_COSINE_SPLITS": "5"}])
def test_contraction_variants_agree_with_oracle(ftk, oracle, switch, env):
    """The shortlist has three implementations behind one decision rule (single walk with a running row maximum,
    maximum-then-collect, chunked for long descriptors) and a split count chosen from the grid size: each of them,
    at forced split counts, must give the oracle's indices — force and nearby, ragged sizes."""
    for k, v in env.items():
        switch(k, v)
    rs = np.random.RandomState(21)
    for n_ref, n_cur, dim in ((700, 2100, 256), (260, 1500, 128), (130, 513, 200)):
        ref, cur, _ = synth.make_float_descriptors(n_ref, n_cur, dim=dim, noise=0.25)
        _, g = matcher(ftk, 0.4).ForceMatch(ref, cur)
        _, c = oracle.match_float(ref, cur, 0.4)
        assert np.array_equal(g, c), (env, n_ref, n_cur, dim)
        cur_uv = rs.uniform(0, 400, size=(n_cur, 2)).astype(np.float32)
        pred_uv = rs.uniform(0, 400, size=(n_ref, 2)).astype(np.float32)
        m = matcher(ftk, 0.9, col=60, row=45)
        _, g = m.NearbyMatch(ref, cur, pred_uv, cur_uv)
        _, c = oracle.match_float(ref, cur, 0.9, pred_uv, cur_uv, max_col=60, max_row=45)
        assert np.array_equal(g, c), (env, "nearby", n_ref, n_cur, dim)


@pytest.mark.parametrize("splits", [None, "1", "2"])
def test_candidates_in_ascending_order_of_similarity(ftk, oracle, switch, splits):
    """Worst case for the running row maximum: every later cur row beats all earlier ones for one ref row, so the single
    walk collects an entry per tile until the list overflows and the row takes the exact scan.  Other rows see the
    same cur rows in an unrelated order.  Results must not depend on any of it."""
    if splits is not None:
        # one or two workgroups walk ALL tiles: the rows below then stage an entry per tile, more than a wave's staging
        # region holds, and must fall back to the exact scan
        switch("FTK_COSINE_SPLITS", splits)
    rs = np.random.RandomState(3)
    dim, n_cur = 256, 4096
    base = rs.standard_normal(dim).astype(np.float32)
    base /= np.linalg.norm(base)
    noise = rs.standard_normal((n_cur, dim)).astype(np.float32)
    noise /= np.linalg.norm(noise, axis=1, keepdims=True)
    w = np.linspace(0.0, 0.98, n_cur, dtype=np.float32)[:, None]  # similarity to `base` grows with j
    cur = (w * base[None, :] + (1.0 - w) * noise).astype(np.float32)
    ref = rs.standard_normal((300, dim)).astype(np.float32)
    ref[:40] = base[None, :] + 0.01 * rs.standard_normal((40, dim)).astype(np.float32)
    for thr in (0.5, 1.0):
        _, g = matcher(ftk, thr).ForceMatch(ref, cur)
        _, c = oracle.match_float(ref, cur, thr)
        assert np.array_equal(g, c)
    assert (c[:40] > n_cur // 2).all()


@pytest.mark.parametrize("dim", [256, 128, 64])
def test_runs_of_identical_neighbours(ftk, oracle, dim):
    """Identical (and nearly identical) candidates next to each other land in ONE lane's share of a tile: more than two
    scores inside the shortlist margin there make the kernel name the whole share.  The lowest index among exact
    duplicates must win, with and without a window, also when the run crosses a tile boundary and at the ragged end."""
    rs = np.random.RandomState(11)
    n_ref, n_cur = 300, 1000
    ref, cur, perm = synth.make_float_descriptors(n_ref, n_cur, dim=dim, noise=0.2)
    for start, run in ((10, 40), (60, 9), (500, 3), (n_cur - 7, 7)):
        cur[start:start + run] = cur[start]                         # exact duplicates
    cur[700:730] = cur[700] * (1.0 + 3e-5 * rs.standard_normal((30, dim))).astype(np.float32)  # near duplicates
    ok, g = matcher(ftk, 0.6).ForceMatch(ref, cur)
    ok, c = oracle.match_float(ref, cur, 0.6)
    assert np.array_equal(g, c), np.flatnonzero(g != c)[:10]
    assert np.isin(c, [10, 60, 500, n_cur - 7]).sum() > 0           # some rows do pick the first of a run
    cur_uv = rs.uniform(0, 300, size=(n_cur, 2)).astype(np.float32)
    pred_uv = rs.uniform(0, 300, size=(n_ref, 2)).astype(np.float32)
    m = matcher(ftk, 0.9, col=80, row=60)
    _, g = m.NearbyMatch(ref, cur, pred_uv, cur_uv)
    _, c = oracle.match_float(ref, cur, 0.9, pred_uv, cur_uv, max_col=80, max_row=60)
    assert np.array_equal(g, c)


@pytest.mark.parametrize("dim", [256, 100])
def test_device_buffers_not_16_byte_aligned(ftk, oracle, dim):
    """Device-resident descriptors handed over at an address that is only 4-byte aligned (a view into a larger tensor):
    the packet-wide loads of the norm / conversion kernels do not apply and the element-wise forms must give the same
    indices."""
    import torch
    from feature_tracker_amd import device as D
    ref, cur, _ = synth.make_float_descriptors(333, 777, dim=dim, noise=0.3)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        big_r = torch.zeros(ref.size + 3, dtype=torch.float32, device=dev)
        big_c = torch.zeros(cur.size + 1, dtype=torch.float32, device=dev)
        d_ref = big_r[3:].view(ref.shape)   # base + 12 bytes
        d_cur = big_c[1:].view(cur.shape)   # base + 4 bytes
        d_ref.copy_(torch.from_numpy(ref))
        d_cur.copy_(torch.from_numpy(cur))
        assert d_ref.data_ptr() % 16 != 0 and d_cur.data_ptr() % 16 != 0
        idx = torch.full((ref.shape[0],), -1, dtype=torch.int32, device=dev)
        D.cosine_match_device(ctx, d_ref, d_cur, 0.5, idx)
        got = idx.cpu().numpy()
    assert np.array_equal(got, oracle.match_float(ref, cur, 0.5)[1])


@pytest.mark.parametrize("n,dim,window", [(3000, 256, 30), (3000, 128, 8), (2500, 64, 200), (2100, 100, 25)])
def test_nearby_match_in_spatial_order(ftk, oracle, n, dim, window):
    """Features in raster order (what a detector scanning the image returns): a workgroup's rows see a band of the
    image and the kernel walks only the candidate tiles whose bounding box can reach a window.  Same indices as the
    scalar loop, including rows / candidates with NaN coordinates (which pass every window test) and a window that
    nothing falls into."""
    rs = np.random.RandomState(17)
    ref, cur, perm = synth.make_float_descriptors(n, n, dim=dim, noise=0.25)
    cur_uv = rs.uniform(0, 752, size=(n, 2)).astype(np.float32)
    order = np.lexsort((cur_uv[:, 0], np.floor(cur_uv[:, 1] / 4)))  # raster order in bands of 4 rows
    cur, cur_uv = cur[order], cur_uv[order]
    inv = np.empty(n, np.int64)
    inv[order] = np.arange(n)
    pred_uv = rs.uniform(0, 752, size=(n, 2)).astype(np.float32)
    # half of the rows predict the position of a candidate planted for them (+ a few pixels), the rest are elsewhere
    partner = np.full(n, -1, np.int64)
    partner[perm] = inv                     # ref row perm[j] was planted as candidate j, now at position inv[j]
    near = (rs.rand(n) < 0.5) & (partner >= 0)
    pred_uv[near] = cur_uv[partner[near]] + rs.uniform(-window, window, size=(int(near.sum()), 2)).astype(np.float32) * np.float32(0.8)
    rorder = np.lexsort((pred_uv[:, 0], np.floor(pred_uv[:, 1] / 4)))
    ref, pred_uv = ref[rorder], pred_uv[rorder]
    pred_uv[5] = np.nan
    pred_uv[n // 2, 1] = np.nan
    cur_uv[7, 0] = np.nan
    cur_uv[n - 3] = np.nan
    for col, row in ((window, window // 2 + 1), (0, 0)):
        m = matcher(ftk, 0.45, col=col, row=row)
        with np.errstate(all="ignore"):
            ok_g, g = m.NearbyMatch(ref, cur, pred_uv, cur_uv)
            ok_c, c = oracle.match_float(ref, cur, 0.45, pred_uv, cur_uv, max_col=col, max_row=row)
        assert ok_g and ok_c
        assert np.array_equal(g, c), (col, row, np.flatnonzero(g != c)[:10])
    assert (c >= 0).sum() >= 1  # the NaN rows / candidates still match something at window 0


@pytest.mark.parametrize("splits", [None, "1"])
def test_long_walk_over_many_candidates(ftk, oracle, switch, splits):
    """Few reference rows against 40 000 candidates: one or two workgroups per row group walk hundreds of tiles, so a
    wave's staging region fills several times over and is emptied inside the walk.  Random descriptors (no planted
    partner) make every row set new maxima again and again."""
    if splits is not None:
        switch("FTK_COSINE_SPLITS", splits)
    rs = np.random.RandomState(31)
    ref = rs.standard_normal((600, 64)).astype(np.float32)
    cur = rs.standard_normal((40000, 64)).astype(np.float32)
    cur[12345] = ref[17] * 3.0  # one exact partner
    ok, g = matcher(ftk, 1.0).ForceMatch(ref, cur)
    ok_c, c = oracle.match_float(ref, cur, 1.0)
    assert ok and ok_c
    assert np.array_equal(g, c), np.flatnonzero(g != c)[:10]
    assert c[17] == 12345


def test_large_set_properties(ftk):
    """40 000 x 40 000 x 128: more tiles than the NearbyMatch tile list holds per slice is not needed here, but the walk
    is 4 x longer per workgroup than at 10 000 and the grid has 79 row groups.  Size-independent properties: every row
    finds its planted partner (force and, with exact predictions, nearby), and a window that excludes the partner
    returns no match."""
    n = 40000
    ref, cur, perm = synth.make_float_descriptors(n, n, dim=128, noise=0.15)
    planted = np.full(n, -1, dtype=np.int64)
    planted[perm[::-1]] = np.arange(n)[::-1]
    ok, idx = matcher(ftk, 0.1).ForceMatch(ref, cur)
    assert ok and np.array_equal(idx, planted.astype(np.int32))
    rs = np.random.RandomState(8)
    cur_uv = rs.uniform(0, 4000, size=(n, 2)).astype(np.float32)
    pred_uv = np.zeros((n, 2), np.float32)
    has = planted >= 0
    pred_uv[has] = cur_uv[planted[has]]
    ok, idx = matcher(ftk, 0.1, col=3, row=3).NearbyMatch(ref, cur, pred_uv, cur_uv)
    assert ok and np.array_equal(idx[has], planted[has].astype(np.int32))
    far = pred_uv + np.float32(10000.0)
    ok, idx = matcher(ftk, 0.1, col=3, row=3).NearbyMatch(ref, cur, far, cur_uv)
    assert ok and (idx == -1).all()


@pytest.mark.parametrize("dim", [128, 256])
def test_nearby_scan_stops_at_the_first_exact_zero_distance(ftk, oracle, dim):
    """NearbyMatch leaves a row's scan at the first in-window candidate whose distance is exactly 0 (descriptor_matcher.h:119).
    A float distance can be slightly NEGATIVE (a cosine rounded above 1), so a later scaled duplicate with d < 0 would win a
    global argmin although the reference never visits it — while an earlier one does win.  Scaled copies of the ref row with
    d == 0 and with d < 0 (found with the oracle's own distance) are placed in every order around each other."""
    rs = np.random.RandomState(dim)
    n_ref, n_cur = 96, 2000
    ref = rs.standard_normal((n_ref, dim)).astype(np.float32)
    cur = rs.standard_normal((n_cur, dim)).astype(np.float32)
    cases = 0
    for i in range(n_ref):
        zero, neg = [], []
        for k in range(300):
            s = np.float32(0.5 + k * 0.0037)
            d = oracle.cosine_distance(ref[i], (ref[i] * s).astype(np.float32))
            (zero if d == 0 else neg if d < 0 else []).append(s)
        if not zero or not neg:
            continue
        cases += 1
        slots = np.sort(rs.choice(n_cur, size=4, replace=False))
        order = [(zero, neg, neg, zero), (neg, zero, neg, zero), (zero, zero, neg, neg), (neg, neg, zero, neg)][i % 4]
        for slot, pool in zip(slots, order):
            cur[slot] = ref[i] * pool[rs.randint(len(pool))]
    assert cases > n_ref // 2
    uv = np.zeros((n_ref, 2), np.float32)
    cuv = np.zeros((n_cur, 2), np.float32)  # every candidate inside every window
    for thr in (0.05, 0.6):
        ok_g, idx_g = matcher(ftk, thr).NearbyMatch(ref, cur, uv, cuv)
        ok_c, idx_c = oracle.match_float(ref, cur, thr, uv, cuv)
        assert ok_g and ok_c and np.array_equal(idx_g, idx_c), np.flatnonzero(idx_g != idx_c)[:10]
        okf, idx_f = matcher(ftk, thr).ForceMatch(ref, cur)
        okc, idx_fc = oracle.match_float(ref, cur, thr)
        assert np.array_equal(idx_f, idx_fc)
    assert (idx_g != idx_f).any()  # the stop changes the answer for some rows, so the case is exercised
