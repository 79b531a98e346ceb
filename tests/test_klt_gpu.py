"""GPU parity tests: the HIP trackers (through the C ABI) against the CPU oracle.

Contract (BASELINE.json north_star): tracked (u, v) within 1e-3 px of the CPU path, status codes
equal.  The kernels are designed to reproduce the scalar fp32 arithmetic in the same order, so the
tests assert the stronger property — bit-identical (u, v), status and iteration counts — and state
the contractual tolerance next to it.
"""
import numpy as np
import pytest

from feature_tracker_amd import synth
from tests import scenes

pytestmark = pytest.mark.gpu

TOL_PX = 1e-3  # north_star tolerance on tracked positions

MODELS = ["basic", "affine", "lssd"]
METHODS = ["inverse", "direct", "fast"]
CLASSES = {"basic": "OpticalFlowBasicKlt", "affine": "OpticalFlowAffineKlt", "lssd": "OpticalFlowLssdKlt"}


def make_tracker(ftk, model, method, half, half_cols=None, max_points=100000, **kw):
    klt = getattr(ftk, CLASSES[model])()
    o = klt.options()
    o.kMethod = method
    o.kPatchRowHalfSize = half
    o.kPatchColHalfSize = half if half_cols is None else half_cols
    o.kMaxTrackPointsNumber = max_points
    for k, v in kw.items():
        setattr(o, k, v)
    return klt


def oracle_kwargs(method, half, half_cols=None, max_points=100000, **kw):
    d = dict(method=method, half=half, half_cols=half_cols, max_points=max_points)
    if "kMaxIteration" in kw:
        d["max_iteration"] = kw["kMaxIteration"]
    if "kMaxToleranceLargeStep" in kw:
        d["max_large_step"] = kw["kMaxToleranceLargeStep"]
    if "kMaxConvergeStep" in kw:
        d["converge"] = kw["kMaxConvergeStep"]
    return d


def assert_parity(gpu, cpu, what=""):
    ok_g, uv_g, st_g, it_g = gpu
    ok_c, uv_c, st_c, it_c = cpu
    assert ok_g == ok_c, what
    assert np.array_equal(st_g, st_c), f"{what}: status differs at {np.nonzero(st_g != st_c)[0][:10]}"
    finite = np.isfinite(uv_c).all(axis=1)
    d = np.abs(uv_g[finite].astype(np.float64) - uv_c[finite].astype(np.float64))
    assert d.size == 0 or d.max() <= TOL_PX, f"{what}: max |duv| = {d.max()} px > {TOL_PX}"
    # design goal: bit-identical results
    assert np.array_equal(uv_g.view(np.uint32), uv_c.view(np.uint32)), (
        f"{what}: not bit-identical, {np.count_nonzero((uv_g.view(np.uint32) != uv_c.view(np.uint32)).any(axis=1))} of {len(uv_c)} differ, "
        f"max {d.max() if d.size else 0}")
    if it_g is not None:
        assert np.array_equal(it_g, it_c), f"{what}: iteration counts differ"


def run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half, cur_uv=None, status=None, prior=None, luminance=False, **kw):
    klt = make_tracker(ftk, model, method, half, **kw)
    if prior is not None:
        if model == "affine":
            klt.predict_affine = np.asarray(prior, np.float32)
        elif model == "lssd":
            klt.predict_R_cr = np.asarray(prior, np.float32)
    if model == "lssd":
        klt.consider_patch_luminance = luminance
    ref_pyr = ftk.ImagePyramid.from_host_levels(ref_levels)
    cur_pyr = ftk.ImagePyramid.from_host_levels(cur_levels)
    ok, c, s = klt.TrackFeatures(ref_pyr, cur_pyr, uv, cur_uv, status)
    gpu = (ok, c, s, klt.last_iterations)
    cpu = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, cur_uv, status, prior=prior, consider_luminance=luminance,
                                   **oracle_kwargs(method, half, **kw))
    return gpu, cpu


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("model", MODELS)
def test_all_variants_small(ftk, oracle, model, method):
    kind = "translation" if model == "basic" else "similarity"
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", kind)
    uv = scenes.features(400, 320, 240, half=5)
    gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=5)
    assert_parity(gpu, cpu, f"{model}/{method}")
    assert (cpu[2] == 1).sum() > 300  # the scene is trackable


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("model", MODELS)
def test_all_variants_hard_motion(ftk, oracle, model, method):
    ref_levels, cur_levels = scenes.scene(320, 240, 4, "hard", "similarity")
    uv = scenes.features(300, 320, 240, half=6, seed=99)
    gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=6)
    assert_parity(gpu, cpu, f"{model}/{method} hard")


def test_config1_basic_inverse(ftk, oracle):
    """BASELINE.json configs[0]: 200 features, 640x480, 3 levels, 11x11."""
    ref_levels, cur_levels = scenes.scene(640, 480, 3)
    uv = scenes.features(200, 640, 480, half=5, border_fraction=0.01)
    gpu, cpu = run_pyramid(ftk, oracle, "basic", "inverse", ref_levels, cur_levels, uv, half=5)
    assert_parity(gpu, cpu, "config1")


@pytest.mark.parametrize("method", METHODS)
def test_config2_basic(ftk, oracle, method):
    """BASELINE.json configs[1]: 2000 features, 640x480, 4 levels, 21x21."""
    ref_levels, cur_levels = scenes.scene(640, 480, 4)
    uv = scenes.features(2000, 640, 480, half=10, border_fraction=0.01)
    gpu, cpu = run_pyramid(ftk, oracle, "basic", method, ref_levels, cur_levels, uv, half=10)
    assert_parity(gpu, cpu, f"config2/{method}")
    assert (cpu[2] == 1).sum() >= 1950


@pytest.mark.parametrize("model", MODELS)
def test_textureless_patch(ftk, oracle, model):
    """Zero Hessian: Eigen's LDLT returns 0 (not NaN) -> ||v||^2 = 0 < threshold -> kTracked at once."""
    ref_levels, cur_levels = scenes.scene(160, 120, 2, kind="flat")
    uv = scenes.features(64, 160, 120, half=4)
    for method in METHODS:
        gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=4)
        assert_parity(gpu, cpu, f"flat {model}/{method}")


@pytest.mark.parametrize("model", MODELS)
def test_border_and_outside_features(ftk, oracle, model):
    """Features on / beyond the border exercise every validity path (partial patches, kOutside, zero valid pixels)."""
    ref_levels, cur_levels = scenes.scene(320, 240, 3)
    rs = np.random.RandomState(5)
    n = 256
    uv = np.empty((n, 2), np.float32)
    uv[:, 0] = rs.uniform(-12, 332, n)
    uv[:, 1] = rs.uniform(-12, 252, n)
    uv[:8] = [[0, 0], [319, 239], [0, 239], [319, 0], [0.5, 0.5], [318.5, 238.5], [-3, 100], [400, 400]]
    for method in METHODS:
        gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=5)
        assert_parity(gpu, cpu, f"border {model}/{method}")


def test_prediction_status_and_cap(ftk, oracle):
    """cur_uv prediction in, incoming status > kTracked skipped, kMaxTrackPointsNumber caps the loop (basic_klt.cpp:9,15)."""
    ref_levels, cur_levels = scenes.scene(320, 240, 3)
    uv = scenes.features(300, 320, 240, half=5)
    pred = uv + np.float32([2.5, -1.5])
    status = (np.arange(300) % 5).astype(np.uint8)
    for model in MODELS:
        for method in METHODS:
            gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=5, cur_uv=pred, status=status, max_points=250)
            assert_parity(gpu, cpu, f"pred {model}/{method}")
            # skipped / capped entries are untouched
            skip = (status > 1) | (np.arange(300) >= 250)
            assert np.array_equal(gpu[1][skip], pred[skip])
            assert np.array_equal(gpu[2][skip], status[skip])


def test_default_cap_is_500(ftk, oracle):
    ref_levels, cur_levels = scenes.scene(320, 240, 3)
    uv = scenes.features(600, 320, 240, half=5)
    gpu, cpu = run_pyramid(ftk, oracle, "basic", "fast", ref_levels, cur_levels, uv, half=5, max_points=500)
    assert_parity(gpu, cpu, "cap500")
    assert np.array_equal(gpu[1][500:], uv[500:]) and (gpu[2][500:] == 0).all()


def test_rectangular_patch_and_options(ftk, oracle):
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "hard")
    uv = scenes.features(200, 320, 240, half=7)
    for model in MODELS:
        for method in METHODS:
            gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=3, half_cols=7, kMaxIteration=6,
                                   kMaxToleranceLargeStep=2, kMaxConvergeStep=1e-3)
            assert_parity(gpu, cpu, f"rect {model}/{method}")


def test_lssd_luminance_and_prior(ftk, oracle):
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "similarity")
    uv = scenes.features(300, 320, 240, half=6)
    th = np.deg2rad(1.0)
    prior = np.float32([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    for method in METHODS:
        for lum in (False, True):
            gpu, cpu = run_pyramid(ftk, oracle, "lssd", method, ref_levels, cur_levels, uv, half=6, prior=prior, luminance=lum)
            assert_parity(gpu, cpu, f"lssd prior {method} lum={lum}")


@pytest.mark.parametrize("model", MODELS)
def test_single_level_overload(ftk, oracle, model):
    """TrackFeatures(GrayImage, GrayImage, ...): affine honours predict_affine_ only here; LSSD never writes cur back (sic)."""
    ref_levels, cur_levels = scenes.scene(320, 240, 1, "easy", "similarity")
    uv = scenes.features(200, 320, 240, half=6)
    pred = uv + np.float32([3.0, -2.0])
    prior = np.float32([[1.01, 0.02], [-0.02, 0.99]])
    for method in METHODS:
        klt = make_tracker(ftk, model, method, 6)
        if model == "affine":
            klt.predict_affine = prior
        if model == "lssd":
            klt.predict_R_cr = prior
        ok, c, s = klt.TrackFeatures(ref_levels[0], cur_levels[0], uv, pred, None)
        cpu = oracle.klt_track_single(model, ref_levels[0], cur_levels[0], uv, pred, None, prior=prior, **oracle_kwargs(method, 6))
        assert_parity((ok, c, s, klt.last_iterations), cpu, f"single {model}/{method}")
        if model == "lssd":
            assert np.array_equal(c, pred)


def test_api_error_behaviour(ftk):
    """Empty input and level mismatch return false before anything is touched (optical_flow.cpp:8-9)."""
    ref_levels, cur_levels = scenes.scene(160, 120, 2)
    klt = ftk.OpticalFlowBasicKlt()
    p2 = ftk.ImagePyramid.from_host_levels(ref_levels)
    p1 = ftk.ImagePyramid.from_host_levels(cur_levels[:1])
    ok, _, _ = klt.TrackFeatures(p2, p2, np.zeros((0, 2), np.float32))
    assert ok is False
    ok, _, _ = klt.TrackFeatures(p2, p1, np.float32([[50, 50]]))
    assert ok is False
    assert klt.OpticalFlowMethodName() == "Basic-Klt"


def test_extract_extend_patch(ftk, oracle):
    ref_levels, _ = scenes.scene(160, 120, 1)
    klt = ftk.OpticalFlowBasicKlt()
    for (u, v) in [(80.3, 60.7), (1.2, 1.9), (158.9, 118.2), (-5.0, 300.0)]:
        cnt, patch, valid = klt.ExtractExtendPatchInReferenceImage(ref_levels[0], (u, v), 9, 11)
        ocnt, opatch, ovalid = oracle.extract_extend_patch(ref_levels[0], u, v, 9, 11)
        assert cnt == ocnt
        assert np.array_equal(valid, ovalid.astype(bool))
        assert np.array_equal(patch.view(np.uint32), opatch.view(np.uint32))


def test_pyramid_build_matches_oracle(ftk, oracle):
    """Device CreateImagePyramid == truncating 2x2 box mean, including odd sizes."""
    from feature_tracker_amd import synth
    # one fused launch builds levels 1..6 from 64 x 64 tiles (pyramid_kernels.hip), deeper levels one by one: odd sizes, sizes
    # around the tile edge, more than seven levels, a full-HD frame
    for (w, h, levels) in [(640, 480, 4), (321, 243, 4), (37, 29, 3), (65, 64, 5), (63, 129, 6), (300, 260, 8), (1920, 1080, 5), (128, 128, 7), (2, 2, 2)]:
        img, _ = synth.make_image_pair(w, h)
        pyr = ftk.ImagePyramid.build(img, levels)
        ref = oracle.create_pyramid(img, levels)
        assert pyr.level() == levels
        for i in range(levels):
            assert np.array_equal(pyr.download_level(i), ref[i]), (w, h, i)


def test_pyramid_update_refills_an_existing_pyramid(ftk, oracle):
    """ftk_pyramid_update: the next frame into the same allocation == a freshly built pyramid == the oracle's, from host memory,
    from device memory and from pinned host memory (stream-ordered); and a tracker on the refilled pair gives the oracle's result."""
    import torch
    from feature_tracker_amd import _native, synth
    w, h, levels = 321, 243, 4
    img_a, img_b = synth.make_image_pair(w, h, (2.2, -1.4))
    pyr = ftk.ImagePyramid.build(img_a, levels)
    ref_b = oracle.create_pyramid(img_b, levels)
    pyr.update(img_b)
    for i in range(levels):
        assert np.array_equal(pyr.download_level(i), ref_b[i]), i
    ref_a = oracle.create_pyramid(img_a, levels)
    d_img = torch.from_numpy(img_a).cuda()
    torch.cuda.synchronize()
    pyr.update(d_img.data_ptr(), "device")
    for i in range(levels):
        assert np.array_equal(pyr.download_level(i), ref_a[i]), i
    pinned = torch.from_numpy(img_b.copy()).pin_memory()
    pyr.update(pinned.data_ptr(), "host_async")  # device-visible pinned memory: read by the pyramid launch itself, no copy
    for i in range(levels):  # download_level synchronises the context's stream
        assert np.array_equal(pyr.download_level(i), ref_b[i]), i
    pageable = np.ascontiguousarray(img_a.copy())
    pyr.update(int(pageable.ctypes.data), "host_async")  # not pinned: served by the copy path
    for i in range(levels):
        assert np.array_equal(pyr.download_level(i), ref_a[i]), i
    # an uploaded pyramid owns its level 0 too; a wrapped / device-borrowed one does not
    up = ftk.ImagePyramid.from_host_levels(ref_a)
    up.update(img_b)
    assert np.array_equal(up.download_level(levels - 1), ref_b[levels - 1])
    borrowed = ftk.ImagePyramid.build_from_device(d_img.data_ptr(), h, w, levels, keepalive=d_img)
    with pytest.raises(_native.FtkError):
        borrowed.update(img_b)
    with pytest.raises(ValueError):
        pyr.update(img_b[:-1])
    # tracking on refilled pyramids
    other = ftk.ImagePyramid.build(img_b, levels)
    other.update(img_a)  # ref := a
    pyr.update(img_b)    # cur := b
    uv = scenes.features(200, w, h, half=5)
    klt = ftk.OpticalFlowBasicKlt()
    klt.options().kMethod, klt.options().kPatchRowHalfSize, klt.options().kPatchColHalfSize = "inverse", 5, 5
    ok, c, s = klt.TrackFeatures(other, pyr, uv)
    ok_c, c_c, s_c, _ = oracle.klt_track_pyramid("basic", ref_a, ref_b, uv, method="inverse", half=5)
    assert ok and np.array_equal(s, s_c) and np.array_equal(c.view(np.uint32), c_c.view(np.uint32))


@pytest.mark.parametrize("model,method", [("basic", "inverse"), ("basic", "fast"), ("lssd", "inverse"), ("affine", "fast"), ("affine", "inverse")])
def test_throughput_mode_is_close_but_reported_not_asserted_exact(ftk, oracle, model, method):
    """ftk_set_reduction_mode(TREE): same products, butterfly sums.  Not the contract — the test only checks that it is a tracker
    (nearly every feature within 1e-2 px of the oracle, statuses almost all equal) and that switching back restores bit-exactness."""
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "similarity" if model != "basic" else "translation")
    uv = scenes.features(600, 320, 240, half=6)
    ctx = ftk.Context()
    cls = {"basic": ftk.OpticalFlowBasicKlt, "affine": ftk.OpticalFlowAffineKlt, "lssd": ftk.OpticalFlowLssdKlt}[model]
    rp, cp = ftk.ImagePyramid.from_host_levels(ref_levels, ctx), ftk.ImagePyramid.from_host_levels(cur_levels, ctx)
    ok_c, c_c, s_c, _ = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, method=method, half=6, max_points=600)

    def run():
        klt = cls(ctx)
        klt.options().kMethod, klt.options().kPatchRowHalfSize, klt.options().kPatchColHalfSize, klt.options().kMaxTrackPointsNumber = method, 6, 6, 600
        return klt.TrackFeatures(rp, cp, uv)

    ctx.set_reduction("tree")
    ok, c, s = run()
    both = (s == 1) & (s_c == 1)
    d = np.linalg.norm(c[both].astype(np.float64) - c_c[both].astype(np.float64), axis=1)
    assert ok and both.mean() > 0.9 and (s != s_c).mean() < 0.02
    assert np.percentile(d, 95) < 1e-2 and (d < 0.3).mean() > 0.99
    ctx.set_reduction("exact")
    ok, c, s = run()
    assert np.array_equal(s, s_c) and np.array_equal(c.view(np.uint32), c_c.view(np.uint32))
    from feature_tracker_amd import _native
    with pytest.raises(_native.FtkError):
        _native.check(_native.lib().ftk_set_reduction_mode(ctx.handle, 7), ctx.handle)


def test_warmup_and_build_info(ftk):
    from feature_tracker_amd import _native
    ctx = ftk.Context()
    ctx.warmup()       # every family
    ctx.warmup(1 | 2)  # again: idempotent
    info = _native.build_info()
    assert len(info["source_hash"]) == 16 and info["arch"] == "gfx950" and "mllvm" in info


@pytest.mark.parametrize("model", MODELS)
def test_tiny_image_and_coarse_levels_smaller_than_patch(ftk, oracle, model):
    """Pyramid levels smaller than the patch footprint: every window is clamped at the border."""
    from feature_tracker_amd import synth
    ref, cur = synth.make_image_pair(64, 48, (1.3, -0.8))
    ref_levels, cur_levels = synth.build_pyramid(ref, 3), synth.build_pyramid(cur, 3)
    uv = scenes.features(96, 64, 48, half=6, border_fraction=0.2)
    for method in METHODS:
        gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=6)
        assert_parity(gpu, cpu, f"tiny {model}/{method}")


@pytest.mark.parametrize("model,half", [("basic", 15), ("lssd", 12), ("affine", 9), ("basic", 0), ("lssd", 1)])
def test_large_and_degenerate_patch_sizes(ftk, oracle, model, half):
    ref_levels, cur_levels = scenes.scene(320, 240, 3)
    uv = scenes.features(80, 320, 240, half=max(half, 2))
    for method in METHODS:
        gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=half)
        assert_parity(gpu, cpu, f"half={half} {model}/{method}")


@pytest.mark.parametrize("half,half_cols", [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1)])
def test_affine_patches_smaller_than_one_product_group(ftk, oracle, half, half_cols):
    """1x1, 1x3, 3x1, 1x5, 3x3 affine patches: with fewer than four pixels the level setup's axis tables would overlap the
    zero padding of the first product group if they shared its space — the non-fast affine variants keep their own
    table space for such patches."""
    ref_levels, cur_levels = scenes.scene(320, 240, 3)
    uv = scenes.features(60, 320, 240, half=3)
    for method in METHODS:
        gpu, cpu = run_pyramid(ftk, oracle, "affine", method, ref_levels, cur_levels, uv, half=half, half_cols=half_cols)
        assert_parity(gpu, cpu, f"half=({half},{half_cols}) affine/{method}")


@pytest.mark.parametrize("model,method,half", [("affine", "inverse", 30), ("affine", "fast", 30), ("affine", "direct", 19), ("basic", "inverse", 40),
                                               ("basic", "fast", 70), ("lssd", "inverse", 24), ("lssd", "fast", 33)])
def test_patches_beyond_a_workgroups_lds_run_the_large_patch_form(ftk, oracle, model, method, half):
    """The reference has no patch-size ceiling (optical_flow.h:24-25: any int32 half size).  Patches whose per-pixel arrays exceed
    160 KB of LDS — 24 sums x 61 x 61 products for affine inverse at half 30, which round 3 refused with FTK_E_UNSUPPORTED — run
    the generic kernel with those arrays in device memory; the results are the oracle's, bit for bit.  Half 70 is also beyond the
    old limit of 63 on the half size itself."""
    ref_levels, cur_levels = scenes.scene(640, 480, 2)
    uv = scenes.features(24, 640, 480, half=min(half, 60), border_fraction=0.1)
    gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=half)
    assert_parity(gpu, cpu, f"large patch half={half} {model}/{method}")


@pytest.mark.parametrize("windows", [1, 2])
@pytest.mark.parametrize("model", MODELS)
def test_large_patch_form_forced_on_ordinary_patches(ftk, oracle, model, windows, switch):
    """FTK_KLT_SPILL=1 sends every variant through the large-patch form at an ordinary size (13 x 13, rectangular 5 x 9), =2 also
    without the LDS image windows (every tap from global memory — what patches from about 280 x 280 get); the feature list is
    tracked in three batches (a 1 MB budget of device memory), with a kMaxTrackPointsNumber that cuts the second batch."""
    switch("FTK_KLT_SPILL", str(windows))
    switch("FTK_KLT_SPILL_BUDGET_MB", "1")
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "hard", "similarity")
    uv = scenes.features(150, 320, 240, half=6, seed=5, border_fraction=0.05)
    for method in METHODS:
        for half, half_cols, cap in ((6, None, 100000), (2, 4, 70)):
            gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=half, half_cols=half_cols, max_points=cap,
                                   luminance=(model == "lssd" and method == "fast"))
            assert_parity(gpu, cpu, f"forced large-patch form ({windows}) {model}/{method} half=({half},{half_cols})")


def test_half_patch_size_beyond_1023_is_a_clean_error(ftk):
    from feature_tracker_amd import _native
    ref_levels, cur_levels = scenes.scene(160, 120, 1)
    klt = make_tracker(ftk, "basic", "inverse", 1024)
    with pytest.raises(_native.FtkError) as e:
        klt.TrackFeatures(ref_levels[0], cur_levels[0], np.float32([[80, 60]]))
    assert e.value.code == -4


@pytest.mark.parametrize("model", MODELS)
def test_non_finite_and_huge_coordinates(ftk, oracle, model):
    """NaN / inf / absurd coordinates in ref_uv and in the prediction must neither fault nor differ from the CPU path."""
    ref_levels, cur_levels = scenes.scene(160, 120, 2)
    good = scenes.features(16, 160, 120, half=4, border_fraction=0.0)
    weird = np.float32([[np.nan, 50], [50, np.nan], [np.inf, 50], [50, -np.inf], [1e30, 1e30], [-1e30, 5], [3e9, 60], [-3e9, -3e9],
                        [2147483648.0, 10], [-2147483904.0, 10], [1e-40, 1e-40], [159.0, 119.0]])
    uv = np.concatenate([good, weird]).astype(np.float32)
    pred = uv.copy()
    pred[:8] = weird[:8]  # trackable ref points with poisoned predictions
    for method in METHODS:
        for cur_uv in (None, pred):
            gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=4, cur_uv=cur_uv)
            ok_g, uv_g, st_g, it_g = gpu
            ok_c, uv_c, st_c, it_c = cpu
            assert np.array_equal(st_g, st_c), f"{model}/{method}: {st_g} vs {st_c}"
            assert np.array_equal(uv_g.view(np.uint32), uv_c.view(np.uint32)) or np.array_equal(np.isnan(uv_g), np.isnan(uv_c)) and \
                np.array_equal(uv_g[~np.isnan(uv_g)], uv_c[~np.isnan(uv_c)]), f"{model}/{method}"
            assert np.array_equal(it_g, it_c)


def test_many_features_single_launch(ftk, oracle):
    """100 000 features in one call (grid = 100 000 workgroups); oracle-checked on a slice, size-independent checks on the rest."""
    ref_levels, cur_levels = scenes.scene(640, 480, 3)
    uv = scenes.features(100000, 640, 480, half=4, border_fraction=0.01)
    klt = make_tracker(ftk, "basic", "fast", 4, max_points=100000)
    ok, c, s = klt.TrackFeatures(ftk.ImagePyramid.from_host_levels(ref_levels), ftk.ImagePyramid.from_host_levels(cur_levels), uv)
    assert ok and (s == 1).mean() > 0.97
    d = c[s == 1] - uv[s == 1]
    assert abs(np.median(d[:, 0]) - 3.3) < 0.15 and abs(np.median(d[:, 1]) + 2.1) < 0.15
    # idempotent: the same call again gives the same bits; and a 2 000-feature slice matches the oracle
    ok2, c2, s2 = klt.TrackFeatures(ftk.ImagePyramid.from_host_levels(ref_levels), ftk.ImagePyramid.from_host_levels(cur_levels), uv)
    assert np.array_equal(c.view(np.uint32), c2.view(np.uint32)) and np.array_equal(s, s2)
    sl = slice(40000, 42000)
    okc, cc, sc, _ = oracle.klt_track_pyramid("basic", ref_levels, cur_levels, uv[sl], method="fast", half=4, max_points=100000)
    assert np.array_equal(c[sl].view(np.uint32), cc.view(np.uint32)) and np.array_equal(s[sl], sc)


def test_sharded_tracker_on_device_world_size_1(ftk, oracle):
    """ShardedKlt with the real device tracker (single rank): packed shard layout, gather and unpack on GPU tensors."""
    import torch
    from feature_tracker_amd import device as D
    from feature_tracker_amd import dist as FD
    ref_levels, cur_levels = scenes.scene(320, 240, 3)
    uv = scenes.features(333, 320, 240, half=5)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        opt = ftk.OpticalFlowOptions()
        opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = "inverse", 5, 5, 333
        klt = D.DeviceKlt("basic", opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
        sharded = FD.ShardedKlt(klt, 333, dev, 1, 0)
        d_ref = torch.from_numpy(uv).to(dev)
        guv, gst = sharded.track(d_ref, d_ref.clone(), torch.zeros(333, dtype=torch.uint8, device=dev))
        stream.synchronize()
    ok, c, s, _ = oracle.klt_track_pyramid("basic", ref_levels, cur_levels, uv, method="inverse", half=5, max_points=333)
    assert np.array_equal(guv.cpu().numpy().view(np.uint32), c.view(np.uint32)) and np.array_equal(gst.cpu().numpy(), s)


@pytest.mark.parametrize("group", [1, 2, 3, 4])
def test_one_wave_features_packed_into_workgroups(ftk, oracle, switch, group):
    """One wave per feature (what large batches use) with 1..4 features per workgroup — the compile-time SOLO instantiations of
    both kernels, incl. the chunked LSSD-fast level: no barrier, own LDS carve per wave, ragged last group.  Every variant must
    still equal the oracle bit for bit."""
    switch("FTK_KLT_WAVES", "1")
    switch("FTK_KLT_GROUP", str(group))
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "similarity")
    uv = scenes.features(301, 320, 240, half=6)  # not a multiple of any group size
    for model in MODELS:
        for method in METHODS:
            gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=6)
            assert_parity(gpu, cpu, f"group={group} {model}/{method}")
    gpu, cpu = run_pyramid(ftk, oracle, "lssd", "fast", ref_levels, cur_levels, uv, half=6, luminance=True)
    assert_parity(gpu, cpu, f"group={group} lssd/fast luminance")


@pytest.mark.parametrize("group", [None, 1, 3])
def test_affine_fast_one_wave_kernel_on_the_edge_cases(ftk, oracle, switch, group):
    """The affine tracker's `fast` method runs klt_fast_kernel<affine> only from 513 features on (a smaller call is its slowest
    feature, and that one is faster on the generic kernel's three waves: ftk_api.cpp fk_model), so the edge-case tests above — all
    of 64 - 400 features — reach it on the generic kernel.  The same cases at 520 - 700 features: border and outside features (clamped
    reference rows instead of the register-fed interior form, zero valid pixels), predictions, incoming failures and a cap that cuts
    a workgroup's group of features, a rectangular patch with tight options, pyramid levels smaller than the patch, the single-level
    overload with its affine prior.  Default packing, one feature per workgroup, and three (ragged last group)."""
    if group is not None:
        switch("FTK_KLT_GROUP", str(group))
    # border / outside
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "similarity")
    rs = np.random.RandomState(11)
    n = 700
    uv = np.empty((n, 2), np.float32)
    uv[:, 0] = rs.uniform(-12, 332, n)
    uv[:, 1] = rs.uniform(-12, 252, n)
    uv[:8] = [[0, 0], [319, 239], [0, 239], [319, 0], [0.5, 0.5], [318.5, 238.5], [-3, 100], [400, 400]]
    gpu, cpu = run_pyramid(ftk, oracle, "affine", "fast", ref_levels, cur_levels, uv, half=5)
    assert_parity(gpu, cpu, "affine/fast one-wave: border")
    # predictions, incoming status, a cap inside a group of four
    uv = scenes.features(601, 320, 240, half=6)
    pred = uv + np.float32([2.5, -1.5])
    status = (np.arange(601) % 5).astype(np.uint8)
    gpu, cpu = run_pyramid(ftk, oracle, "affine", "fast", ref_levels, cur_levels, uv, half=6, cur_uv=pred, status=status, max_points=533)
    assert_parity(gpu, cpu, "affine/fast one-wave: prediction, status, cap")
    skip = (status > 1) | (np.arange(601) >= 533)
    assert np.array_equal(gpu[1][skip], pred[skip]) and np.array_equal(gpu[2][skip], status[skip])
    # rectangular patch, tight options, hard motion
    hard_ref, hard_cur = scenes.scene(320, 240, 4, "hard", "similarity")
    uv = scenes.features(520, 320, 240, half=7, seed=3)
    gpu, cpu = run_pyramid(ftk, oracle, "affine", "fast", hard_ref, hard_cur, uv, half=3, half_cols=7, kMaxIteration=6, kMaxToleranceLargeStep=2,
                           kMaxConvergeStep=1e-3)
    assert_parity(gpu, cpu, "affine/fast one-wave: rectangular")
    gpu, cpu = run_pyramid(ftk, oracle, "affine", "fast", hard_ref, hard_cur, uv, half=7, half_cols=2)
    assert_parity(gpu, cpu, "affine/fast one-wave: rectangular, tall")
    # levels smaller than the patch footprint
    ref, cur = synth.make_image_pair(64, 48, (1.3, -0.8))
    tiny_ref, tiny_cur = synth.build_pyramid(ref, 3), synth.build_pyramid(cur, 3)
    uv = scenes.features(530, 64, 48, half=6, border_fraction=0.2)
    gpu, cpu = run_pyramid(ftk, oracle, "affine", "fast", tiny_ref, tiny_cur, uv, half=6)
    assert_parity(gpu, cpu, "affine/fast one-wave: tiny image")
    # the single-level overload: predict_affine is honoured only here (affine_klt.cpp:70)
    one_ref, one_cur = scenes.scene(320, 240, 1, "easy", "similarity")
    uv = scenes.features(600, 320, 240, half=6)
    pred = uv + np.float32([3.0, -2.0])
    prior = np.float32([[1.01, 0.02], [-0.02, 0.99]])
    klt = make_tracker(ftk, "affine", "fast", 6)
    klt.predict_affine = prior
    ok, c, s = klt.TrackFeatures(one_ref[0], one_cur[0], uv, pred, None)
    cpu = oracle.klt_track_single("affine", one_ref[0], one_cur[0], uv, pred, None, prior=prior, **oracle_kwargs("fast", 6))
    assert_parity((ok, c, s, klt.last_iterations), cpu, "affine/fast one-wave: single level with prior")


def test_lssd_fast_chunked_equals_the_unchunked_level(ftk, oracle, switch):
    """The chunked one-wave LSSD-fast level (64-pixel ring, sums in registers) against the oracle on border / hard-motion /
    rectangular-patch inputs, and against the plain level (FTK_LSSD_CHUNKED=0)."""
    switch("FTK_KLT_WAVES", "1")
    ref_levels, cur_levels = scenes.scene(320, 240, 4, "hard", "similarity")
    rs = np.random.RandomState(8)
    uv = np.stack([rs.uniform(-10, 330, 500), rs.uniform(-10, 250, 500)], axis=1).astype(np.float32)
    th = np.deg2rad(2.0)
    prior = np.float32([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]) * np.float32(1.3)  # columns longer than 1: samples leave the conservative window
    for half, half_cols in ((6, None), (3, 9), (10, None), (1, 1)):
        results = []
        for chunked in ("1", "0"):
            switch("FTK_LSSD_CHUNKED", chunked)
            gpu, cpu = run_pyramid(ftk, oracle, "lssd", "fast", ref_levels, cur_levels, uv, half=half, half_cols=half_cols, prior=prior)
            assert_parity(gpu, cpu, f"chunked={chunked} half={half}x{half_cols}")
            results.append(gpu)
        assert np.array_equal(results[0][1].view(np.uint32), results[1][1].view(np.uint32))


@pytest.mark.parametrize("model,method", [("affine", "inverse"), ("lssd", "fast"), ("basic", "inverse")])
def test_launch_order_from_the_previous_call_changes_nothing(ftk, oracle, model, method, switch):
    """From the third call with the same feature count on, the device entry launches the features longest-first by an earlier
    call's iteration counts (ftk_api.cpp; the sort runs in an extra workgroup of the launch in between, klt_common.h
    klt_order_block; calls of >= 4096 features).  Every call must return what the first one did — the oracle's answer — also
    when the history comes from DIFFERENT inputs (a stale predictor) and when some features are passed through (incoming
    status, kMaxTrackPointsNumber)."""
    import torch
    from feature_tracker_amd import device as D
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "similarity")
    n = 4600
    uv = synth.make_features(n, 320, 240, margin=20.0, border_fraction=0.03, half=5)
    other = uv[::-1].copy()
    status = (np.arange(n) % 11 == 0).astype(np.uint8) * 3
    cap = 4400
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = D.context_on_stream(stream, 0)
    opt = ftk.OpticalFlowOptions()
    opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, 5, 5, cap
    with torch.cuda.stream(stream):
        klt = D.DeviceKlt(model, opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
        d_st = torch.from_numpy(status).to(dev)

        def run(points):
            d_ref = torch.from_numpy(points).to(dev)
            d_out, d_so = torch.empty_like(d_ref), torch.empty(n, dtype=torch.uint8, device=dev)
            klt.track(d_ref, d_ref.clone(), d_st, d_out, d_so)
            stream.synchronize()
            return d_out.cpu().numpy(), d_so.cpu().numpy()

        first = run(uv)                      # list order (no history)
        again = [run(uv) for _ in range(3)]  # the third and fourth call go through the permutation
        run(other)                           # history now comes from other inputs
        run(other)
        stale = run(uv)
    ok, c, s, _ = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, uv, status, method=method, half=5, max_points=cap)
    assert (s == 1).sum() > n // 2
    for name, (g_uv, g_st) in (("first", first), ("second", again[0]), ("third", again[1]), ("fourth", again[2]), ("stale history", stale)):
        assert np.array_equal(g_st, s), name
        assert np.array_equal(g_uv.view(np.uint32), c.view(np.uint32)), name


@pytest.mark.parametrize("model,method", [("basic", "fast"), ("basic", "inverse"), ("lssd", "fast"), ("affine", "inverse"), ("affine", "fast")])
def test_position_keyed_launch_order_when_the_feature_count_changes(ftk, oracle, model, method):
    """A front end drops and re-detects features every frame, so consecutive calls rarely have the same feature count and never get
    the index-keyed launch order.  Such a call is ordered by what the LAST call left at its features' positions (two small launches
    in front of the tracker's: klt_kernels.hip klt_position_order_launch).  The order must be a permutation — every feature tracked
    exactly once, into output buffers that start poisoned — whatever the counts look like: a growing list, a shrinking one, a list
    tracked frame after frame (reference positions = the last call's results), a list with equal counts everywhere."""
    import torch
    from feature_tracker_amd import device as D
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "hard", "similarity")
    n = 4700
    uv = synth.make_features(n, 320, 240, margin=20.0, border_fraction=0.03, half=5)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = D.context_on_stream(stream, 0)
    opt = ftk.OpticalFlowOptions()
    opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, 5, 5, n
    with torch.cuda.stream(stream):
        klt = D.DeviceKlt(model, opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)

        def run(points):
            m = len(points)
            d_ref = torch.from_numpy(np.ascontiguousarray(points)).to(dev)
            d_out = torch.full((m, 2), float("nan"), dtype=torch.float32, device=dev)
            d_so = torch.full((m,), 0xEE, dtype=torch.uint8, device=dev)
            klt.track(d_ref, d_ref.clone(), torch.zeros(m, dtype=torch.uint8, device=dev), d_out, d_so)
            stream.synchronize()
            return d_out.cpu().numpy(), d_so.cpu().numpy()

        lists = [uv[:4200], uv[:4650], uv, uv[50:4500]]   # growing, growing, shrinking: every call has a count of its own
        got = [run(pts) for pts in lists]
        follow = got[2][0].copy()                         # frame after frame: track from where the last call ended
        follow[~np.isfinite(follow).all(axis=1)] = 1.0
        lists.append(follow[:4400])
        got.append(run(lists[-1]))
    for pts, (g_uv, g_st) in zip(lists, got):
        ok, c, s, _ = oracle.klt_track_pyramid(model, ref_levels, cur_levels, pts, pts, None, method=method, half=5, max_points=n)
        assert not (g_st == 0xEE).any(), "a feature was never processed"
        assert np.array_equal(g_st, s)
        same = (g_uv.view(np.uint32) == c.view(np.uint32)) | (np.isnan(g_uv) & np.isnan(c))
        assert same.all()


@pytest.mark.parametrize("n", [4603, 8192])
def test_flat_iteration_counts_give_a_spatial_xcd_major_launch_order(ftk, oracle, n, switch, tmp_path):
    """Without a tail in the iteration counts the launch order is by image region, dealt XCD-major (klt_common.h klt_order_block,
    xcd_major_slot): it must be a permutation for feature counts that do and do not fill the last workgroup, the slots of one XCD
    (workgroup index mod 8) must come in runs that each cover a compact part of the image, and the results stay the oracle's."""
    import torch
    from feature_tracker_amd import device as D
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "translation")
    uv = synth.make_features(n, 320, 240, margin=24.0, border_fraction=0.0, half=5)
    dump = tmp_path / "order.bin"
    switch("FTK_KLT_SCHED_DUMP", str(dump))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = D.context_on_stream(stream, 0)
    opt = ftk.OpticalFlowOptions()
    opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = "inverse", 5, 5, n
    with torch.cuda.stream(stream):
        klt = D.DeviceKlt("basic", opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
        d_ref = torch.from_numpy(uv).to(dev)
        d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
        outs = []
        for _ in range(4):
            d_out, d_so = torch.empty_like(d_ref), torch.empty(n, dtype=torch.uint8, device=dev)
            klt.track(d_ref, d_ref.clone(), d_st, d_out, d_so)
            stream.synchronize()
            outs.append((d_out.cpu().numpy(), d_so.cpu().numpy()))
    ok, c, s, _ = oracle.klt_track_pyramid("basic", ref_levels, cur_levels, uv, uv, np.zeros(n, np.uint8), method="inverse", half=5, max_points=n)
    for g_uv, g_st in outs:
        assert np.array_equal(g_st, s)
        assert np.array_equal(g_uv.view(np.uint32), c.view(np.uint32))
    d = np.fromfile(dump, dtype=np.int32)
    order, iters = d[:n], d[n:]
    assert np.array_equal(np.sort(order), np.arange(n)), "not a permutation"
    if np.array_equal(order, np.arange(n)):
        pytest.skip("the iteration counts of this scene have a tail: the order is by count, not by region")
    group, run = 4, 64  # one-wave features, four per workgroup (ftk_api.cpp default); kOrderRunGroups workgroups per run
    dealt = (n // group) // (8 * run) * (8 * run)
    assert dealt > 0
    w = np.arange(dealt)                    # workgroup index -> (XCD, position among that XCD's workgroups)
    run_of = (w % 8) + 8 * ((w // 8) // run)
    area_all = uv[:, 0].std() * uv[:, 1].std()
    areas = []
    for r in np.unique(run_of):
        slots = (w[run_of == r][:, None] * group + np.arange(group)[None, :]).ravel()
        pts = uv[order[slots]]
        areas.append(pts[:, 0].std() * pts[:, 1].std())
    assert np.median(areas) < 0.25 * area_all, (np.median(areas), area_all)


@pytest.mark.parametrize("n", [4603, 8192])
def test_one_position_buffer_updated_in_place_over_many_calls(ftk, oracle, n):
    """ref_uv == cur_uv_in == cur_uv_out (one position buffer updated in place: include/ftk.h allows the out tensors to alias the in
    tensors) on calls large enough to carry the launch-order sort block (>= 4096 features): the block's spatial order would read
    the reference positions in two passes while the feature workgroups overwrite them — such a call must fall back to
    an order that does not read them, and every one of five chained calls must return the oracle's answer for ITS inputs."""
    import torch
    from feature_tracker_amd import device as D
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "translation")
    uv = synth.make_features(n, 320, 240, margin=30.0, border_fraction=0.0, half=5)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = D.context_on_stream(stream, 0)
    opt = ftk.OpticalFlowOptions()
    opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = "inverse", 5, 5, n
    zeros = np.zeros(n, np.uint8)
    with torch.cuda.stream(stream):
        klt = D.DeviceKlt("basic", opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
        d_pos = torch.from_numpy(uv).to(dev)
        d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
        expect = uv
        for call in range(5):
            d_so = torch.empty(n, dtype=torch.uint8, device=dev)
            klt.track(d_pos, d_pos, d_st, d_pos, d_so)
            stream.synchronize()
            ok, c, s, _ = oracle.klt_track_pyramid("basic", ref_levels, cur_levels, expect, expect, zeros, method="inverse", half=5, max_points=n)
            assert np.array_equal(d_so.cpu().numpy(), s), f"call {call}"
            assert np.array_equal(d_pos.cpu().numpy().view(np.uint32), c.view(np.uint32)), f"call {call}"
            expect = c


def test_host_images_reach_the_pyramid_launch_through_the_pinned_slots(ftk, oracle):
    """ftk_pyramid_build / ftk_pyramid_update of a PAGEABLE host image: a CPU copy into one of two pinned slots, read by the pyramid
    launch, no synchronisation.  Back-to-back builds (a slot is reused by the third: its event gates the copy), growing and
    shrinking images (the slots are reallocated), buffers overwritten right after the call returns — every level == the oracle's."""
    rs = np.random.RandomState(7)
    pyramids, expected = [], []
    for (h, w, levels) in ((120, 160, 3), (1080, 1920, 4), (97, 131, 2), (1200, 1600, 5), (64, 64, 2), (480, 752, 4)):
        img = rs.randint(0, 256, size=(h, w), dtype=np.uint8)
        expected.append(oracle.create_pyramid(img.copy(), levels))  # (the oracle's level 0 is a view of its argument)
        pyramids.append(ftk.ImagePyramid.build(img, levels))
        img[:] = 0  # the caller's buffer is free as soon as the call returns
    for pyr, exp in zip(pyramids, expected):
        for i, level in enumerate(exp):
            assert np.array_equal(pyr.download_level(i), level), (level.shape, i)
    # the synchronous update entry takes the same route
    img = rs.randint(0, 256, size=(480, 752), dtype=np.uint8)
    exp = oracle.create_pyramid(img.copy(), 4)
    pyramids[-1].update(img)
    img[:] = 255
    for i, level in enumerate(exp):
        assert np.array_equal(pyramids[-1].download_level(i), level), i


def test_position_keyed_trades_of_launch_slots_process_every_feature_once(ftk, oracle, switch, tmp_path):
    """A list that is reshuffled between calls: the launch order (by list index) is stale, and early launch slots trade places with
    late ones whose POSITION predicts many iterations (klt_common.h sched_resolve_slot; multi-wave trackers, >= 4096 features).
    Trades must happen here, and every feature must come out exactly as the oracle has it — whichever slot ran it."""
    import torch
    from feature_tracker_amd import device as D
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "hard", "similarity")
    n = 4700
    uv = synth.make_features(n, 320, 240, margin=20.0, border_fraction=0.02, half=6)
    dump = tmp_path / "trades.txt"
    switch("FTK_KLT_SWAP_DUMP", str(dump))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = D.context_on_stream(stream, 0)
    opt = ftk.OpticalFlowOptions()
    opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = "inverse", 6, 6, n
    rs = np.random.RandomState(3)
    lists = [uv] + [np.ascontiguousarray(uv[rs.permutation(n)]) for _ in range(4)]
    results = []
    with torch.cuda.stream(stream):
        klt = D.DeviceKlt("affine", opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
        d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
        for pts in lists:
            d_ref = torch.from_numpy(pts).to(dev)
            d_out, d_so = torch.empty_like(d_ref), torch.empty(n, dtype=torch.uint8, device=dev)
            klt.track(d_ref, d_ref.clone(), d_st, d_out, d_so)
            stream.synchronize()
            results.append((d_out.cpu().numpy(), d_so.cpu().numpy()))
    for pts, (g_uv, g_st) in zip(lists, results):
        ok, c, s, it = oracle.klt_track_pyramid("affine", ref_levels, cur_levels, pts, pts, np.zeros(n, np.uint8), method="inverse", half=6, max_points=n)
        assert np.array_equal(g_st, s)
        assert np.array_equal(g_uv.view(np.uint32), c.view(np.uint32))
    assert int(it.max()) >= 20, "the scene has no long feature: nothing to trade"
    trades, own = (int(x) for x in dump.read_text().split())
    assert trades > 0, "no early slot traded places with a late one"


def _real_pair():
    import os
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = np.ascontiguousarray(np.array(Image.open(os.path.join(root, "tests", "data", "optical_flow", "ref_image.png")).convert("L"), dtype=np.uint8))
    cur = np.ascontiguousarray(np.array(Image.open(os.path.join(root, "tests", "data", "optical_flow", "cur_image.png")).convert("L"), dtype=np.uint8))
    return synth.build_pyramid(ref, 4), synth.build_pyramid(cur, 4)


@pytest.mark.parametrize("model,method", [(m, k) for m in MODELS for k in METHODS])
def test_real_pair_both_tail_classes_and_the_early_launch_order_change_nothing(ftk, oracle, model, method, switch):
    """Round 5's launch policies on the reference's example pair (features that never converge: the calls have a long tail): the
    kernels report each call's longest feature, a variant with long calls is looked up in the long-tail half of the wave-policy
    table (klt_wave_policy.inc) and gets the longest-first launch order from 1 024 features on.  Both halves of the table (pinned with
    FTK_KLT_TAIL_CLASS), the class the library detects by itself over repeated calls, and the order switched on and off must all
    return the oracle's bits — on a list whose features are passed through in places (incoming status, kMaxTrackPointsNumber)."""
    import torch
    from feature_tracker_amd import device as D
    ref_levels, cur_levels = _real_pair()
    n, cap, half = 1500, 1450, 6
    rows, cols = ref_levels[0].shape
    rs = np.random.RandomState(5)
    uv = np.stack([rs.uniform(30, cols - 30, n), rs.uniform(30, rows - 30, n)], axis=1).astype(np.float32)
    status = (np.arange(n) % 13 == 0).astype(np.uint8) * 2
    ok, c, s, _ = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, uv, status, method=method, half=half, max_points=cap)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    opt = ftk.OpticalFlowOptions()
    opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, half, half, cap
    for tail_class, sched_min in ((None, None), ("0", None), ("1", None), (None, "100000"), ("1", "1024")):
        switch("FTK_KLT_TAIL_CLASS", tail_class)
        switch("FTK_KLT_SCHED_MIN", sched_min)
        with torch.cuda.stream(stream):
            ctx = D.context_on_stream(stream, 0)
            klt = D.DeviceKlt(model, opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
            d_ref, d_st = torch.from_numpy(uv).to(dev), torch.from_numpy(status).to(dev)
            for call in range(5):  # the tail class and the launch order settle over the first calls: every one of them must be right
                d_out, d_so = torch.full_like(d_ref, -7.0), torch.full((n,), 9, dtype=torch.uint8, device=dev)
                klt.track(d_ref, d_ref.clone(), d_st, d_out, d_so)
                stream.synchronize()
                what = f"{model}/{method} tail class {tail_class} order from {sched_min}, call {call}"
                assert np.array_equal(d_so.cpu().numpy(), s), what
                assert np.array_equal(d_out.cpu().numpy().view(np.uint32), c.view(np.uint32)), what


def test_pair_arrays_at_an_odd_float_offset_are_refused(ftk):
    """The kernels move a feature's (u, v) as one 8-byte access (include/ftk.h): a device pair array that is only 4-byte aligned is an
    FTK_E_INVALID_ARGUMENT, not a misaligned 64-bit access on the device."""
    import torch
    from feature_tracker_amd import device as D
    from feature_tracker_amd._native import FtkError
    ref_levels, cur_levels = scenes.scene(320, 240, 3)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        opt = ftk.OpticalFlowOptions()
        opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = "inverse", 5, 5, 64
        klt = D.DeviceKlt("basic", opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
        flat = torch.zeros(2 * 64 + 2, dtype=torch.float32, device=dev)
        odd = flat[1:129].view(64, 2)  # starts one float into the allocation: 4-byte aligned only
        good = torch.full((64, 2), 100.0, dtype=torch.float32, device=dev)
        st, so = torch.zeros(64, dtype=torch.uint8, device=dev), torch.zeros(64, dtype=torch.uint8, device=dev)
        for args in ((odd, good, good.clone()), (good, odd, good.clone()), (good, good.clone(), odd)):
            with pytest.raises(FtkError) as err:
                klt.track(args[0], args[1], st, args[2], so)
            assert err.value.code == -1 and "8-byte aligned" in str(err.value)
        klt.track(good, good.clone(), st, good.clone(), so)  # the aligned call goes through
        stream.synchronize()


@pytest.mark.parametrize("model,method", [("lssd", "fast"), ("affine", "inverse"), ("lssd", "inverse")])
def test_position_keyed_order_below_4096_features_on_a_changing_list(ftk, oracle, model, method, switch):
    """With a long tail the launch order applies from 1 024 features on (round 5) — also the position-keyed one a call gets when the
    feature COUNT has just changed (a front end that drops and re-detects features): lists of 1 500, 1 400, 1 600, 1 450 features of
    the reference's pair, frame after frame on one tracker, every result the oracle's."""
    import torch
    from feature_tracker_amd import device as D
    ref_levels, cur_levels = _real_pair()
    rows, cols = ref_levels[0].shape
    rs = np.random.RandomState(11)
    switch("FTK_KLT_TAIL_CLASS", "1")
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    opt = ftk.OpticalFlowOptions()
    opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, 6, 6, 100000
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        klt = D.DeviceKlt(model, opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
        pool = np.stack([rs.uniform(30, cols - 30, 1700), rs.uniform(30, rows - 30, 1700)], axis=1).astype(np.float32)
        for n in (1500, 1400, 1600, 1450, 1450):
            uv = np.ascontiguousarray(pool[rs.permutation(1700)[:n]])
            d_ref = torch.from_numpy(uv).to(dev)
            d_out, d_so = torch.full_like(d_ref, -3.0), torch.full((n,), 7, dtype=torch.uint8, device=dev)
            klt.track(d_ref, d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev), d_out, d_so)
            stream.synchronize()
            ok, c, s, _ = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, method=method, half=6, max_points=100000)
            assert np.array_equal(d_so.cpu().numpy(), s), (model, method, n)
            assert np.array_equal(d_out.cpu().numpy().view(np.uint32), c.view(np.uint32)), (model, method, n)
