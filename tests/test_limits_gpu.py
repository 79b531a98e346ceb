"""GPU parity tests for inputs that used to be refused as FTK_E_UNSUPPORTED (device limits must never change a result,
SURVEY.md section 8(b) "Errors") and for the threading guarantee of a shared context (include/ftk.h, Conventions)."""
import threading

import numpy as np
import pytest

from feature_tracker_amd import synth
from tests.test_klt_gpu import METHODS, MODELS, assert_parity, run_pyramid

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("size", [(16, 8), (9, 33), (5, 5), (2, 2)])
def test_pyramid_levels_down_to_one_row_or_column(ftk, oracle, model, size):
    """A deep pyramid of a small image ends in levels of 2x1 / 1x2 / 1x1 pixels.  The reference runs them like any other level
    (GetPixelValue's closed-rectangle rule leaves at most the border row valid) — so must the device path."""
    w, h = size
    ref, cur = synth.make_image_pair(max(w, 8) * 4, max(h, 8) * 4, (0.7, -0.4))
    ref, cur = np.ascontiguousarray(ref[:h, :w]), np.ascontiguousarray(cur[:h, :w])
    levels = 1
    while (w >> levels) >= 1 and (h >> levels) >= 1:
        levels += 1
    ref_levels, cur_levels = synth.build_pyramid(ref, levels), synth.build_pyramid(cur, levels)
    assert min(ref_levels[-1].shape) == 1
    rs = np.random.RandomState(w * 100 + h)
    uv = np.stack([rs.uniform(-1, w, 40), rs.uniform(-1, h, 40)], axis=1).astype(np.float32)
    uv[:4] = [[0, 0], [w - 1, h - 1], [0.5, 0.5], [w / 2, h / 2]]
    for method in METHODS:
        for half in (1, 3):
            gpu, cpu = run_pyramid(ftk, oracle, model, method, ref_levels, cur_levels, uv, half=half)
            assert_parity(gpu, cpu, f"{w}x{h} {model}/{method} half={half}")


@pytest.mark.parametrize("n_bits", [65, 96, 150, 333, 520, 1000, 2048])
def test_descriptor_widths_that_are_not_a_power_of_two_words(ftk, oracle, n_bits):
    """3, 5, 11, 17, 32, 64 words per descriptor: padded or generic scan, same indices as the scalar loop (host-buffer entry)."""
    ref, cur, _ = synth.make_descriptors(300, 260, n_bits=n_bits, flips=max(1, n_bits // 13))
    rs = np.random.RandomState(n_bits)
    cur_uv = rs.uniform(0, 200, (260, 2)).astype(np.float32)
    pred_uv = rs.uniform(0, 200, (300, 2)).astype(np.float32)
    m = ftk.BriefMatcher()
    m.options().kMaxValidDescriptorDistance = n_bits / 4.0
    m.options().kMaxValidPredictColDistance = m.options().kMaxValidPredictRowDistance = 60
    ok, idx = m.ForceMatch(ref, cur)
    ok_c, idx_c = oracle.force_match(ref, cur, n_bits / 4.0)
    assert ok and np.array_equal(idx, idx_c) and (idx_c >= 0).sum() > 100
    ok, idx = m.NearbyMatch(ref, cur, pred_uv, cur_uv)
    ok_c, idx_c = oracle.nearby_match(ref, cur, pred_uv, cur_uv, n_bits / 4.0, 60, 60)
    assert ok and np.array_equal(idx, idx_c)


@pytest.mark.parametrize("n_words", [3, 5, 12, 20])
def test_device_entry_accepts_any_descriptor_width(ftk, oracle, n_words):
    """ftk_hamming_match_device on device-resident words of a width no register-tiled kernel exists for."""
    import torch
    from feature_tracker_amd import device as D
    n_bits = 32 * n_words - 7
    ref, cur, _ = synth.make_descriptors(500, 400, n_bits=n_bits, flips=n_bits // 12)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        d_ref = torch.from_numpy(ftk.pack_brief(ref).view(np.int32)).to(dev)
        d_cur = torch.from_numpy(ftk.pack_brief(cur).view(np.int32)).to(dev)
        assert d_ref.shape[1] == n_words
        d_idx = torch.full((500,), -1, dtype=torch.int32, device=dev)
        D.hamming_match_device(ctx, d_ref, d_cur, n_bits, n_bits / 4.0, d_idx)
        stream.synchronize()
    ok_c, idx_c = oracle.force_match(ref, cur, n_bits / 4.0)
    assert np.array_equal(d_idx.cpu().numpy(), idx_c)


def test_direct_method_with_more_features_than_fit_in_lds(ftk, oracle):
    """3 500 tracked features in ONE pose problem (the projection table moves from LDS to device memory): pose, pixels,
    status and iteration count still those of the scalar loop."""
    from tests.test_direct_method_gpu import assert_identical, run_both, scene
    rl, cl, uv, pts = scene(levels=2, n=3500, half=1)
    g, c = run_both(ftk, oracle, rl, cl, uv, pts, max_points=3500, half=1, max_iteration=6)
    assert_identical(g, c)


def test_levels_beyond_32_bit_pixel_offsets_are_refused(ftk, gpu_ctx):
    """include/ftk.h, image pyramids: a level of 2^24 rows / columns or 2^32 pixels cannot be addressed by the trackers' 32-bit
    offsets; the pyramid calls say so (FTK_E_UNSUPPORTED) instead of tracking on wrapped addresses.  Nothing is read at wrap time,
    so a small allocation stands in for the image."""
    import torch
    from feature_tracker_amd._native import FtkError
    t = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    for rows, cols in (((1 << 24), 8), (8, (1 << 24)), (1 << 16, 1 << 16)):
        with pytest.raises(FtkError) as e:
            ftk.ImagePyramid.from_device_levels([(t.data_ptr(), rows, cols)], gpu_ctx, keepalive=[t])
        assert e.value.code == -4, e.value
    ok = ftk.ImagePyramid.from_device_levels([(t.data_ptr(), 64, 64)], gpu_ctx, keepalive=[t])
    assert ok is not None


def test_two_threads_share_one_context(ftk, oracle):
    """Separate tracker / matcher objects on the process-wide context, driven from two threads at once (a stereo front end):
    calls are serialised inside the C ABI, so every result equals the single-threaded one."""
    from tests import scenes
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "similarity")
    jobs = []
    for k, (model, method, n) in enumerate([("basic", "inverse", 900), ("lssd", "fast", 20000), ("affine", "inverse", 300), ("basic", "fast", 17000)]):
        uv = scenes.features(n, 320, 240, half=5, seed=50 + k)
        cpu = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, method=method, half=5, max_points=n)
        jobs.append((model, method, uv, cpu))
    ref_bits, cur_bits, _ = synth.make_descriptors(3000, 2500)
    ok_c, idx_c = oracle.force_match(ref_bits, cur_bits, 60.0)
    rp, cp = ftk.ImagePyramid.from_host_levels(ref_levels), ftk.ImagePyramid.from_host_levels(cur_levels)
    errors = []

    def track_loop(which):
        from tests.test_klt_gpu import make_tracker
        try:
            for rep in range(6):
                for model, method, uv, cpu in jobs[which::2]:
                    klt = make_tracker(ftk, model, method, 5, max_points=len(uv))
                    ok, c, s = klt.TrackFeatures(rp, cp, uv)
                    assert_parity((ok, c, s, klt.last_iterations), cpu, f"thread {which} {model}/{method}")
                m = ftk.BriefMatcher()
                m.options().kMaxValidDescriptorDistance = 60
                ok, idx = m.ForceMatch(ref_bits, cur_bits)
                assert ok and np.array_equal(idx, idx_c)
        except BaseException as exc:  # surfaced in the main thread
            errors.append(exc)

    threads = [threading.Thread(target=track_loop, args=(w,)) for w in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]
