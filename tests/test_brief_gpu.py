"""GPU parity tests: device BRIEF descriptors (bit-packed) against the oracle, and the device-only
descriptor -> matcher pipeline."""
import os

import numpy as np
import pytest

from feature_tracker_amd import synth

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "optical_flow")


@pytest.mark.parametrize("n_bits,half", [(256, 8), (128, 4), (200, 12), (512, 15), (32, 1)])
def test_brief_bits_match_oracle(ftk, oracle, n_bits, half):
    img, _ = synth.make_image_pair(320, 240)
    rs = np.random.RandomState(n_bits)
    uv = np.stack([rs.uniform(-5, 325, 400), rs.uniform(-5, 245, 400)], axis=1).astype(np.float32)
    uv[:6] = [[half + 1, half + 1], [half + 0.49, half + 1], [319 - half - 1, 239 - half - 1], [319 - half - 0.5, 100], [np.nan, 5], [1e20, 5]]
    d = ftk.BriefDescriptor()
    d.options().kLength, d.options().kHalfPatchSize = n_bits, half
    ok, bits = d.Compute(img, uv)
    ok_c, bits_c = oracle.brief_compute(img, uv, n_bits, half)
    assert ok and ok_c
    assert np.array_equal(bits, bits_c)
    assert bits_c.any(axis=1).sum() > 200  # interior features carry information
    words = d.compute_packed(img, uv)
    assert np.array_equal(words, ftk.pack_brief(bits_c))
    assert np.array_equal(ftk.unpack_brief(words, n_bits), bits_c)


def test_descriptor_to_matcher_pipeline_on_device(ftk, oracle):
    """brief_compute_device -> hamming_match_device without a host hop, on the reference's example pair."""
    import torch
    from PIL import Image
    from feature_tracker_amd import device as D
    ref = np.array(Image.open(os.path.join(DATA, "ref_image.png")))
    cur = np.array(Image.open(os.path.join(DATA, "cur_image.png")))
    ref_uv = synth.make_features(300, 752, 480, seed=21, half=8, border_fraction=0.0)
    # features of the current frame: the reference features moved by the (oracle-)tracked flow, plus noise-free duplicates
    ok, cur_uv, st, _ = oracle.klt_track_pyramid("basic", synth.build_pyramid(ref, 4), synth.build_pyramid(cur, 4), ref_uv, method="fast", half=6)
    cur_uv = cur_uv[st == 1]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        rp, cp = D.upload_pyramid([ref], ctx, dev), D.upload_pyramid([cur], ctx, dev)
        d_ruv, d_cuv = torch.from_numpy(ref_uv).to(dev), torch.from_numpy(cur_uv).to(dev)
        d_rw = torch.zeros((len(ref_uv), 8), dtype=torch.int32, device=dev)
        d_cw = torch.zeros((len(cur_uv), 8), dtype=torch.int32, device=dev)
        D.brief_compute_device(ctx, rp, d_ruv, 256, 8, d_rw)
        D.brief_compute_device(ctx, cp, d_cuv, 256, 8, d_cw)
        d_idx = torch.full((len(ref_uv),), -1, dtype=torch.int32, device=dev)
        D.hamming_match_device(ctx, d_rw, d_cw, 256, 60.0, d_idx, pred_uv=d_ruv, cur_uv=d_cuv, max_col=50, max_row=50)
        stream.synchronize()
        idx = d_idx.cpu().numpy()
    _, rb = oracle.brief_compute(ref, ref_uv, 256, 8)
    _, cb = oracle.brief_compute(cur, cur_uv, 256, 8)
    okc, idx_c = oracle.nearby_match(rb, cb, ref_uv, cur_uv, 60.0, max_col=50, max_row=50)
    assert np.array_equal(idx, idx_c)
    # most tracked features are re-found by their descriptor
    tracked_ids = np.nonzero(st == 1)[0]
    hits = sum(1 for k, i in enumerate(tracked_ids) if idx_c[i] == k)
    assert hits > 0.6 * len(tracked_ids)
