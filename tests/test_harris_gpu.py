"""GPU parity tests: device Harris response / detection against the oracle (bit-exact)."""
import os

import numpy as np
import pytest

from feature_tracker_amd import synth

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "optical_flow")


def images():
    from PIL import Image
    real = np.array(Image.open(os.path.join(DATA, "ref_image.png")))
    s1, _ = synth.make_image_pair(320, 240)
    s2, _ = synth.make_image_pair(97, 61)
    flat = np.full((64, 80), 33, np.uint8)
    return {"real": real, "synthetic": s1, "small_odd": s2, "flat": flat}


@pytest.mark.parametrize("name", ["real", "synthetic", "small_odd", "flat"])
def test_response_map_bit_exact(ftk, oracle, name):
    img = images()[name]
    got = ftk.FeaturePointHarrisDetector().response(img)
    exp = oracle.harris_response(img)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("name,max_n,dist,thr", [("real", 300, 25, 40.0), ("real", 50, 10, 1000.0), ("synthetic", 300, 20, 40.0),
                                                ("synthetic", 10000, 2, -5.0), ("small_odd", 100, 5, 1.0), ("flat", 100, 10, 40.0),
                                                ("synthetic", 20, 1, 100.0)])
def test_detection_matches_oracle(ftk, oracle, name, max_n, dist, thr):
    img = images()[name]
    det = ftk.FeaturePointHarrisDetector()
    det.options().kMinFeatureDistance, det.options().kMinValidResponse = dist, thr
    ok, uv = det.DetectGoodFeatures(img, max_n)
    exp = oracle.harris_detect(img, max_n, dist, thr)
    assert ok and uv.shape == exp.shape and np.array_equal(uv, exp)
    if name != "flat" and thr > 0:
        assert len(uv) > 0
        # suppression property: no two survivors closer than the minimum distance (Chebyshev)
        d = np.abs(uv[:, None, :] - uv[None, :, :]).max(axis=2) + np.eye(len(uv)) * 1e9
        assert d.min() >= dist


def test_detect_then_track_end_to_end(ftk, oracle):
    """The reference test flow (test_optical_flow.cpp:41-83) on its own example pair, entirely through the device:
    Harris -> pyramids -> BasicKlt fast; identical to the oracle pipeline."""
    from PIL import Image
    ref = np.array(Image.open(os.path.join(DATA, "ref_image.png")))
    cur = np.array(Image.open(os.path.join(DATA, "cur_image.png")))
    det = ftk.FeaturePointHarrisDetector()
    det.options().kMinFeatureDistance, det.options().kMinValidResponse = 25, 40.0
    ok, uv = det.DetectGoodFeatures(ref, 300)
    klt = ftk.OpticalFlowBasicKlt()
    ok, c, s = klt.TrackFeatures(ftk.ImagePyramid.build(ref, 4), ftk.ImagePyramid.build(cur, 4), uv)
    ouv = oracle.harris_detect(ref, 300, 25, 40.0)
    okc, oc, os_, _ = oracle.klt_track_pyramid("basic", oracle.create_pyramid(ref, 4), oracle.create_pyramid(cur, 4), ouv, method="fast", half=6)
    assert np.array_equal(uv, ouv) and np.array_equal(s, os_) and np.array_equal(c.view(np.uint32), oc.view(np.uint32))
    assert (s == 1).mean() > 0.8
