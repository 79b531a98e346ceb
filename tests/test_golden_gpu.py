"""The HIP path against the committed golden fixtures (tests/golden/*.npz): bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CLASSES = {"basic": "OpticalFlowBasicKlt", "affine": "OpticalFlowAffineKlt", "lssd": "OpticalFlowLssdKlt"}
CASES = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))


@pytest.mark.parametrize("name", CASES)
def test_hip_reproduces_golden(ftk, name):
    z = np.load(os.path.join(GOLDEN, name))
    kind = str(z["kind"])
    if kind.startswith("klt"):
        model = str(z["model"])
        klt = getattr(ftk, CLASSES[model])()
        o = klt.options()
        o.kMethod, o.kPatchRowHalfSize, o.kPatchColHalfSize, o.kMaxTrackPointsNumber = str(z["method"]), int(z["half"]), int(z["half_cols"]), int(z["max_points"])
        if model == "affine":
            klt.predict_affine = z["prior"]
        if model == "lssd":
            klt.predict_R_cr = z["prior"]
            klt.consider_patch_luminance = bool(z["luminance"])
        levels = int(z["levels"])
        cur_uv = z["cur_uv"] if z["cur_uv"].size else None
        status = z["status"] if z["status"].size else None
        if kind == "klt_pyramid":
            ref = ftk.ImagePyramid.from_host_levels([z[f"ref{i}"] for i in range(levels)])
            cur = ftk.ImagePyramid.from_host_levels([z[f"cur{i}"] for i in range(levels)])
        else:
            ref, cur = z["ref0"], z["cur0"]
        ok, c, s = klt.TrackFeatures(ref, cur, z["ref_uv"], cur_uv, status)
        assert ok
        assert np.array_equal(s, z["out_status"])
        assert np.abs(c.astype(np.float64) - z["out_uv"].astype(np.float64))[np.isfinite(z["out_uv"])].max() <= 1e-3  # north_star tolerance
        assert np.array_equal(c.view(np.uint32), z["out_uv"].view(np.uint32))  # design goal: bit-identical
        assert np.array_equal(klt.last_iterations, z["out_iters"])
    elif kind == "direct":
        levels = int(z["levels"])
        dm = ftk.DirectMethod()
        dm.options().kPatchRowHalfSize = dm.options().kPatchColHalfSize = int(z["half"])
        dm.options().kMaxTrackPointsNumber = int(z["max_points"])
        ok, c, q, p, st = dm.TrackFeatures(ftk.ImagePyramid.from_host_levels([z[f"ref{i}"] for i in range(levels)]),
                                           ftk.ImagePyramid.from_host_levels([z[f"cur{i}"] for i in range(levels)]), z["K"], z["p_c_in_ref"], z["ref_uv"],
                                           z["cur_uv"], z["q_rc"], z["p_rc"], z["status"])
        assert ok and dm.last_iterations == int(z["out_iters"][0])
        assert np.array_equal(st, z["out_status"])
        assert np.abs(c.astype(np.float64) - z["out_uv"].astype(np.float64)).max() <= 1e-3  # north_star tolerance
        assert np.array_equal(c.view(np.uint32), z["out_uv"].view(np.uint32))
        assert np.array_equal(q.view(np.uint32), z["out_q"].view(np.uint32)) and np.array_equal(p.view(np.uint32), z["out_p"].view(np.uint32))
    elif kind in ("float_force", "float_nearby"):
        m = ftk.CosineMatcher()
        m.options().kMaxValidDescriptorDistance = float(z["max_distance"])
        if kind == "float_force":
            ok, idx = m.ForceMatch(z["ref_desc"], z["cur_desc"])
        else:
            m.options().kMaxValidPredictColDistance = int(z["max_col"])
            m.options().kMaxValidPredictRowDistance = int(z["max_row"])
            ok, idx = m.NearbyMatch(z["ref_desc"], z["cur_desc"], z["pred_uv"], z["cur_uv"])
        assert ok and np.array_equal(idx, z["out_index"])
    else:
        m = ftk.BriefMatcher()
        m.options().kMaxValidDescriptorDistance = float(z["max_distance"])
        if kind == "force":
            ok, idx = m.ForceMatch(z["ref_bits"], z["cur_bits"])
        else:
            m.options().kMaxValidPredictColDistance = int(z["max_col"])
            m.options().kMaxValidPredictRowDistance = int(z["max_row"])
            ok, idx = m.NearbyMatch(z["ref_bits"], z["cur_bits"], z["pred_uv"], z["cur_uv"])
        assert ok and np.array_equal(idx, z["out_index"])
