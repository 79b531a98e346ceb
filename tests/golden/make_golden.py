"""Generates the golden fixtures in this directory and replays them.

PROVENANCE: the reference (Horizon1026/Feature_Tracker) holds no golden vectors and cannot be built
in this image (un-vendored Slam_Utility / Eigen), so these fixtures are produced by THIS repo's
oracle (oracle/liboracle.so, gcc -O3 -ffp-contract=off, x86-64) — they are regression pins that make
compiler / platform drift of the oracle and of the HIP path visible, not reference pins.
Each .npz stores the complete inputs (images included) and the expected outputs.

    python -m tests.golden.make_golden        # regenerate (only when the oracle's definition changes)
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

MODELS = ["basic", "affine", "lssd"]
METHODS = ["inverse", "direct", "fast"]


def run_case(oracle, z):
    """Replays one fixture through the oracle binding; returns {name: array} of outputs."""
    kind = str(z["kind"])
    if kind in ("klt_pyramid", "klt_single"):
        levels = int(z["levels"])
        ref_levels = [z[f"ref{i}"] for i in range(levels)]
        cur_levels = [z[f"cur{i}"] for i in range(levels)]
        kw = dict(method=str(z["method"]), half=int(z["half"]), half_cols=int(z["half_cols"]), max_points=int(z["max_points"]),
                  prior=z["prior"], consider_luminance=bool(z["luminance"]))
        cur_uv = z["cur_uv"] if z["cur_uv"].size else None
        status = z["status"] if z["status"].size else None
        if kind == "klt_pyramid":
            ok, c, s, it = oracle.klt_track_pyramid(str(z["model"]), ref_levels, cur_levels, z["ref_uv"], cur_uv, status, **kw)
        else:
            ok, c, s, it = oracle.klt_track_single(str(z["model"]), ref_levels[0], cur_levels[0], z["ref_uv"], cur_uv, status, **kw)
        return {"uv": c, "status": s, "iters": it}
    if kind == "force":
        ok, idx = oracle.force_match(z["ref_bits"], z["cur_bits"], float(z["max_distance"]))
        return {"index": idx}
    if kind == "nearby":
        ok, idx = oracle.nearby_match(z["ref_bits"], z["cur_bits"], z["pred_uv"], z["cur_uv"], float(z["max_distance"]), int(z["max_col"]),
                                      int(z["max_row"]))
        return {"index": idx}
    if kind == "direct":
        levels = int(z["levels"])
        ok, c, q, p, st, it = oracle.direct_track([z[f"ref{i}"] for i in range(levels)], [z[f"cur{i}"] for i in range(levels)], z["K"], z["p_c_in_ref"],
                                                  z["ref_uv"], z["cur_uv"] if z["cur_uv"].size else None, z["q_rc"], z["p_rc"],
                                                  z["status"] if z["status"].size else None, half=int(z["half"]), max_points=int(z["max_points"]))
        return {"uv": c, "q": q, "p": p, "status": st, "iters": np.array([it], np.uint32)}
    if kind == "float_force":
        ok, idx = oracle.match_float(z["ref_desc"], z["cur_desc"], float(z["max_distance"]))
        return {"index": idx}
    if kind == "float_nearby":
        ok, idx = oracle.match_float(z["ref_desc"], z["cur_desc"], float(z["max_distance"]), z["pred_uv"], z["cur_uv"], int(z["max_col"]),
                                     int(z["max_row"]))
        return {"index": idx}
    raise ValueError(kind)


def _klt_case(name, model, method, kind, ref_levels, cur_levels, ref_uv, half, half_cols=None, cur_uv=None, status=None, prior=None,
              luminance=False, max_points=100000):
    d = dict(kind=kind, model=model, method=method, levels=len(ref_levels), half=half, half_cols=half if half_cols is None else half_cols,
             max_points=max_points, prior=np.eye(2, dtype=np.float32) if prior is None else np.asarray(prior, np.float32),
             luminance=luminance, ref_uv=np.asarray(ref_uv, np.float32),
             cur_uv=np.zeros((0, 2), np.float32) if cur_uv is None else np.asarray(cur_uv, np.float32),
             status=np.zeros(0, np.uint8) if status is None else np.asarray(status, np.uint8))
    for i, (r, c) in enumerate(zip(ref_levels, cur_levels)):
        d[f"ref{i}"] = r
        d[f"cur{i}"] = c
    return name, d


def generate(only=()):
    sys.path.insert(0, ROOT)
    from feature_tracker_amd import synth
    from tests import oracle_lib as oracle

    cases = []
    ref, cur = synth.make_image_pair(160, 120, (2.3, -1.6), rotation_deg=1.0, scale=1.01)
    ref_levels, cur_levels = synth.build_pyramid(ref, 3), synth.build_pyramid(cur, 3)
    rs = np.random.RandomState(2024)
    uv = np.stack([rs.uniform(-4, 164, 96), rs.uniform(-4, 124, 96)], axis=1).astype(np.float32)
    uv[:64] = synth.make_features(64, 160, 120, seed=9, margin=20.0, border_fraction=0.0)
    th = np.deg2rad(0.8)
    rot = np.float32([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    for model in MODELS:
        for method in METHODS:
            cases.append(_klt_case(f"klt_{model}_{method}", model, method, "klt_pyramid", ref_levels, cur_levels, uv, half=4))
    status = (np.arange(96) % 5).astype(np.uint8)
    cases.append(_klt_case("klt_basic_fast_pred_status_cap", "basic", "fast", "klt_pyramid", ref_levels, cur_levels, uv, half=3, half_cols=5,
                           cur_uv=uv + np.float32([1.5, -1.0]), status=status, max_points=80))
    cases.append(_klt_case("klt_lssd_fast_luminance_prior", "lssd", "fast", "klt_pyramid", ref_levels, cur_levels, uv, half=4, prior=rot,
                           luminance=True))
    cases.append(_klt_case("klt_affine_inverse_single_prior", "affine", "inverse", "klt_single", ref_levels[:1], cur_levels[:1], uv, half=4,
                           cur_uv=uv + np.float32([2.0, -1.5]), prior=[[1.01, 0.02], [-0.02, 0.99]]))
    cases.append(_klt_case("klt_lssd_direct_single", "lssd", "direct", "klt_single", ref_levels[:1], cur_levels[:1], uv, half=4,
                           cur_uv=uv + np.float32([2.0, -1.5]), prior=rot))

    bits_ref, bits_cur, _ = synth.make_descriptors(96, 130, n_bits=256, flips=25)
    bits_cur[7] = bits_cur[3]  # a tie
    cases.append(("match_force", dict(kind="force", ref_bits=bits_ref, cur_bits=bits_cur, max_distance=np.float32(60.0))))
    cuv = np.stack([rs.uniform(0, 160, 130), rs.uniform(0, 120, 130)], axis=1).astype(np.float32)
    puv = np.stack([rs.uniform(0, 160, 96), rs.uniform(0, 120, 96)], axis=1).astype(np.float32)
    cases.append(("match_nearby", dict(kind="nearby", ref_bits=bits_ref, cur_bits=bits_cur, pred_uv=puv, cur_uv=cuv,
                                       max_distance=np.float32(140.0), max_col=50, max_row=40)))

    # float descriptors (cosine distance), stored as fp16-exact values to keep the fixtures small
    fref, fcur, _ = synth.make_float_descriptors(96, 130, dim=128, noise=0.3)
    fref, fcur = fref.astype(np.float16).astype(np.float32), fcur.astype(np.float16).astype(np.float32)
    fcur[7] = fcur[3]  # a tie
    fcur[11] = 0.0     # irregular candidate
    cases.append(("match_float_force", dict(kind="float_force", ref_desc=fref.astype(np.float16), cur_desc=fcur.astype(np.float16),
                                            max_distance=np.float32(0.2))))
    cases.append(("match_float_nearby", dict(kind="float_nearby", ref_desc=fref.astype(np.float16), cur_desc=fcur.astype(np.float16), pred_uv=puv,
                                             cur_uv=cuv, max_distance=np.float32(0.6), max_col=50, max_row=40)))

    # NearbyMatch over enough candidates (>= 2 048) for the window-aware early exits of both matchers, features in raster
    # order (bands of 4 rows), NaN coordinates on both sides (a NaN passes every window test, descriptor_matcher.h:108-111)
    rs2 = np.random.RandomState(77)
    n_r, n_c = 600, 2304
    cuv2 = np.stack([rs2.uniform(0, 640, n_c), rs2.uniform(0, 480, n_c)], axis=1).astype(np.float32)
    puv2 = np.stack([rs2.uniform(0, 640, n_r), rs2.uniform(0, 480, n_r)], axis=1).astype(np.float32)
    oc = np.lexsort((cuv2[:, 0], np.floor(cuv2[:, 1] / 4)))
    orf = np.lexsort((puv2[:, 0], np.floor(puv2[:, 1] / 4)))
    b_ref, b_cur, _ = synth.make_descriptors(n_r, n_c, n_bits=64, flips=6, seed=5)
    f_ref, f_cur, _ = synth.make_float_descriptors(n_r, n_c, dim=32, noise=0.3, seed=6)
    cuv2, puv2 = np.ascontiguousarray(cuv2[oc]), np.ascontiguousarray(puv2[orf])
    cuv2[100, 0] = np.nan
    puv2[7] = np.nan
    cases.append(("match_nearby_raster", dict(kind="nearby", ref_bits=np.ascontiguousarray(b_ref[orf]), cur_bits=np.ascontiguousarray(b_cur[oc]),
                                              pred_uv=puv2, cur_uv=cuv2, max_distance=np.float32(40.0), max_col=60, max_row=25)))
    f_ref16 = np.ascontiguousarray(f_ref[orf]).astype(np.float16)
    f_cur16 = np.ascontiguousarray(f_cur[oc]).astype(np.float16)
    cases.append(("match_float_nearby_raster", dict(kind="float_nearby", ref_desc=f_ref16, cur_desc=f_cur16, pred_uv=puv2, cur_uv=cuv2,
                                                    max_distance=np.float32(0.7), max_col=60, max_row=25)))

    # DirectMethod: small scene, prediction + initial pose + status, a point behind the camera, a cap
    from tests import scenes as _scenes
    rl, cl = _scenes.scene(160, 120, 3, "easy", "similarity")
    duv = _scenes.features(60, 160, 120, half=4)
    dz = (3.0 + 0.02 * np.arange(60)).astype(np.float32)
    dK = np.float32([150.0, 152.0, 80.5, 60.25])
    dpts = np.stack([(duv[:, 0] - dK[2]) / dK[0] * dz, (duv[:, 1] - dK[3]) / dK[1] * dz, dz], axis=1).astype(np.float32)
    dpts[5, 2] = -1.0
    dcase = dict(kind="direct", levels=3, half=4, max_points=50, K=dK, p_c_in_ref=dpts, ref_uv=duv, cur_uv=(duv + np.float32([0.5, -0.25])).astype(np.float32),
                 q_rc=np.float32([0.9999, 0.002, -0.003, 0.001]), p_rc=np.float32([0.01, -0.02, 0.005]), status=(np.arange(60) % 4).astype(np.uint8))
    for i in range(3):
        dcase[f"ref{i}"], dcase[f"cur{i}"] = rl[i], cl[i]
    cases.append(("direct_method", dcase))

    for name, d in cases:
        if only and name not in only:
            continue
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        z = np.load(path)
        out = run_case(oracle, z)
        np.savez_compressed(path, **d, **{"out_" + k: v for k, v in out.items()})
        print(name, {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    generate(tuple(sys.argv[1:]))
