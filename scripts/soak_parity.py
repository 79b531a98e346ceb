#!/usr/bin/env python3
"""Randomised parity soak (GPU box): the HIP path against the CPU oracle on randomly drawn problems for a fixed
time budget — every tracker variant, both descriptor matchers, the direct method — comparing bit for bit.

    python scripts/soak_parity.py [seconds] [seed] [family]   prints one summary line per family, exits 1 on any mismatch
                                                              family: all (default) | matcher (both matchers only, larger sets)
                                                              | direct (batches of 1 .. 80 pose problems through the device
                                                                batch entry: the spread kernel with 32 .. 2 producer workgroups
                                                                per problem, and one workgroup per problem beyond what fits)
                                                              | tree (REPORTING mode: the trackers in the throughput mode,
                                                                ftk_set_reduction_mode(TREE) — per-variant px-error statistics
                                                                against the oracle, nothing asserted, exit code 0)

Random axes: image size (odd sizes included), motion (translation / rotation / scale, up to ~15 px), pyramid depth,
patch size (rectangular too), feature count, border features, predictions, incoming status, kMaxTrackPointsNumber,
descriptor lengths / set sizes / thresholds / windows, duplicated and zero descriptors, pose priors and depths.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import feature_tracker_amd as F  # noqa: E402
from feature_tracker_amd import synth  # noqa: E402
from tests import oracle_lib as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
ONLY = sys.argv[3] if len(sys.argv) > 3 else "all"
CLASSES = {"basic": F.OpticalFlowBasicKlt, "affine": F.OpticalFlowAffineKlt, "lssd": F.OpticalFlowLssdKlt}
stats = {}
tree_stats = {}
if ONLY == "tree":
    F.default_context().set_reduction("tree")  # the trackers of this process share the default context


def note(family, ok, detail=""):
    s = stats.setdefault(family, [0, 0, ""])
    s[0] += 1
    if not ok:
        s[1] += 1
        s[2] = s[2] or detail


def random_scene():
    w, h = int(rs.choice([160, 199, 320, 333, 640])), int(rs.choice([120, 151, 240, 255, 480]))
    t = (float(rs.uniform(-15, 15)), float(rs.uniform(-12, 12)))
    rot, sc = (0.0, 1.0) if rs.rand() < 0.4 else (float(rs.uniform(-3, 3)), float(rs.uniform(0.97, 1.03)))
    ref, cur = synth.make_image_pair(w, h, t, rotation_deg=rot, scale=sc)
    if rs.rand() < 0.1:
        cur[:, : w // 3] = 128  # a textureless band
    levels = int(rs.randint(1, 6))
    while (min(w, h) >> (levels - 1)) < 8:
        levels -= 1
    return w, h, levels, synth.build_pyramid(ref, levels), synth.build_pyramid(cur, levels), t


def klt_round():
    w, h, levels, rl, cl, t = random_scene()
    n = int(rs.choice([1, 7, 64, 300, 1000, 2500]))
    # half of the draws on the patch sizes that have compile-time instantiations (5, 6, 10: klt_fill_geometry), half anywhere in [1, 10]
    half, half_c = (int(rs.choice([5, 6, 10])) if rs.rand() < 0.5 else int(rs.randint(1, 11))), None
    if rs.rand() < 0.06:  # now and then a patch beyond a workgroup's LDS: the large-patch form (few features: the oracle walks every pixel)
        half, n = int(rs.choice([19, 24, 33, 45])), int(rs.choice([1, 7, 24]))
    if rs.rand() < 0.2:
        half_c = int(rs.randint(1, 11))
    uv = synth.make_features(n, w, h, seed=int(rs.randint(1 << 30)), margin=min(40.0, w / 8.0), border_fraction=float(rs.choice([0.0, 0.05, 0.3])), half=half)
    pred = status = None
    if rs.rand() < 0.4:
        pred = (uv + np.float32(t) * np.float32(rs.uniform(0.5, 1.2)) + rs.uniform(-1, 1, uv.shape)).astype(np.float32)
    if rs.rand() < 0.3:
        status = rs.randint(0, 5, n).astype(np.uint8)
    max_points = int(rs.choice([n, max(1, n // 2), 500]))
    rp, cp = F.ImagePyramid.from_host_levels(rl), F.ImagePyramid.from_host_levels(cl)
    for model in ("basic", "affine", "lssd"):
        for method in ("inverse", "direct", "fast"):
            klt = CLASSES[model]()
            o = klt.options()
            o.kMethod, o.kPatchRowHalfSize, o.kPatchColHalfSize, o.kMaxTrackPointsNumber = method, half, half if half_c is None else half_c, max_points
            lum = model == "lssd" and rs.rand() < 0.5
            prior = None
            if model == "lssd":
                klt.consider_patch_luminance = lum
                if rs.rand() < 0.3:
                    a = float(rs.uniform(-0.05, 0.05))
                    prior = np.float32([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
                    klt.predict_R_cr = prior
            ok, c, s = klt.TrackFeatures(rp, cp, uv, pred, status)
            ok2, oc, os_, oit = O.klt_track_pyramid(model, rl, cl, uv, pred, status, prior=prior, consider_luminance=lum, method=method, half=half,
                                                    half_cols=half_c, max_points=max_points)
            if ONLY == "tree":
                fin = np.isfinite(oc).all(axis=1) & np.isfinite(c).all(axis=1)
                d = np.linalg.norm(c[fin].astype(np.float64) - oc[fin].astype(np.float64), axis=1)
                t = tree_stats.setdefault(f"klt/{model}/{method}", {"n": 0, "gt": 0, "max": 0.0, "status": 0, "nonfinite": 0, "identical": 0, "d": []})
                t["n"] += int(fin.sum())
                t["gt"] += int((d > 1e-3).sum())
                t["max"] = max(t["max"], float(d.max()) if d.size else 0.0)
                t["status"] += int((s != os_).sum())
                t["nonfinite"] += int((~fin).sum() - (~np.isfinite(oc).all(axis=1)).sum())
                t["identical"] += int((c.view(np.uint32) == oc.view(np.uint32)).all(axis=1).sum())
                if len(t["d"]) < 400000:
                    t["d"].extend(d.tolist())
                continue
            same = ok == ok2 and np.array_equal(s, os_) and np.array_equal(c.view(np.uint32), oc.view(np.uint32)) and np.array_equal(klt.last_iterations, oit)
            note(f"klt/{model}/{method}", same, f"{w}x{h} L{levels} h{half}/{half_c} n{n} cap{max_points}")


def matcher_round():
    big = ONLY == "matcher"
    n_ref, n_cur = int(rs.choice([1, 33, 300, 1500] + ([700, 4000] if big else []))), int(rs.choice([1, 40, 257, 2000] + ([129, 2500, 5000] if big else [])))
    n_bits = int(rs.choice([32, 64, 200, 256, 512]))
    ref, cur, _ = synth.make_descriptors(n_ref, n_cur, n_bits=n_bits, flips=max(1, n_bits // int(rs.choice([8, 13, 30]))), seed=int(rs.randint(1 << 30)))
    if n_cur > 4:
        cur[rs.randint(n_cur)] = cur[rs.randint(n_cur)]
    thr = float(rs.choice([0.0, 5.0, 30.0, 60.0, 1000.0]))
    cuv = rs.uniform(0, 500, (n_cur, 2)).astype(np.float32)
    puv = rs.uniform(0, 500, (n_ref, 2)).astype(np.float32)
    win = int(rs.choice([5, 50, 600]))
    if rs.rand() < 0.5 and n_cur > 1:  # features in raster order (what the matchers' window-aware skipping keys on), sometimes with a NaN
        oc = np.lexsort((cuv[:, 0], np.floor(cuv[:, 1] / 4)))
        cur, cuv = np.ascontiguousarray(cur[oc]), np.ascontiguousarray(cuv[oc])
        orf = np.lexsort((puv[:, 0], np.floor(puv[:, 1] / 4)))
        ref, puv = np.ascontiguousarray(ref[orf]), np.ascontiguousarray(puv[orf])
        raster = (oc, orf)
        if rs.rand() < 0.2:
            cuv[rs.randint(n_cur), rs.randint(2)] = np.nan
        if rs.rand() < 0.2:
            puv[rs.randint(n_ref), rs.randint(2)] = np.nan
    else:
        raster = None
    m = F.BriefMatcher()
    m.options().kMaxValidDescriptorDistance, m.options().kMaxValidPredictColDistance, m.options().kMaxValidPredictRowDistance = thr, win, win // 2 + 1
    ok, idx = m.ForceMatch(ref, cur)
    note("match/hamming/force", np.array_equal(idx, O.force_match(ref, cur, thr)[1]), f"{n_ref}x{n_cur}x{n_bits} thr{thr}")
    ok, idx = m.NearbyMatch(ref, cur, puv, cuv)
    note("match/hamming/nearby", np.array_equal(idx, O.nearby_match(ref, cur, puv, cuv, thr, win, win // 2 + 1)[1]), f"{n_ref}x{n_cur}x{n_bits} thr{thr} w{win}")
    dim = int(rs.choice([3, 32, 64, 100, 128, 192, 200, 256, 320]))
    fr, fc, _ = synth.make_float_descriptors(n_ref, n_cur, dim=dim, noise=float(rs.choice([0.1, 0.3, 1.0])), seed=int(rs.randint(1 << 30)), normalize=bool(rs.rand() < 0.5))
    if raster is not None:
        fc, fr = np.ascontiguousarray(fc[raster[0]]), np.ascontiguousarray(fr[raster[1]])
    if n_cur > 4:
        fc[rs.randint(n_cur)] = fc[rs.randint(n_cur)]
        if rs.rand() < 0.3:
            fc[rs.randint(n_cur)] = 0
        if rs.rand() < 0.4:  # runs of near-identical neighbours: several scores inside the shortlist margin in one tile
            start, run = int(rs.randint(n_cur - 4)), int(rs.randint(2, min(40, n_cur - 4) + 1))
            run = min(run, n_cur - start)
            fc[start:start + run] = fc[start] * (1.0 + 1e-4 * rs.standard_normal((run, dim))).astype(np.float32)
    fthr = float(rs.choice([0.0, 0.05, 0.3, 0.6, 2.0]))
    c = F.CosineMatcher()
    c.options().kMaxValidDescriptorDistance, c.options().kMaxValidPredictColDistance, c.options().kMaxValidPredictRowDistance = fthr, win, win // 2 + 1
    with np.errstate(all="ignore"):
        ok, idx = c.ForceMatch(fr, fc)
        note("match/cosine/force", np.array_equal(idx, O.match_float(fr, fc, fthr)[1]), f"{n_ref}x{n_cur}x{dim} thr{fthr}")
        ok, idx = c.NearbyMatch(fr, fc, puv, cuv)
        note("match/cosine/nearby", np.array_equal(idx, O.match_float(fr, fc, fthr, puv, cuv, win, win // 2 + 1)[1]), f"{n_ref}x{n_cur}x{dim} thr{fthr} w{win}")


def direct_round():
    w, h, levels, rl, cl, t = random_scene()
    n = int(rs.choice([1, 20, 150, 400]))
    half = int(rs.randint(1, 8))
    uv = synth.make_features(n, w, h, seed=int(rs.randint(1 << 30)), margin=min(40.0, w / 8.0), border_fraction=0.05, half=half)
    fx, fy, cx, cy = float(rs.uniform(200, 700)), float(rs.uniform(200, 700)), w / 2 + float(rs.uniform(-9, 9)), h / 2 + float(rs.uniform(-9, 9))
    z = rs.uniform(2, 30, n).astype(np.float32)
    if n > 3:
        z[rs.randint(n)] = -1.0
    pts = np.stack([(uv[:, 0] - cx) / fx * z, (uv[:, 1] - cy) / fy * z, z], axis=1).astype(np.float32)
    q0 = np.float32([1, 0, 0, 0]) if rs.rand() < 0.5 else (np.float32([1, *rs.uniform(-0.01, 0.01, 3)]))
    p0 = np.zeros(3, np.float32) if rs.rand() < 0.5 else rs.uniform(-0.05, 0.05, 3).astype(np.float32)
    cap = int(rs.choice([n, max(1, n // 2), 500]))
    dm = F.DirectMethod()
    o = dm.options()
    o.kMaxTrackPointsNumber, o.kPatchRowHalfSize, o.kPatchColHalfSize = cap, half, half
    K = [fx, fy, cx, cy]
    with np.errstate(all="ignore"):
        ok, c, q, p, s = dm.TrackFeatures(F.ImagePyramid.from_host_levels(rl), F.ImagePyramid.from_host_levels(cl), K, pts, uv, None, q0, p0)
        ok2, oc, oq, op, os_, oit = O.direct_track(rl, cl, K, pts, uv, None, q0, p0, half=half, max_points=cap)
    if ONLY == "tree":
        t = tree_stats.setdefault("direct (pixels)", {"n": 0, "gt": 0, "max": 0.0, "status": 0, "nonfinite": 0, "identical": 0, "d": []})
        fin = np.isfinite(oc).all(axis=1) & np.isfinite(c).all(axis=1)
        d = np.linalg.norm(c[fin].astype(np.float64) - oc[fin].astype(np.float64), axis=1)
        t["n"] += int(fin.sum())
        t["gt"] += int((d > 1e-3).sum())
        t["max"] = max(t["max"], float(d.max()) if d.size else 0.0)
        t["status"] += int((s != os_).sum())
        t["identical"] += int((c.view(np.uint32) == oc.view(np.uint32)).all(axis=1).sum())
        if len(t["d"]) < 400000:
            t["d"].extend(d.tolist())
        return
    same = (ok == ok2 and dm.last_iterations == oit and np.array_equal(s, os_) and np.array_equal(c.view(np.uint32), oc.view(np.uint32)) and
            np.array_equal(q.view(np.uint32), oq.view(np.uint32)) and np.array_equal(p.view(np.uint32), op.view(np.uint32)))
    note("direct", same, f"{w}x{h} L{levels} h{half} n{n} cap{cap}")


def direct_batch_round():
    """One launch of a random batch of pose problems (ftk_direct_track_batch_device); every problem against the oracle run on it alone."""
    import torch
    from feature_tracker_amd import device as D
    w, h = int(rs.choice([320, 640])), int(rs.choice([240, 480]))
    t = (float(rs.uniform(-6, 6)), float(rs.uniform(-5, 5)))
    ref, cur = synth.make_image_pair(w, h, t)
    levels = int(rs.randint(2, 5))
    rl, cl = synth.build_pyramid(ref, levels), synth.build_pyramid(cur, levels)
    count = int(rs.choice([1, 2, 5, 6, 7, 12, 24, 40, 70, 76, 80]))
    half = int(rs.choice([5, 6, 7]))
    fx, fy, cx, cy = float(rs.uniform(300, 600)), float(rs.uniform(300, 600)), w / 2 + float(rs.uniform(-5, 5)), h / 2 + float(rs.uniform(-5, 5))
    K = [fx, fy, cx, cy]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        rp, cp = D.upload_pyramid(rl, ctx, dev), D.upload_pyramid(cl, ctx, dev)
        problems, host = [], []
        for k in range(count):
            n = int(rs.choice([0, 1, 60, 110, 150, 220, 300])) if rs.rand() < 0.3 else int(rs.randint(100, 301))
            uv = synth.make_features(max(n, 1), w, h, seed=int(rs.randint(1 << 30)), margin=min(40.0, w / 8.0), half=half)[:n]
            z = rs.uniform(2, 30, n).astype(np.float32)
            pts = np.stack([(uv[:, 0] - cx) / fx * z, (uv[:, 1] - cy) / fy * z, z], axis=1).astype(np.float32).reshape(-1, 3)
            q0 = np.float32([1, 0, 0, 0]) if rs.rand() < 0.5 else np.float32([1, *rs.uniform(-0.01, 0.01, 3)])
            p0 = np.zeros(3, np.float32) if rs.rand() < 0.5 else rs.uniform(-0.05, 0.05, 3).astype(np.float32)
            host.append((uv, pts, q0, p0))
            problems.append(dict(ref=rp, cur=cp, K=K, p_c_in_ref=torch.from_numpy(pts).to(dev).reshape(-1, 3), ref_uv=torch.from_numpy(uv).to(dev).reshape(-1, 2),
                                 cur_uv=torch.from_numpy(uv.copy()).to(dev).reshape(-1, 2), pose=torch.from_numpy(np.concatenate([q0, p0]).astype(np.float32)).to(dev),
                                 status=torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)[:n], status_valid=False,
                                 iterations=torch.zeros(1, dtype=torch.int32, device=dev)))
        opt = F.DirectMethodOptions()
        opt.kMaxTrackPointsNumber, opt.kPatchRowHalfSize, opt.kPatchColHalfSize = 300, half, half
        D.DeviceDirectBatch(opt, problems, ctx).track()
        stream.synchronize()
    same = True
    for (uv, pts, q0, p0), pr in zip(host, problems):
        pose = pr["pose"].cpu().numpy()
        if len(uv) == 0:
            same = same and np.array_equal(pose.view(np.uint32), np.concatenate([q0, p0]).astype(np.float32).view(np.uint32))
            continue
        with np.errstate(all="ignore"):
            ok, c, q, p, st, it = O.direct_track(rl, cl, K, pts, uv, None, q0, p0, half=half, max_points=300)
        same = same and (np.array_equal(pose[:4].view(np.uint32), np.float32(q).view(np.uint32)) and np.array_equal(pose[4:].view(np.uint32), np.float32(p).view(np.uint32)) and
                         np.array_equal(pr["cur_uv"].cpu().numpy().view(np.uint32), c.view(np.uint32)) and np.array_equal(pr["status"].cpu().numpy(), st) and
                         int(pr["iterations"].cpu().numpy()[0]) == it)
    note(f"direct batch of {count:2d}", same, f"{w}x{h} L{levels} h{half}")


t_end = time.time() + budget
rounds = 0
t_report = time.time() + 60.0
while time.time() < t_end:
    if ONLY == "matcher":
        matcher_round()
    elif ONLY == "direct":
        direct_batch_round()
    elif ONLY == "tree":
        klt_round()
        direct_round()
    else:
        klt_round()
        matcher_round()
        direct_round()
    rounds += 1
    if time.time() > t_report:  # a line a minute: the GPU box takes a silent command for a hung one
        print(f"soak: {rounds} rounds, {sum(v[0] for v in stats.values())} comparisons, {sum(v[1] for v in stats.values())} mismatches so far", flush=True)
        t_report = time.time() + 60.0
if ONLY == "tree":
    print("throughput mode (ftk_set_reduction_mode(TREE)) against the oracle: REPORT, nothing asserted")
    print(f"{'variant':22s} {'features':>9s} {'identical':>10s} {'max px':>10s} {'p99 px':>10s} {'frac>1e-3':>10s} {'status !=':>10s}")
    for fam in sorted(tree_stats):
        t = tree_stats[fam]
        d = np.array(t["d"]) if t["d"] else np.zeros(1)
        print(f"{fam:22s} {t['n']:9d} {t['identical'] / max(t['n'], 1):10.4f} {t['max']:10.3g} {np.percentile(d, 99):10.3g} {t['gt'] / max(t['n'], 1):10.2e} {t['status']:10d}")
    print(f"soak (tree, reporting): {rounds} rounds in {budget:.0f} s")
    sys.exit(0)
bad = 0
for fam in sorted(stats):
    n, f, d = stats[fam]
    bad += f
    print(f"{fam:24s} cases {n:5d}  mismatches {f}" + (f"   first: {d}" if f else ""))
print(f"soak: {rounds} rounds in {budget:.0f} s budget, {sum(v[0] for v in stats.values())} comparisons, {bad} mismatches")
sys.exit(1 if bad else 0)
