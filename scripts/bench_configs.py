#!/usr/bin/env python3
"""Times every BASELINE.json configuration (and all nine tracker variants on config 2) on one
MI355X next to the CPU oracle, and checks parity on the same inputs.

    python scripts/bench_configs.py [--quick] > gpurun_out/bench_configs.jsonl

One JSON line per case: GPU kernel time (HIP events on the launch stream, median of `reps`
launches on device-resident inputs), host-call time through the C ABI (H2D of the feature vectors,
launch, D2H), CPU oracle time (single thread), parity (max |duv|, bit-identical fraction, status
mismatches).  Used for the tables in DESIGN.md; bench.py remains the contractual benchmark.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def klt_case(name, cfg, torch, F, D, synth, oracle, reps, cpu_reps, motion=(3.3, -2.1)):
    n, w, h, levels, half, model, method = cfg["n"], cfg["width"], cfg["height"], cfg["levels"], cfg["half"], cfg["model"], cfg["method"]
    if model == "basic":
        ref_img, cur_img = synth.make_image_pair(w, h, motion)
    else:
        ref_img, cur_img = synth.make_image_pair(w, h, motion, rotation_deg=1.5, scale=1.02)
    ref_levels, cur_levels = synth.build_pyramid(ref_img, levels), synth.build_pyramid(cur_img, levels)
    uv = synth.make_features(n, w, h, half=half)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        ref_pyr, cur_pyr = D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev)
        opt = F.OpticalFlowOptions()
        opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, half, half, n
        klt = D.DeviceKlt(model, opt, ref_pyr, cur_pyr, ctx, consider_luminance=cfg.get("luminance", False))
        d_ref = torch.from_numpy(uv).to(dev)
        d_in = d_ref.clone()
        d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
        d_out = torch.empty_like(d_ref)
        d_sto = torch.empty_like(d_st)
        d_it = torch.zeros(n, dtype=torch.int32, device=dev)
        klt.track(d_ref, d_in, d_st, d_out, d_sto, d_it)
        stream.synchronize()
        for _ in range(3):
            klt.track(d_ref, d_in, d_st, d_out, d_sto, None)
        times = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            klt.track(d_ref, d_in, d_st, d_out, d_sto, None)
            e1.record(stream)
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
        gpu_uv, gpu_st, iters = d_out.cpu().numpy(), d_sto.cpu().numpy(), d_it.cpu().numpy()
        # Launch order (calls of >= 4096 features are launched longest-first by the iteration counts of two calls before,
        # ftk_klt_track_device): the loop above repeats ONE call, so its predictor is perfect ("warm").  Two more regimes:
        #   cold     — no history: a call with another feature count in between resets it, every timed call runs in list order;
        #   permuted — the feature list is reshuffled between calls (features re-detected in another order): the counts of two
        #              calls ago belong to other features, i.e. the order in use is an arbitrary one.
        order_regimes = None
        if n >= 4096:
            def timed(fn_before, args):
                out = []
                for k in range(reps):
                    fn_before(k)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    klt.track(*args(k))
                    e1.record(stream)
                    e1.synchronize()
                    out.append(e0.elapsed_time(e1))
                return float(np.median(out))

            def reset_history(_k):
                klt.track(d_ref[: n - 1], d_in[: n - 1], d_st[: n - 1], d_out[: n - 1], d_sto[: n - 1], None)

            cold_ms = timed(reset_history, lambda k: (d_ref, d_in, d_st, d_out, d_sto, None))
            rs = np.random.RandomState(5)
            shuffled = [torch.from_numpy(np.ascontiguousarray(uv[rs.permutation(n)])).to(dev) for _ in range(5)]
            for k in range(4):  # history of the same length as in the timed loop
                klt.track(shuffled[k % 5], shuffled[k % 5], d_st, d_out, d_sto, None)
            perm_ms = timed(lambda _k: None, lambda k: (shuffled[k % 5], shuffled[k % 5], d_st, d_out, d_sto, None))
            order_regimes = {"warm_ms": float(np.median(times)), "cold_ms": cold_ms, "permuted_ms": perm_ms}
            stream.synchronize()
    # host-call path (what the C++ / Python classes do per TrackFeatures)
    cls = {"basic": F.OpticalFlowBasicKlt, "affine": F.OpticalFlowAffineKlt, "lssd": F.OpticalFlowLssdKlt}[model]()
    o = cls.options()
    o.kMethod, o.kPatchRowHalfSize, o.kPatchColHalfSize, o.kMaxTrackPointsNumber = method, half, half, n
    if model == "lssd":
        cls.consider_patch_luminance = cfg.get("luminance", False)
    rp, cp = F.ImagePyramid.from_host_levels(ref_levels), F.ImagePyramid.from_host_levels(cur_levels)
    cls.TrackFeatures(rp, cp, uv)
    host_times = []
    for _ in range(max(3, reps // 4)):
        t0 = time.perf_counter()
        cls.TrackFeatures(rp, cp, uv)
        host_times.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    up = F.ImagePyramid.from_host_levels(ref_levels)
    upload_ms = (time.perf_counter() - t0) * 1e3
    del up
    cpu_times = []
    for _ in range(cpu_reps):
        t0 = time.perf_counter()
        ok, cuv, cst, cit = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, method=method, half=half, max_points=n,
                                                     consider_luminance=cfg.get("luminance", False))
        cpu_times.append((time.perf_counter() - t0) * 1e3)
    finite = np.isfinite(cuv).all(axis=1)
    d = np.abs(gpu_uv[finite].astype(np.float64) - cuv[finite].astype(np.float64)).max(axis=1) if finite.any() else np.zeros(0)
    gpu_ms, cpu_ms = float(np.median(times)), float(np.median(cpu_times))
    r_bytes = (2 * half + 4) ** 2 if method != "direct" else (2 * half + 2) ** 2
    c_bytes = (2 * half + 2) ** 2 if method != "direct" else (2 * half + 4) ** 2
    algo = n * (levels * r_bytes + 26) + int(iters.sum()) * c_bytes
    return {
        "case": name, "model": model, "method": method, "n": n, "image": f"{w}x{h}", "levels": levels, "patch": 2 * half + 1,
        "gpu_kernel_ms": gpu_ms, "gpu_features_per_s": n / gpu_ms * 1e3, "gpu_host_call_ms": float(np.median(host_times)),
        "pyramid_upload_ms": upload_ms, "cpu_ms": cpu_ms, "cpu_features_per_s": n / cpu_ms * 1e3, "speedup_kernel": cpu_ms / gpu_ms,
        "tracked_fraction": float((cst == 1).mean()), "mean_iters": float(iters.mean()), "max_iters": int(iters.max()),
        "parity_max_px": float(d.max()) if d.size else 0.0, "parity_frac_gt_1e-3": float((d > 1e-3).mean()) if d.size else 0.0,
        "bit_identical": bool(np.array_equal(gpu_uv.view(np.uint32), cuv.view(np.uint32))), "status_mismatches": int((gpu_st != cst).sum()),
        "algorithmic_bytes": int(algo), "algorithmic_GBps": algo / (gpu_ms * 1e-3) / 1e9, "launch_order_regimes": order_regimes,
    }


def matcher_case(name, n_ref, n_cur, nearby, torch, F, D, synth, oracle, reps, cpu_pairs, raster=False):
    ref, cur, perm = synth.make_descriptors(n_ref, n_cur, flips=20)
    rs = np.random.RandomState(11)
    cur_uv = np.stack([rs.uniform(0, 640, n_cur), rs.uniform(0, 480, n_cur)], axis=1).astype(np.float32)
    pred_uv = np.stack([rs.uniform(0, 640, n_ref), rs.uniform(0, 480, n_ref)], axis=1).astype(np.float32)
    if raster:  # the order a detector scanning the image returns features in (bands of 4 rows)
        oc = np.lexsort((cur_uv[:, 0], np.floor(cur_uv[:, 1] / 4)))
        cur, cur_uv = np.ascontiguousarray(cur[oc]), np.ascontiguousarray(cur_uv[oc])
        orf = np.lexsort((pred_uv[:, 0], np.floor(pred_uv[:, 1] / 4)))
        ref, pred_uv = np.ascontiguousarray(ref[orf]), np.ascontiguousarray(pred_uv[orf])
    ref_w, cur_w = F.pack_brief(ref), F.pack_brief(cur)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        d_ref = torch.from_numpy(ref_w.view(np.int32)).to(dev)
        d_cur = torch.from_numpy(cur_w.view(np.int32)).to(dev)
        d_idx = torch.full((n_ref,), -1, dtype=torch.int32, device=dev)
        d_ws = torch.zeros(n_ref, dtype=torch.int64, device=dev)
        d_pred = torch.from_numpy(pred_uv).to(dev) if nearby else None
        d_cuv = torch.from_numpy(cur_uv).to(dev) if nearby else None
        args = dict(pred_uv=d_pred, cur_uv=d_cuv, max_col=50, max_row=50)  # context-owned key workspace
        D.hamming_match_device(ctx, d_ref, d_cur, 256, 60.0, d_idx, **args)
        stream.synchronize()
        times = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            D.hamming_match_device(ctx, d_ref, d_cur, 256, 60.0, d_idx, **args)
            e1.record(stream)
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
        gpu_idx = d_idx.cpu().numpy()
    # CPU oracle on a bounded sample of ref rows (full candidate set), scaled by pairs
    rows = max(1, min(n_ref, cpu_pairs // n_cur))
    t0 = time.perf_counter()
    if nearby:
        ok, cidx = oracle.nearby_match(ref[:rows], cur, pred_uv[:rows], cur_uv, 60.0, max_col=50, max_row=50)
    else:
        ok, cidx = oracle.force_match(ref[:rows], cur, 60.0)
    cpu_s = time.perf_counter() - t0
    gpu_ms = float(np.median(times))
    pairs = n_ref * n_cur
    return {
        "case": name, "mode": "nearby" if nearby else "force", "n_ref": n_ref, "n_cur": n_cur, "bits": 256, "gpu_kernel_ms": gpu_ms,
        "gpu_pairs_per_s": pairs / gpu_ms * 1e3, "cpu_sample_rows": rows, "cpu_sample_s": cpu_s,
        "cpu_ns_per_pair": cpu_s / (rows * n_cur) * 1e9 if not nearby else None, "cpu_full_estimate_s": cpu_s * n_ref / rows,
        "speedup": (cpu_s * n_ref / rows) / (gpu_ms * 1e-3), "indices_bit_exact_on_sample": bool(np.array_equal(gpu_idx[:rows], cidx)),
        "matched": int((gpu_idx >= 0).sum()),
        # 256-bit descriptors are compared on the matrix cores: one int8 multiply-add per bit pair; the peak is the
        # dense int8 rate of MI355X_MICROARCH.md (2 x BF16 per clock)
        "roofline": {"bound": "mfma", "achieved": 2.0 * pairs * 256 / (gpu_ms * 1e-3) / 1e12, "peak": 5000.0, "unit": "TOP/s (int8, per call incl. the epilogue launch)",
                     "frac": 2.0 * pairs * 256 / (gpu_ms * 1e-3) / 1e12 / 5000.0, "traffic": None},
    }


def float_matcher_case(name, n_ref, n_cur, dim, nearby, torch, F, D, synth, oracle, reps, cpu_pairs, raster=False):
    """SuperPoint-256 / DISK-128 shaped cosine matcher.  raster: both feature sets in the order a
    detector scanning the image returns them (bands of 4 rows), instead of random order."""
    ref, cur, perm = synth.make_float_descriptors(n_ref, n_cur, dim=dim, noise=0.2)
    rs = np.random.RandomState(11)
    cur_uv = np.stack([rs.uniform(0, 640, n_cur), rs.uniform(0, 480, n_cur)], axis=1).astype(np.float32)
    pred_uv = np.stack([rs.uniform(0, 640, n_ref), rs.uniform(0, 480, n_ref)], axis=1).astype(np.float32)
    if raster:
        oc = np.lexsort((cur_uv[:, 0], np.floor(cur_uv[:, 1] / 4)))
        cur, cur_uv = np.ascontiguousarray(cur[oc]), np.ascontiguousarray(cur_uv[oc])
        orf = np.lexsort((pred_uv[:, 0], np.floor(pred_uv[:, 1] / 4)))
        ref, pred_uv = np.ascontiguousarray(ref[orf]), np.ascontiguousarray(pred_uv[orf])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        d_ref, d_cur = torch.from_numpy(ref).to(dev), torch.from_numpy(cur).to(dev)
        d_idx = torch.full((n_ref,), -1, dtype=torch.int32, device=dev)
        d_pred = torch.from_numpy(pred_uv).to(dev) if nearby else None
        d_cuv = torch.from_numpy(cur_uv).to(dev) if nearby else None
        args = dict(pred_uv=d_pred, cur_uv=d_cuv, max_col=50, max_row=50)
        D.cosine_match_device(ctx, d_ref, d_cur, 0.1, d_idx, **args)
        stream.synchronize()
        times = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            D.cosine_match_device(ctx, d_ref, d_cur, 0.1, d_idx, **args)
            e1.record(stream)
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
        gpu_idx = d_idx.cpu().numpy()
    rows = max(1, min(n_ref, cpu_pairs // n_cur))
    t0 = time.perf_counter()
    if nearby:
        ok, cidx = oracle.match_float(ref[:rows], cur, 0.1, pred_uv[:rows], cur_uv, max_col=50, max_row=50)
    else:
        ok, cidx = oracle.match_float(ref[:rows], cur, 0.1)
    cpu_s = time.perf_counter() - t0
    gpu_ms = float(np.median(times))
    flops = 2.0 * n_ref * n_cur * dim  # ONE fp16 MFMA walk over all pairs (the default single-walk kernel); per CALL, prep / recheck included in the time
    return {
        "case": name, "mode": "nearby" if nearby else "force", "n_ref": n_ref, "n_cur": n_cur, "dim": dim, "gpu_call_ms": gpu_ms,
        "gpu_pairs_per_s": n_ref * n_cur / gpu_ms * 1e3, "mfma_TFLOPs": flops / (gpu_ms * 1e-3) / 1e12, "cpu_sample_rows": rows,
        "cpu_sample_s": cpu_s, "cpu_full_estimate_s": cpu_s * n_ref / rows, "speedup": (cpu_s * n_ref / rows) / (gpu_ms * 1e-3),
        "indices_bit_exact_on_sample": bool(np.array_equal(gpu_idx[:rows], cidx)), "matched": int((gpu_idx >= 0).sum()),
    }


def direct_method_cases(torch, F, D, synth, oracle, quick=False):
    """DirectMethod: the reference's shape (300 points, 13 x 13, 5 levels, 1241 x 376) as one problem and as
    a batch of independent problems in one launch; CPU = the oracle, single thread."""
    w, h, levels, n, half = 1241, 376, 5, 300, 6
    fx = fy = 718.856
    cx, cy = 607.1928, 185.2157
    ref_img, cur_img = synth.make_image_pair(w, h, (2.6, -1.2))
    rl, cl = synth.build_pyramid(ref_img, levels), synth.build_pyramid(cur_img, levels)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    out = []
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        rp, cp = D.upload_pyramid(rl, ctx, dev), D.upload_pyramid(cl, ctx, dev)
        for batch in (1, 64, 256):
            problems, host = [], []
            for b in range(batch):
                uv = synth.make_features(n, w, h, seed=100 + b, half=half)
                z = np.full(n, 8.0, np.float32)
                pts = np.stack([(uv[:, 0] - cx) / fx * z, (uv[:, 1] - cy) / fy * z, z], axis=1).astype(np.float32)
                host.append((uv, pts))
                problems.append(dict(ref=rp, cur=cp, K=[fx, fy, cx, cy], p_c_in_ref=torch.from_numpy(pts).to(dev), ref_uv=torch.from_numpy(uv).to(dev),
                                     cur_uv=torch.from_numpy(uv).to(dev), pose=torch.zeros(7, dtype=torch.float32, device=dev),
                                     status=torch.zeros(n, dtype=torch.uint8, device=dev), status_valid=False,
                                     iterations=torch.zeros(1, dtype=torch.int32, device=dev)))
            opt = F.DirectMethodOptions()
            opt.kMaxTrackPointsNumber = n
            runner = D.DeviceDirectBatch(opt, problems, ctx)

            def reset():
                for pr, (uv, _) in zip(problems, host):
                    pr["cur_uv"].copy_(torch.from_numpy(uv))
                    pr["pose"].copy_(torch.tensor([1, 0, 0, 0, 0, 0, 0], dtype=torch.float32))
            times = []
            for _ in range(3 if quick else 6):
                reset()
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                runner.track()
                e1.record(stream)
                e1.synchronize()
                times.append(e0.elapsed_time(e1))
            gpu_ms = float(np.median(times))
            pose0 = problems[0]["pose"].cpu().numpy()
            it0 = int(problems[0]["iterations"].cpu().numpy()[0])
            # the same launch in the throughput mode (butterfly sums; reported, not the contract)
            ctx.set_reduction("tree")
            tree_times = []
            for _ in range(3):
                reset()
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                runner.track()
                e1.record(stream)
                e1.synchronize()
                tree_times.append(e0.elapsed_time(e1))
            ctx.set_reduction("exact")
            tree_pose = problems[0]["pose"].cpu().numpy()
            tree_it = int(problems[0]["iterations"].cpu().numpy()[0])
            uv, pts = host[0]
            t0 = time.perf_counter()
            ok, c, q, p, st, it = oracle.direct_track(rl, cl, [fx, fy, cx, cy], pts, uv, max_points=n)
            cpu_ms = (time.perf_counter() - t0) * 1e3
            out.append({"case": f"direct_method_{n}pts_{w}x{h}_{levels}lvl_batch{batch}", "problems": batch, "gpu_launch_ms": gpu_ms,
                        "gpu_ms_per_problem": gpu_ms / batch, "cpu_ms_per_problem": cpu_ms, "speedup": cpu_ms * batch / gpu_ms, "iterations": it0,
                        "bit_identical_pose": bool(np.array_equal(pose0[:4].view(np.uint32), q.view(np.uint32)) and
                                                   np.array_equal(pose0[4:].view(np.uint32), p.view(np.uint32))),
                        "iterations_equal": it0 == it,
                        "throughput_mode": {"gpu_launch_ms": float(np.median(tree_times)), "iterations": tree_it,
                                            "max_abs_pose_difference": float(np.abs(tree_pose.astype(np.float64) - pose0.astype(np.float64)).max())}})
    return out


def producer_cases(torch, F, D, synth, oracle, reps):
    """Harris detection and BRIEF description on the reference's example-sized image."""
    from PIL import Image
    img = np.array(Image.open(os.path.join(ROOT, "tests", "data", "optical_flow", "ref_image.png")))
    out = []
    det = F.FeaturePointHarrisDetector()
    det.options().kMinFeatureDistance, det.options().kMinValidResponse = 25, 40.0
    pyr = F.ImagePyramid.from_host_levels([img])
    det.DetectGoodFeatures(pyr, 300)
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ok, uv = det.DetectGoodFeatures(pyr, 300)
        t.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    ouv = oracle.harris_detect(img, 300, 25, 40.0)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    out.append({"case": "harris_752x480_300", "gpu_host_call_ms": float(np.median(t)), "cpu_ms": cpu_ms, "features": int(len(uv)),
                "identical": bool(np.array_equal(uv, ouv))})
    for n in (300, 10000):
        fuv = synth.make_features(n, 752, 480, seed=5, half=8)
        d = F.BriefDescriptor()
        d.compute_packed(pyr, fuv)
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            words = d.compute_packed(pyr, fuv)
            t.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        okc, bits = oracle.brief_compute(img, fuv, 256, 8)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        out.append({"case": f"brief256_{n}", "gpu_host_call_ms": float(np.median(t)), "cpu_ms": cpu_ms,
                    "identical": bool(np.array_equal(words, F.pack_brief(bits)))})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="'cosine': just the float-descriptor matcher cases; 'match': just the Hamming matcher at 10 000 x 10 000")
    args = ap.parse_args()
    import torch

    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import synth
    from tests import oracle_lib as oracle

    oracle.lib()
    reps = 10 if args.quick else 40
    if args.only == "cosine":
        float_matcher_cases(torch, F, D, synth, oracle, reps, args.quick)
        return
    if args.only == "direct":
        for out in direct_method_cases(torch, F, D, synth, oracle, args.quick):
            print(json.dumps(out), flush=True)
        return
    if args.only == "match":
        for name, n_ref, n_cur, nearby, raster in (("match_config4_force", 10000, 10000, False, False), ("match_config4_nearby", 10000, 10000, True, False),
                                                   ("match_config4_nearby_raster_order", 10000, 10000, True, True)):
            print(json.dumps(matcher_case(name, n_ref, n_cur, nearby, torch, F, D, synth, oracle, reps, 4_000_000, raster)), flush=True)
        return
    if args.only.startswith("fast"):
        # the reference's default method on every model, at a front end's sizes and at the BASELINE sizes; "fast:basic" etc. narrows it
        want = args.only.split(":")[1] if ":" in args.only else ""
        for model in ("basic", "affine", "lssd"):
            if want and model != want:
                continue
            for n, half, levels in ((300, 6, 4), (2000, 6, 4), (2000, 10, 4), (10000, 6, 4)):
                c = dict(synth.CONFIGS["config2"])
                c.update(model=model, method="fast", half=half, n=n, levels=levels)
                print(json.dumps(klt_case(f"fast/{model}/{n}x{2 * half + 1}", c, torch, F, D, synth, oracle, reps, 1)), flush=True)
        return
    cases = []
    for key in ("config1", "config2", "config3", "config4", "config5_shard"):
        cases.append((key, dict(synth.CONFIGS[key])))
    for model in ("basic", "affine", "lssd"):
        for method in ("inverse", "direct", "fast"):
            c = dict(synth.CONFIGS["config2"])
            c.update(model=model, method=method, half=6)
            cases.append((f"variants_2000x13x13/{model}/{method}", c))
    c = dict(synth.CONFIGS["config4"])
    c["luminance"] = True
    cases.append(("config4_luminance", c))
    for name, cfg in cases:
        big = cfg["n"] * cfg["levels"] > 60000
        out = klt_case(name, cfg, torch, F, D, synth, oracle, reps, 1 if (big or args.quick) else 3)
        print(json.dumps(out), flush=True)
    for out in producer_cases(torch, F, D, synth, oracle, reps):
        print(json.dumps(out), flush=True)
    for name, n_ref, n_cur, nearby, raster in (("match_config4_force", 10000, 10000, False, False), ("match_config4_nearby", 10000, 10000, True, False),
                                               ("match_config4_nearby_raster_order", 10000, 10000, True, True),
                                               ("match_300x300_nearby", 300, 300, True, False), ("match_2000_force", 2000, 2000, False, False)):
        out = matcher_case(name, n_ref, n_cur, nearby, torch, F, D, synth, oracle, reps, 4_000_000 if args.quick else 20_000_000, raster)
        print(json.dumps(out), flush=True)
    float_matcher_cases(torch, F, D, synth, oracle, reps)
    for out in direct_method_cases(torch, F, D, synth, oracle, args.quick):
        print(json.dumps(out), flush=True)


def float_matcher_cases(torch, F, D, synth, oracle, reps, quick=False):
    for name, n_ref, n_cur, dim, nearby, raster in (("cosine_superpoint256_10000_force", 10000, 10000, 256, False, False),
                                                    ("cosine_superpoint256_10000_nearby", 10000, 10000, 256, True, False),
                                                    ("cosine_superpoint256_10000_nearby_raster_order", 10000, 10000, 256, True, True),
                                                    ("cosine_disk128_10000_force", 10000, 10000, 128, False, False),
                                                    ("cosine_superpoint256_300_nearby", 300, 300, 256, True, False)):
        out = float_matcher_case(name, n_ref, n_cur, dim, nearby, torch, F, D, synth, oracle, reps, 2_000_000 if quick else 8_000_000, raster)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
