#!/usr/bin/env python3
"""bench.py — tracked features/s of the pyramidal KLT hot path on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): BasicKlt inverse,
2000 features, 640x480 synthetic pair, 4-level pyramid, 21x21 patch — per GPU (weak scaling).
A step = ONE TrackFeatures pass over the batch: one kernel launch on device-resident inputs
(pyramids, ref_uv, predicted cur_uv, status already in HBM) writing (u, v) + status, and for
N > 1 the RCCL all-gather of every rank's packed result shard.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line (contract in the task statement) with these extra objects:
  roofline      : algorithmic bytes of the tracker kernel / its average launch duration against the
                  8 TB/s HBM peak (the bound north_star names); traffic = PMC-derived HBM bytes per
                  launch from the committed counter summary under profiles/, else null.
  roofline_valu : the bound that actually binds this kernel — vector-ALU issue: SQ_INSTS_VALU per
                  launch (committed PMC summary) / kernel time against 1024 SIMDs x 2.4 GHz / 2 cycles
                  per wave64 instruction (MI355X_MICROARCH.md, cycle constants).
  parity        : px-error vs CPU (BASELINE.json's metric): the first step's (u, v, status) of the
                  device path against the oracle on the same inputs.
  cpu_baseline  : the oracle (CPU restatement, single thread) timed on a bounded sample of the same
                  workload on this host.
The timed region contains ONLY the K steps (no event records); the kernel duration for the roofline
objects comes from a separate pass of the same launches bracketed by HIP events on the launch stream.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(iters: np.ndarray, levels: int, half: int, method: str) -> int:
    """B(f) = sum_l [R + it(f,l) * C] + 26 with R, C the image footprints of the
    gradient-side and residual-side samples; it(f, .) is counted by the kernel."""
    if method == "direct":
        r, c = (2 * half + 2) ** 2, (2 * half + 4) ** 2
    else:
        r, c = (2 * half + 4) ** 2, (2 * half + 2) ** 2
    return int(iters.size * (levels * r + 26) + int(iters.astype(np.int64).sum()) * c)


# Vector-ALU issue peak: 256 CUs x 4 SIMD-32, one wave64 instruction per 2 cycles per SIMD (MI355X_MICROARCH.md, cycle
# constants) — `peak` and `frac` of roofline_valu are quoted against THIS figure in every round (round 3 had moved
# them to a self-measured rate, which made fractions incomparable across rounds).  What a saturated SIMD was MEASURED to sustain
# on this chip (scripts/microbench/int_valu_rate.hip, profiles/r2_microbench_int_valu.txt: 2.5 cycles per wave-instruction) is
# reported beside it as the secondary field frac_of_measured_peak.
VALU_CYCLES_PER_WAVE_INST_MEASURED = 2.5
VALU_PEAK_DATASHEET = 1024 * 2.4e9 / 2.0
VALU_PEAK_MEASURED = 1024 * 2.4e9 / VALU_CYCLES_PER_WAVE_INST_MEASURED


def pmc_profile(workload: str, source_hash: str):
    """Counter-derived per-launch figures of the workload's tracker kernel from the committed rocprofv3 PMC summary
    (profiles/pmc_traffic.json, written by scripts/summarize_profile.py from separate --pmc passes over THIS bench command): HBM
    bytes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE) and vector-ALU wave-instructions.  Counters cannot be collected inside a timed
    run, so they are tied to the build instead: the summary records the library's source hash (ftk_build_info()), and figures
    whose hash is not the RUNNING library's are refused — returned as (None, why) — rather than silently reported for kernels
    that have changed since.  Returns (entry, None) when they belong to this build."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        return None, "no counter summary committed (profiles/pmc_traffic.json)"
    w = d.get("workloads", {}).get(workload)
    if not w:
        return None, f"no counter summary for workload {workload}"
    recorded = w.get("source_hash")
    if not recorded:
        return None, f"the committed counters ({w.get('source')}) predate build hashes: taken on an earlier build of the kernels"
    if recorded != source_hash:
        return None, f"the committed counters ({w.get('source')}) were taken on build {recorded}; this library is build {source_hash}"
    return w, None


def parity_report(gpu_uv, gpu_st, cpu_uv, cpu_st) -> dict:
    """px-error vs CPU: |d(u, v)| per feature (Euclidean, over features whose CPU result is finite) and status mismatches."""
    finite = np.isfinite(cpu_uv).all(axis=1)
    d = np.linalg.norm(gpu_uv[finite].astype(np.float64) - cpu_uv[finite].astype(np.float64), axis=1) if finite.any() else np.zeros(0)
    return {
        "max_px": float(d.max()) if d.size else 0.0, "p99_px": float(np.percentile(d, 99)) if d.size else 0.0,
        "frac_gt_1e-3": float((d > 1e-3).mean()) if d.size else 0.0, "status_mismatches": int((gpu_st != cpu_st).sum()),
        "bit_identical": bool(np.array_equal(gpu_uv.view(np.uint32), cpu_uv.view(np.uint32)) and np.array_equal(gpu_st, cpu_st)),
        "features": int(cpu_st.size), "tolerance_px": 1e-3, "against": "oracle/liboracle.so (CPU restatement, parity unpinned)",
    }


def oracle_once(cfg, ref_levels, cur_levels, uv):
    from tests import oracle_lib

    oracle_lib.lib()
    t0 = time.perf_counter()
    ok, c, s, it = oracle_lib.klt_track_pyramid(cfg["model"], ref_levels, cur_levels, uv, method=cfg["method"], half=cfg["half"], max_points=cfg["n"])
    return c, s, it, time.perf_counter() - t0


def cpu_baseline(cfg, ref_levels, cur_levels, uv, budget_s=12.0):
    """Times the oracle on this host: single thread (the reference has no threads), whole-workload
    calls repeated until ~budget_s of CPU work, median per-call time."""
    times = []
    t_all = time.perf_counter()
    while True:
        times.append(oracle_once(cfg, ref_levels, cur_levels, uv)[3])
        if time.perf_counter() - t_all > budget_s or len(times) >= 200:
            break
    med = float(np.median(times))
    return {
        "value": cfg["n"] / med, "unit": "tracked features/s", "cores": 1, "kind": "port",
        "sample": f"{len(times)} full calls of the workload ({cfg['n']} features each), median {med * 1e3:.2f} ms/call, "
                  f"oracle/liboracle.so (gcc -O3, no -march, -ffp-contract=off), host cpus={os.cpu_count()}",
    }


def with_pyramid_upload(args, cfg, ctx, klt, opt, ref_img, cur_img, d_ref, d_cur_in, d_st_in, n, levels, stream, out_views):
    """the metric "with H2D pyramid upload".  The reference's own timed region (test/test_optical_flow.cpp:69-73)
    spans CreateImagePyramid x 2 + TrackFeatures; here one call = both raw frames copied from pinned host memory into HBM,
    levels >= 1 of both pyramids rebuilt on the device (ftk_pyramid_update, no allocation), then the same tracker launch on the
    same features — all stream-ordered, one synchronisation at the end of the K calls.  `value` is never the headline (inputs
    are not resident when the timed region starts); it is reported beside it."""
    import torch

    import feature_tracker_amd as F
    from feature_tracker_amd import device as D

    h_ref, h_cur = torch.from_numpy(np.ascontiguousarray(ref_img)).pin_memory(), torch.from_numpy(np.ascontiguousarray(cur_img)).pin_memory()
    pr, pc = F.ImagePyramid.build(ref_img, levels, ctx), F.ImagePyramid.build(cur_img, levels, ctx)
    klt_up = D.DeviceKlt(cfg["model"], opt, pr, pc, ctx)
    launch = klt_up.bind(d_ref, d_cur_in, d_st_in, out_views[0], out_views[1], None)
    p_ref, p_cur = h_ref.data_ptr(), h_cur.data_ptr()

    def call():
        pr.update(p_ref, "host_async")
        pc.update(p_cur, "host_async")
        launch()

    for _ in range(max(3, min(args.warmup, 10))):
        call()
    stream.synchronize()
    steps = max(10, min(args.steps, 200))
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    stream.synchronize()
    elapsed = time.perf_counter() - t0
    # one call alone, synchronised: the latency a caller that waits for every frame sees
    lat = []
    for _ in range(20):
        t1 = time.perf_counter()
        call()
        stream.synchronize()
        lat.append(time.perf_counter() - t1)
    return {"value": n * steps / elapsed, "unit": "tracked features/s", "ms_per_call": elapsed / steps * 1e3, "calls_timed": steps,
            "ms_per_call_synchronised": float(np.median(lat)) * 1e3,
            "includes": f"two {ref_img.shape[1]}x{ref_img.shape[0]} frames in pinned host memory, read over PCIe by the pyramid launches themselves (no copy-engine transfer) + device 2x2 downsample of levels 1..{levels - 1} "
                        "of both pyramids (ftk_pyramid_update) + the tracker launch; inputs are NOT resident when the region starts",
            "result_uv": out_views[0], "result_st": out_views[1]}


def _scene(cfg, synth):
    if cfg["model"] == "basic":
        ref_img, cur_img = synth.make_image_pair(cfg["width"], cfg["height"], (3.3, -2.1))
    else:
        ref_img, cur_img = synth.make_image_pair(cfg["width"], cfg["height"], (3.3, -2.1), rotation_deg=1.5, scale=1.02)
    return synth.build_pyramid(ref_img, cfg["levels"]), synth.build_pyramid(cur_img, cfg["levels"])


def tracker_config_leg(name, cfg, steps, ctx, dev, stream, source_hash):
    """One BASELINE.json configuration timed the way the headline is (K back-to-back launches on device-resident inputs, wall
    clock around them, one synchronisation on each side), checked against the oracle on the same inputs.  Calls of >= 4096
    features are launched through the launch order of an earlier call (ftk_klt_track_device): repeating ONE call makes that
    predictor perfect, so the back-to-back figure is the "warm" regime; "cold" (a call with another feature count in between resets
    the index-keyed history — what a front end that drops and re-detects features does every frame; one bracketed call at a time) is
    given beside it.  Since round 4 such a call of the LSSD / affine trackers is ordered by what the call before left at its
    features' positions (ftk_api.cpp, klt_position_order_launch); the Basic trackers run it in list order."""
    import torch

    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import synth
    from tests import oracle_lib

    n, levels, half = cfg["n"], cfg["levels"], cfg["half"]
    lum = bool(cfg.get("luminance", False))
    ref_levels, cur_levels = _scene(cfg, synth)
    uv = synth.make_features(n, cfg["width"], cfg["height"], half=half)
    opt = F.OpticalFlowOptions()
    opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = cfg["method"], half, half, n
    klt = D.DeviceKlt(cfg["model"], opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx, consider_luminance=lum)
    d_ref = torch.from_numpy(uv).to(dev)
    d_in, d_st = d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev)
    outs = [(torch.empty_like(d_ref), torch.empty_like(d_st)) for _ in range(2)]
    d_it = torch.zeros(n, dtype=torch.int32, device=dev)
    klt.track(d_ref, d_in, d_st, outs[0][0], outs[0][1], d_it)
    stream.synchronize()
    iters = d_it.cpu().numpy().astype(np.uint32)
    first_uv, first_st = outs[0][0].cpu().numpy().copy(), outs[0][1].cpu().numpy().copy()
    launches = [klt.bind(d_ref, d_in, d_st, o[0], o[1], None) for o in outs]
    # Untimed launches for at least 10 ms: the leg before this one ended with its CPU oracle (0.1 - 0.5 s during which the device idles and
    # drops its clock), and a 20-step timed region of 3 ms would otherwise sit inside the ramp back up (config 4: 153 us per step at 20
    # steps against 143 at 100 — the same launches).  What is measured is the steady back-to-back rate, as for the headline.
    t_warm, k = time.perf_counter(), 0
    while k < 6 or time.perf_counter() - t_warm < 10e-3:
        launches[k & 1]()
        k += 1
        if k % 8 == 0:
            stream.synchronize()
    stream.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        launches[k & 1]()
    stream.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    last_uv, last_st = outs[(steps - 1) & 1][0].cpu().numpy(), outs[(steps - 1) & 1][1].cpu().numpy()
    cold_ms = None
    if n >= 4096:
        lat = []
        for _ in range(7):
            klt.track(d_ref[: n - 1], d_in[: n - 1], d_st[: n - 1], outs[1][0][: n - 1], outs[1][1][: n - 1], None)  # another count: history reset
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            launches[0]()
            e1.record(stream)
            e1.synchronize()
            lat.append(e0.elapsed_time(e1))
        cold_ms = float(np.median(lat))
    t0 = time.perf_counter()
    ok, cpu_uv, cpu_st, cpu_it = oracle_lib.klt_track_pyramid(cfg["model"], ref_levels, cur_levels, uv, method=cfg["method"], half=half, max_points=n,
                                                              consider_luminance=lum)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    same = lambda a, b: bool(np.array_equal(a[0].view(np.uint32), cpu_uv.view(np.uint32)) and np.array_equal(b, cpu_st))
    algo = algorithmic_bytes(iters, levels, half, cfg["method"])
    pmc, _refused = pmc_profile(name, source_hash)
    valu = (pmc or {}).get("valu_insts_per_launch")
    return {
        "workload": f"{cfg['model']} KLT {cfg['method']}{' + consider_patch_luminance' if lum else ''}, {n} features, {cfg['width']}x{cfg['height']}, {levels}-level pyramid, "
                    f"{2 * half + 1}x{2 * half + 1} patch",
        "ms_per_step": ms, "features_per_s": n / (ms * 1e-3), "steps": steps, "regime": "warm" if n >= 4096 else "n/a (below 4096 features: list order)",
        "cold_ms": cold_ms, "roofline_frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo,
        "valu_frac": (valu / (ms * 1e-3) / VALU_PEAK_DATASHEET) if valu else None,
        "bit_identical": same((first_uv,), first_st) and same((last_uv,), last_st) and bool(np.array_equal(iters, cpu_it)),
        "mean_iterations_per_feature": float(iters.mean()), "cpu_ms": cpu_ms, "x_cpu": cpu_ms / ms,
    }


def matcher_config_leg(steps, ctx, dev, stream):
    """BASELINE.json configs[3], second half: BRIEF-256 brute-force ForceMatch, 10 000 x 10 000, back to back like the trackers."""
    import torch

    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import synth
    from tests import oracle_lib

    n_ref = n_cur = 10000
    ref, cur, _ = synth.make_descriptors(n_ref, n_cur, flips=20)
    d_ref = torch.from_numpy(F.pack_brief(ref).view(np.int32)).to(dev)
    d_cur = torch.from_numpy(F.pack_brief(cur).view(np.int32)).to(dev)
    d_idx = torch.full((n_ref,), -1, dtype=torch.int32, device=dev)
    t_warm, k = time.perf_counter(), 0
    while k < 4 or time.perf_counter() - t_warm < 10e-3:  # as in tracker_config_leg: out of the clock ramp that follows an idle device
        D.hamming_match_device(ctx, d_ref, d_cur, 256, 60.0, d_idx)
        k += 1
        if k % 8 == 0:
            stream.synchronize()
    stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        D.hamming_match_device(ctx, d_ref, d_cur, 256, 60.0, d_idx)
    stream.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    gpu_idx = d_idx.cpu().numpy()
    rows = 2000  # the oracle on a bounded sample of reference rows against ALL candidates
    t0 = time.perf_counter()
    ok, cidx = oracle_lib.force_match(ref[:rows], cur, 60.0)
    cpu_ms_full = (time.perf_counter() - t0) * 1e3 * n_ref / rows
    pairs = float(n_ref) * n_cur
    return {
        "workload": "BRIEF-256 ForceMatch, 10000 x 10000 descriptors (hamming_match_mfma_kernel + match_epilogue_kernel)",
        "ms_per_step": ms, "pairs_per_s": pairs / (ms * 1e-3), "descriptors_per_s": n_ref / (ms * 1e-3), "steps": steps, "regime": "n/a",
        "roofline_bound": "mfma (int8)", "roofline_frac": 2.0 * pairs * 256 / (ms * 1e-3) / 5.0e15, "valu_frac": None,
        "bit_identical": bool(np.array_equal(gpu_idx[:rows], cidx)), "checked_rows": rows, "matched": int((gpu_idx >= 0).sum()),
        "cpu_ms": cpu_ms_full, "x_cpu": cpu_ms_full / ms,
    }


REAL_PAIR = (os.path.join(ROOT, "tests", "data", "optical_flow", "ref_image.png"), os.path.join(ROOT, "tests", "data", "optical_flow", "cur_image.png"))


def real_image_features(ref_img, n, half, ctx):
    """The reference's own front end (test/test_optical_flow.cpp:34-39: Harris corners, kMinValidResponse 40) on the reference's own
    example frame, topped up to n on a jittered grid (the reference's test stops at 300 corners; a front end that keeps 2 000 tracks
    alive fills in weaker points the same way)."""
    import feature_tracker_amd as F

    det = F.FeaturePointHarrisDetector(ctx)
    det.options().kMinFeatureDistance, det.options().kMinValidResponse = 15, 40.0
    _, uv = det.DetectGoodFeatures(ref_img, n)
    rows, cols = ref_img.shape
    margin = 4 * half + 8
    k = np.arange(max(0, n - uv.shape[0]), dtype=np.int64)
    fill = np.stack([margin + (k * 37) % (cols - 2 * margin) + 0.25 * (k % 4), margin + (k * 53) % (rows - 2 * margin) + 0.5 * (k % 2)], axis=1).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([uv, fill], axis=0)[:n], dtype=np.float32), int(uv.shape[0])


def real_images_leg(steps, ctx, dev, stream):
    """The workload the reference's users have: the reference's example pair (test/test_optical_flow.cpp:31-32; the
    two PNGs are data files its tests load, committed under tests/data/), Harris corners topped up to 300 and 2 000, 13 x 13 patches,
    4 levels (test_optical_flow.cpp:24-27), every model x method.  Real frames hold features that never converge and run kMaxIteration
    iterations on every level, so a call of fewer than 4 096 features lasts as long as its slowest feature.  Per row: K back-to-back
    launches on device-resident buffers (ms_per_step, as the headline is timed), ONE synchronous host-vector call (ftk_klt_track, median),
    and the first launch's (u, v, status, iteration counts) against the oracle on the same inputs."""
    import ctypes as C

    import torch
    from PIL import Image

    import feature_tracker_amd as F
    from feature_tracker_amd import _native as NL
    from feature_tracker_amd import device as D
    from feature_tracker_amd import synth
    from tests import oracle_lib

    ref_img = np.ascontiguousarray(np.array(Image.open(REAL_PAIR[0]).convert("L"), dtype=np.uint8))
    cur_img = np.ascontiguousarray(np.array(Image.open(REAL_PAIR[1]).convert("L"), dtype=np.uint8))
    levels, half = 4, 6
    ref_levels, cur_levels = synth.build_pyramid(ref_img, levels), synth.build_pyramid(cur_img, levels)
    rp, cp = D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev)
    rows = {}
    t_leg = time.perf_counter()
    for n in (300, 2000):
        uv, n_harris = real_image_features(ref_img, n, half, ctx)
        d_ref = torch.from_numpy(uv).to(dev)
        d_in, d_st = d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev)
        outs = [(torch.empty_like(d_ref), torch.empty_like(d_st)) for _ in range(2)]
        d_it = torch.zeros(n, dtype=torch.int32, device=dev)
        for model in ("basic", "affine", "lssd"):
            for method in ("inverse", "direct", "fast"):
                opt = F.OpticalFlowOptions()
                opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, half, half, n
                klt = D.DeviceKlt(model, opt, rp, cp, ctx)
                klt.track(d_ref, d_in, d_st, outs[0][0], outs[0][1], d_it)
                stream.synchronize()
                iters = d_it.cpu().numpy().astype(np.uint32)
                g_uv, g_st = outs[0][0].cpu().numpy().copy(), outs[0][1].cpu().numpy().copy()
                launches = [klt.bind(d_ref, d_in, d_st, o[0], o[1], None) for o in outs]
                t_warm, k = time.perf_counter(), 0
                while k < 6 or time.perf_counter() - t_warm < 5e-3:  # out of the clock ramp that follows the oracle of the row before
                    launches[k & 1]()
                    k += 1
                    if k % 8 == 0:
                        stream.synchronize()
                stream.synchronize()
                t0 = time.perf_counter()
                for k in range(steps):
                    launches[k & 1]()
                stream.synchronize()
                ms = (time.perf_counter() - t0) / steps * 1e3
                l_uv, l_st = outs[(steps - 1) & 1][0].cpu().numpy(), outs[(steps - 1) & 1][1].cpu().numpy()
                # the literal call: host vectors in and out, one synchronous ftk_klt_track (what the C++ class does after its normalisation)
                h_ref, h_cur, h_st = uv.copy(), uv.copy(), np.zeros(n, dtype=np.uint8)
                o_n = opt.to_native()
                fn = NL.lib().ftk_klt_track
                a = (ctx.handle, NL.MODELS[model], C.byref(o_n), rp.handle, cp.handle, h_ref.ctypes.data_as(C.c_void_p), h_cur.ctypes.data_as(C.c_void_p),
                     h_st.ctypes.data_as(C.c_void_p), n, None, 0, 0, None)
                lat, rc = [], 0
                for k in range(24):
                    h_cur[:] = h_ref
                    h_st[:] = 0
                    t0 = time.perf_counter()
                    rc |= int(fn(*a))
                    if k >= 4:
                        lat.append(time.perf_counter() - t0)
                ok, c_uv, c_st, c_it = oracle_lib.klt_track_pyramid(model, ref_levels, cur_levels, uv, method=method, half=half, max_points=n)
                same = lambda u, s: bool(np.array_equal(u.view(np.uint32), c_uv.view(np.uint32)) and np.array_equal(s, c_st))
                rows[f"{model}_{method}_{n}"] = {
                    "ms_per_step": ms, "host_call_ms": float(np.median(lat)) * 1e3, "features_per_s": n / (ms * 1e-3), "tracked": int((g_st == 1).sum()),
                    "mean_iterations_per_feature": float(iters.mean()), "max_iterations_of_a_feature": int(iters.max()),
                    "features_at_max_iterations": int((iters >= levels * int(o_n.max_iteration)).sum()),
                    "bit_identical": bool(same(g_uv, g_st) and same(l_uv, l_st) and rc == 0 and same(h_cur, h_st) and np.array_equal(iters, c_it)),
                    "harris_corners": n_harris}
    return {"what": "the reference's example pair (tests/data/optical_flow/{ref,cur}_image.png, 752x480), Harris corners (min distance 15, min response 40) topped up to n on a "
                    "jittered grid, 13x13 patch, 4 levels, defaults otherwise; keys: model_method_n; ms_per_step = back-to-back device-resident launches, host_call_ms = one "
                    "synchronous ftk_klt_track with host vectors (median of 20); bit_identical = first and last launch, the host call and the iteration counts against the oracle",
            "steps": steps, "rows": rows, "all_bit_identical": all(r["bit_identical"] for r in rows.values()), "seconds_spent": time.perf_counter() - t_leg}


def host_call_leg(cfg, ref_levels, cur_levels, uv, expect_uv, expect_st):
    """SURVEY.md 8(d), literally: N / wall time of ONE TrackFeatures call — host vectors in, host vectors out, one synchronous
    ftk_klt_track (H2D of ref_uv / cur_uv / status, the tracker launch, D2H) with both pyramids already resident
    (optical_flow.cpp:6-26 takes ready-made pyramids).  Median over single synchronous calls; never the headline `value`."""
    import feature_tracker_amd as F

    klt = {"basic": F.OpticalFlowBasicKlt, "affine": F.OpticalFlowAffineKlt, "lssd": F.OpticalFlowLssdKlt}[cfg["model"]]()
    o = klt.options()
    o.kMethod, o.kPatchRowHalfSize, o.kPatchColHalfSize, o.kMaxTrackPointsNumber = cfg["method"], cfg["half"], cfg["half"], cfg["n"]
    rp, cp = F.ImagePyramid.from_host_levels(ref_levels), F.ImagePyramid.from_host_levels(cur_levels)
    for _ in range(5):
        ok, c, st = klt.TrackFeatures(rp, cp, uv)
    lat = []
    for _ in range(60):
        t0 = time.perf_counter()
        ok, c, st = klt.TrackFeatures(rp, cp, uv)
        lat.append(time.perf_counter() - t0)
    med = float(np.median(lat))
    # The same call as a C / C++ caller makes it: ftk_klt_track itself on prepared host arrays (what host/src/optical_flow.cpp does after
    # its input normalisation) — without the Python mirror's per-call numpy conversions, option marshalling and result arrays.
    import ctypes as C
    from feature_tracker_amd import _native as NL
    from feature_tracker_amd.tracker import default_context
    ctx = default_context()
    opt = klt.options().to_native()
    n = int(cfg["n"])
    h_ref = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    h_cur, h_st = h_ref.copy(), np.zeros(n, dtype=np.uint8)
    fn = NL.lib().ftk_klt_track
    args = (ctx.handle, klt._model, C.byref(opt), rp.handle, cp.handle, h_ref.ctypes.data_as(C.c_void_p), h_cur.ctypes.data_as(C.c_void_p),
            h_st.ctypes.data_as(C.c_void_p), n, None, 0, 0, None)
    lat_c, rc = [], 0
    for k in range(65):
        h_cur[:] = h_ref  # no prediction: cur = ref (optical_flow.cpp:12-14); the caller's preparation, outside the timed call
        h_st[:] = 0
        t0 = time.perf_counter()
        rc |= int(fn(*args))
        if k >= 5:
            lat_c.append(time.perf_counter() - t0)
    med_c = float(np.median(lat_c))
    return {"ms": med * 1e3, "features_per_s": cfg["n"] / med, "calls": len(lat), "p10_ms": float(np.percentile(lat, 10)) * 1e3, "p90_ms": float(np.percentile(lat, 90)) * 1e3,
            "bit_identical_to_resident_path": bool(ok and np.array_equal(c.view(np.uint32), expect_uv.view(np.uint32)) and np.array_equal(st, expect_st)),
            "what": "one synchronous ftk_klt_track per call through the Python mirror of feature_tracker.h (host float vectors in and out over PCIe, "
                    "pyramids resident in HBM); c_abi_*: the same call on prepared host arrays, as the C++ class makes it",
            "c_abi_ms": med_c * 1e3, "c_abi_features_per_s": cfg["n"] / med_c, "c_abi_p10_ms": float(np.percentile(lat_c, 10)) * 1e3,
            "c_abi_p90_ms": float(np.percentile(lat_c, 90)) * 1e3,
            "c_abi_bit_identical_to_resident_path": bool(rc == 0 and np.array_equal(h_cur.view(np.uint32), expect_uv.view(np.uint32)) and np.array_equal(h_st, expect_st))}


def sharded_steps(slots, d_ref, d_in, d_st, steps, collective):
    """The plain per-step loop of the strong-scaling path: step k launches this rank's block into result slot k & 1 and
    all-gathers that slot's packed shard.  Shared by run_sharded (GPU) and the CPU rehearsal of --dry-launch --shard-total."""
    for k in range(steps):
        slots[k & 1].launch_local(d_ref, d_in, d_st)
        slots[k & 1].gather(force_collective=collective)


CONFIG5_TOTAL = 200000  # BASELINE.json configs[4]: 200 000 features, 1920x1080, 4 levels, sharded over the GPUs of the node


def config5_sharded_leg(steps, warmup, world, rank, dev, make_tracker, native=None, total=CONFIG5_TOTAL, use_graph=True):
    """BASELINE.json configs[4] — the one configuration that IS multi-GPU — measured by every `bench.py --gpus N` run with N > 1 (or with
    FTK_BENCH_FORCE_DIST=1 at N = 1) without further flags: `total` features block-sharded over the ranks, ONE all-gather of the packed
    (uv, status) shards per step, strong scaling.  Runs on EVERY rank (it contains collectives) after the weak-scaling region and returns
    the `config5_sharded` object: per-step time (max over ranks), the kernel and the all-gather timed separately, gathered == unsharded.

    Device-agnostic on purpose: `--dry-launch` calls this same function over gloo with a stand-in tracker on the CPU, so that the step
    loop, the slots, the timing reductions and the object's schema have met N > 1 ranks before the first 8-GPU lease (tests/
    test_bench_launch_cpu.py).  make_tracker(total) -> (tracker, d_ref, d_in, d_st, unsharded(d_uv_out, d_st_out)); native(total, d_ref,
    d_in, d_st) -> dict for the C-ABI path (ftk_klt_track_sharded_device: RCCL issued by libftk_hip.so), or None where there is no device."""
    import torch
    import torch.distributed as dist

    from feature_tracker_amd import dist as FD

    gpu = torch.device(dev).type == "cuda"
    sync = torch.cuda.synchronize if gpu else (lambda: None)

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(fn, count):
        """fn() `count` times back to back, barrier + synchronise on both sides, max over ranks; seconds per call."""
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        for k in range(count):
            fn(k)
        sync()
        dist.barrier()
        return max_over_ranks(time.perf_counter() - t0) / count

    tracker, d_ref, d_in, d_st, unsharded = make_tracker(total)
    slots = [FD.ShardedKlt(tracker, total, dev, world, rank) for _ in range(2)]  # two alternating result slots
    sharded_steps(slots, d_ref, d_in, d_st, max(2, warmup), True)
    sync()
    graph = None
    if gpu and use_graph and steps >= 2 and os.environ.get("FTK_BENCH_NO_GRAPH") != "1":
        # as in the weak-scaling region: the K steps captured once, gather k on a side stream beside kernel k + 1; every rank must take
        # the same path, so capture success is all-reduced
        ok = 1
        stream = torch.cuda.current_stream()
        try:
            torch.cuda.synchronize()
            time.sleep(0.5)
            side = torch.cuda.Stream(device=dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                cap_stream = torch.cuda.current_stream()
                gather_done = {}
                for k in range(steps):
                    slot = slots[k & 1]
                    if k >= 2:
                        cap_stream.wait_event(gather_done[k - 2])
                    slot.launch_local(d_ref, d_in, d_st)
                    kernel_done = torch.cuda.Event()
                    kernel_done.record(cap_stream)
                    side.wait_event(kernel_done)
                    with torch.cuda.stream(side):
                        slot.gather(force_collective=True)
                        gather_done[k] = torch.cuda.Event()
                        gather_done[k].record(side)
                cap_stream.wait_stream(side)
            graph = g
        except Exception as exc:
            ok, graph = 0, None
            if rank == 0:
                print(f"bench.py: config5_sharded: graph capture unavailable ({type(exc).__name__}: {exc}); using the per-step loop", file=sys.stderr)
        torch.cuda.synchronize()
        agree = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 0:
            graph = None
        if graph is not None:
            graph.replay()
            torch.cuda.synchronize()
    if graph is not None:
        per_step = timed(lambda k: graph.replay(), 1) / steps
    else:
        per_step = timed(lambda k: (slots[k & 1].launch_local(d_ref, d_in, d_st), slots[k & 1].gather(force_collective=True)), steps)
    loop_step = per_step if graph is None else timed(lambda k: (slots[k & 1].launch_local(d_ref, d_in, d_st), slots[k & 1].gather(force_collective=True)), steps)
    kernel_s = timed(lambda k: slots[k & 1].launch_local(d_ref, d_in, d_st), steps)
    gather_s = timed(lambda k: slots[k & 1].gather(force_collective=True), steps)
    # every rank holds every rank's shard: the gathered result must equal ONE unsharded launch over all features on this rank
    want_uv, want_st = torch.empty_like(d_ref), torch.empty_like(d_st)
    unsharded(want_uv, want_st)
    sync()
    same = True
    for sl in slots:
        guv, gst = FD.unpack_gathered(sl.gathered, total, world)
        same = same and bool(torch.equal(guv.view(torch.int32), want_uv.view(torch.int32)) and torch.equal(gst, want_st))
    flag = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    tracked = float((want_st == 1).float().mean().item())
    cap = FD.shard_capacity(total, world)
    out = {
        "workload": f"BASELINE configs[4]: {total} features in total block-sharded x{world}, one all-gather of packed (uv, status) per step; strong scaling",
        "total_features": int(total), "features_per_rank": int(cap), "packed_bytes_per_rank": int(FD.packed_bytes(cap)), "rccl_ranks": int(dist.get_world_size()),
        "backend": dist.get_backend(), "steps": int(steps),
        "torch": {"ms_per_step": per_step * 1e3, "features_per_s": total / per_step, "ms_per_step_plain_loop": loop_step * 1e3,
                  "kernel_us": kernel_s * 1e6, "all_gather_us": gather_s * 1e6, "graph": graph is not None,
                  "gathered_equals_unsharded_bitwise": bool(flag.item() == 1), "tracked_fraction": tracked,
                  "what": "feature_tracker_amd.dist.ShardedKlt: ftk_klt_track_device on this rank's block + torch.distributed all_gather_into_tensor; "
                          "every figure is the max over ranks between barriers; kernel_us / all_gather_us: the two halves of a step timed alone, back to back"},
    }
    if native is not None:
        try:
            out["native_comm"] = native(total, d_ref, d_in, d_st, want_uv, want_st, timed)
        except Exception as exc:  # reported, never fatal for the line (every rank takes the same branch: the failure modes are local set-up errors)
            out["native_comm"] = {"error": f"{type(exc).__name__}: {exc}"}
    else:
        out["native_comm"] = {"skipped": "no device on this path (dry launch)"}
    return out


def _config5_device_parts(world, rank, local_rank, dev, ctx):
    """make_tracker / native for config5_sharded_leg on a HIP device (bench.py main and run_native_comm)."""
    import torch
    import torch.distributed as dist

    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import synth

    cfg = dict(synth.CONFIGS["config5_shard"])
    w, h, levels, half = cfg["width"], cfg["height"], cfg["levels"], cfg["half"]
    state = {}

    def make_tracker(total):
        ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1))
        ref_levels, cur_levels = synth.build_pyramid(ref_img, levels), synth.build_pyramid(cur_img, levels)
        uv = synth.make_features(total, w, h, half=half)  # the same full list on every rank
        opt = F.OpticalFlowOptions()
        opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = cfg["method"], half, half, total
        klt = D.DeviceKlt(cfg["model"], opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
        d_ref = torch.from_numpy(uv).to(dev)
        d_in, d_st = d_ref.clone(), torch.zeros(total, dtype=torch.uint8, device=dev)
        state["klt"] = klt
        return klt, d_ref, d_in, d_st, (lambda o_uv, o_st: klt.track(d_ref, d_in, d_st, o_uv, o_st, None))

    def native(total, d_ref, d_in, d_st, want_uv, want_st, timed):
        from feature_tracker_amd import _native as NL

        klt = state["klt"]
        stream = torch.cuda.current_stream()
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(D.Comm.unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, src=0)
        stream.synchronize()
        comm = D.Comm(ctx, rank, world, bytes(uid.cpu().numpy().tobytes()))
        try:
            d_out, d_sto = torch.zeros_like(d_ref), torch.zeros_like(d_st)
            launch = klt.bind_sharded(comm, d_ref, d_in, d_st, d_out, d_sto)
            for _ in range(3):
                launch()
            steps = state.get("steps", 20)
            per_step = timed(lambda k: launch(), steps)
            torch.cuda.synchronize()
            same = bool(torch.equal(d_out.view(torch.int32), want_uv.view(torch.int32)) and torch.equal(d_sto, want_st))
            flag = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return {"ms_per_step": per_step * 1e3, "features_per_s": total / per_step, "rccl_ranks": int(NL.lib().ftk_comm_world(comm.handle)),
                    "gathered_equals_unsharded_bitwise": bool(flag.item() == 1), "steps": steps,
                    "what": "ftk_klt_track_sharded_device per step: tracker kernel on this rank's block, ncclAllGather issued by libftk_hip.so on the context "
                            "stream, scatter kernel; plain per-step loop; max over ranks between barriers"}
        finally:
            comm.close()

    return make_tracker, native, state


def run_config5_leg(args, world, rank, local_rank, dev, ctx, stream, out):
    """The config5_sharded object of an N > 1 (or forced-dist) run, under a deadline: if the leg does not finish (a collective that never
    completes on a first 8-GPU lease), rank 0 still prints the weak-scaling line — with the failure named — and every rank leaves."""
    import threading

    import torch

    def expire():
        if rank == 0 and out is not None:
            out["config5_sharded"] = {"error": f"did not finish within {args.config5_timeout:.0f} s; the weak-scaling figures above are complete"}
            print(json.dumps(out), flush=True)
        os._exit(0 if rank == 0 else 5)

    guard = threading.Timer(args.config5_timeout, expire)
    guard.daemon = True
    guard.start()
    try:
        with torch.cuda.stream(stream):
            make_tracker, native, state = _config5_device_parts(world, rank, local_rank, dev, ctx)
            steps = max(5, min(args.steps, 50))
            state["steps"] = steps
            leg = config5_sharded_leg(steps, min(args.warmup, 5), world, rank, dev, make_tracker, native, total=args.config5_total)
    finally:
        guard.cancel()
    return leg


def run_sharded(args, cfg, world, rank, local_rank, dev, use_dist):
    """Strong scaling: args.shard_total features of the workload's geometry, sharded over the ranks, one all-gather per step."""
    import torch
    import torch.distributed as dist

    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import dist as FD
    from feature_tracker_amd import synth

    n, w, h, levels, half = args.shard_total, cfg["width"], cfg["height"], cfg["levels"], cfg["half"]
    ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1))
    ref_levels, cur_levels = synth.build_pyramid(ref_img, levels), synth.build_pyramid(cur_img, levels)
    uv = synth.make_features(n, w, h, half=half)  # the same full list on every rank
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, local_rank)
        opt = F.OpticalFlowOptions()
        opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = cfg["method"], half, half, n
        klt = D.DeviceKlt(cfg["model"], opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
        slots = [FD.ShardedKlt(klt, n, dev, world, rank) for _ in range(2)]  # two alternating result slots
        sharded = slots[0]
        d_ref = torch.from_numpy(uv).to(dev)
        d_in = d_ref.clone()
        d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
        sharded_steps(slots, d_ref, d_in, d_st, max(2, args.warmup), use_dist)
        stream.synchronize()
        # As in the weak-scaling path: with a collective per step the K steps are captured ONCE into a HIP graph in which the
        # gather of step k runs on a side stream beside the kernel of step k + 1 (the kernel of step k + 2 waits for the gather
        # of step k, which reads that slot's packed shard); capture success is all-reduced so that every rank takes the same path.
        graph = None
        if use_dist and args.steps >= 2 and os.environ.get("FTK_BENCH_NO_GRAPH") != "1":
            ok = 1
            try:
                # quiesce first: every collective issued so far has completed AND the process group's watchdog has had time to
                # retire their work objects, so that it holds nothing to poll while the capture runs
                torch.cuda.synchronize()
                time.sleep(0.5)
                side = torch.cuda.Stream(device=dev)
                g = torch.cuda.CUDAGraph()
                # thread_local: the process group's watchdog thread polls its events with hipEventQuery while we capture; in the
                # default (global) mode such a call from ANY thread invalidates the capture — an intermittent failure that grows
                # with the length of the capture
                with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                    gather_done = {}
                    for k in range(args.steps):
                        slot = slots[k & 1]
                        if k >= 2:
                            stream.wait_event(gather_done[k - 2])
                        slot.launch_local(d_ref, d_in, d_st)
                        kernel_done = torch.cuda.Event()
                        kernel_done.record(stream)
                        side.wait_event(kernel_done)
                        with torch.cuda.stream(side):
                            slot.gather(force_collective=True)
                            gather_done[k] = torch.cuda.Event()
                            gather_done[k].record(side)
                    stream.wait_stream(side)
                graph = g
            except Exception as exc:  # capture is not available for this collective / runtime: plain loop
                ok = 0
                graph = None
                if rank == 0:
                    print(f"bench.py: graph capture unavailable ({type(exc).__name__}: {exc}); using the per-step loop", file=sys.stderr)
            torch.cuda.synchronize()
            agree = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) == 0:
                graph = None
            if graph is not None:
                graph.replay()  # one untimed replay: first-use initialisation of the instantiated graph
                torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if graph is not None:
            graph.replay()
        else:
            sharded_steps(slots, d_ref, d_in, d_st, args.steps, use_dist)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        guv, gst = FD.unpack_gathered(sharded.gathered, n, world)
        tracked = float((gst == 1).float().mean().item())
        # result check: the gathered result must equal ONE unsharded launch over all n features on this rank, bit for bit
        d_all, d_all_st = torch.empty_like(d_ref), torch.empty_like(d_st)
        klt.track(d_ref, d_in, d_st, d_all, d_all_st, None)
        torch.cuda.synchronize()
        gathered_ok = bool(torch.equal(guv.view(torch.int32), d_all.view(torch.int32)) and torch.equal(gst, d_all_st))
        assert gathered_ok, "gathered sharded result differs from the unsharded launch"
    rccl_ranks = int(dist.get_world_size()) if use_dist else 1
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "tracked features/sec (sharded)", "value": n * args.steps / elapsed, "unit": "tracked features/s", "n_gpus": world,
            "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg['model']} KLT {cfg['method']}, {n} features in total sharded x{world}, {w}x{h}, {levels}-level pyramid, "
                                   f"{2 * half + 1}x{2 * half + 1} patch", "tracked_fraction": tracked,
                       "gathered_equals_unsharded_bitwise": gathered_ok,
                       "parallelism": (f"features sharded x{world}, pyramids replicated, one RCCL all-gather of packed (uv,status) per step"
                                       + (", steps captured in one HIP graph (gather k overlaps kernel k+1)" if graph is not None else ""))}}), flush=True)
    if use_dist:
        dist.destroy_process_group()


def run_native_comm(args, cfg, world, rank, local_rank, dev):
    """The contractual weak-scaling workload (cfg['n'] features per GPU) through the NATIVE multi-GPU entry point: every step is
    ONE ftk_klt_track_sharded_device call — tracker kernel on this rank's block, RCCL ncclAllGather issued by libftk_hip.so on the
    context stream, scatter kernel.  torch.distributed is used only to hand the RCCL unique id from rank 0 to the others."""
    import torch
    import torch.distributed as dist

    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import synth

    n_local, w, h, levels, half = cfg["n"], cfg["width"], cfg["height"], cfg["levels"], cfg["half"]
    n = n_local * world
    ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1))
    ref_levels, cur_levels = synth.build_pyramid(ref_img, levels), synth.build_pyramid(cur_img, levels)
    # the global feature list, the same on every rank: rank r's block is the list the torch.distributed path gives rank r
    uv = np.concatenate([synth.make_features(n_local, w, h, seed=12345 + r, half=half) for r in range(world)], axis=0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, local_rank)
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(D.Comm.unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(uid, src=0)
        stream.synchronize()
        comm = D.Comm(ctx, rank, world, bytes(uid.cpu().numpy().tobytes()))
        opt = F.OpticalFlowOptions()
        opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = cfg["method"], half, half, n
        klt = D.DeviceKlt(cfg["model"], opt, D.upload_pyramid(ref_levels, ctx, dev), D.upload_pyramid(cur_levels, ctx, dev), ctx)
        d_ref = torch.from_numpy(uv).to(dev)
        d_in, d_st = d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev)
        d_out, d_sto = torch.empty_like(d_ref), torch.empty_like(d_st)
        launch = klt.bind_sharded(comm, d_ref, d_in, d_st, d_out, d_sto)
        for _ in range(max(1, args.warmup)):
            launch()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            launch()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        # result check: this rank's block of the gathered result == its own unsharded launch on that block
        b, e = D.shard_bounds(n, world, rank)
        d_chk, d_chk_st = torch.empty(e - b, 2, dtype=torch.float32, device=dev), torch.empty(e - b, dtype=torch.uint8, device=dev)
        klt.track(d_ref[b:e], d_in[b:e], d_st[b:e], d_chk, d_chk_st, None)
        torch.cuda.synchronize()
        ok = bool(torch.equal(d_out[b:e].view(torch.int32), d_chk.view(torch.int32)) and torch.equal(d_sto[b:e], d_chk_st))
        assert ok, "gathered block differs from the local launch"
        tracked = float((d_sto == 1).float().mean().item())
        from feature_tracker_amd import _native as NL
        rccl_ranks = int(NL.lib().ftk_comm_world(comm.handle))  # the communicator libftk_hip.so itself holds
        comm.close()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    out = None
    if rank == 0:
        out = {
            "metric": "tracked features/sec (2000 pts, 640x480, 4-lvl pyr)", "value": n * args.steps / elapsed, "unit": "tracked features/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['model']} KLT {cfg['method']}, {n_local} features/GPU, {w}x{h}, {levels}-level pyramid, "
                                   f"{2 * half + 1}x{2 * half + 1} patch",
                       "parallelism": f"features sharded x{world}, pyramids replicated, one ncclAllGather of packed (uv,status) per step issued by "
                                      "libftk_hip.so (ftk_klt_track_sharded_device) + scatter kernel; plain per-step loop",
                       "tracked_fraction": tracked, "gathered_block_equals_local_launch": ok}}
    if dist.is_initialized() and not args.no_config5_leg:
        leg = run_config5_leg(args, world, rank, local_rank, dev, ctx, stream, out)
        if out is not None:
            out["config5_sharded"] = leg
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def _free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_ranks: int, argv, timeout_s: float) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as CHILD processes (one per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1) and return the worst exit code.
    This parent never imports torch and never touches a GPU (no exec of a GPU-initialised process); rank 0's JSON line
    goes to the inherited stdout, the other ranks print nothing there."""
    import subprocess

    port = int(os.environ.get("MASTER_PORT") or _free_port())
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FTK_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    deadline = time.monotonic() + timeout_s
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
        if live and (rc != 0 or time.monotonic() > deadline):
            # one rank failed (the others would wait in a collective for ever) or the launch timed out: end exactly the
            # processes started above
            if rc == 0:
                rc = 124
                print(f"bench.py: the {n_ranks}-rank run did not finish within {timeout_s:.0f} s", file=sys.stderr, flush=True)
            for p in live:
                p.terminate()
            t_kill = time.monotonic() + 10.0
            for p in live:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.monotonic()))
                except Exception:
                    p.kill()
                    p.wait()
            live = []
        elif live:
            time.sleep(0.05)
    return rc


class _DryTracker:
    """Rehearsal scaffolding of --dry-launch only: a stand-in for the device tracker that writes a known function of the GLOBAL feature
    index, so that the sharding bookkeeping around it can be checked over gloo on the CPU.  Never part of a measurement."""

    def __init__(self, cap):
        self.max_track_points = int(cap)

    def track(self, ref_uv, cur_uv_in, status_in, cur_uv_out, status_out, iters, max_track_points=None):
        import torch

        m = ref_uv.shape[0]
        limit = m if max_track_points is None else int(max_track_points)
        tracked = torch.arange(m) < limit
        cur_uv_out.copy_(torch.where(tracked[:, None], ref_uv * 2.0 + 1.0, cur_uv_in))
        status_out.copy_(torch.where(tracked, (ref_uv[:, 0].to(torch.int64) % 3).to(torch.uint8), status_in))


def dry_launch(args, world: int, rank: int) -> None:
    """--dry-launch: the rank plumbing of an N-rank run without any device work (gloo on the CPU): every rank joins the
    process group, one all-reduce counts the ranks, one all-gather moves a packed result shard of the workload's size, and rank 0
    prints the line's launch fields.  This is what the CPU test of `bench.py --gpus 2` runs."""
    import torch
    import torch.distributed as dist

    from feature_tracker_amd import dist as FD
    from feature_tracker_amd import synth

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ones = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(ones)
    n = synth.CONFIGS[args.workload]["n"]
    packed = torch.full((FD.packed_bytes(n),), rank, dtype=torch.uint8)
    gathered = FD.all_gather_results(packed, world, force_collective=True)
    per = FD.packed_bytes(n)
    shards_ok = all(bool((gathered[r * per:(r + 1) * per] == r).all()) for r in range(world))
    sharded = None
    if args.shard_total > 0:
        # The strong-scaling path's bookkeeping (run_sharded: block bounds, per-rank capacity, packed shards in two alternating
        # slots, one all-gather per step, unpacking in global feature order, the GLOBAL kMaxTrackPointsNumber) with N > 1 ranks
        # before the first hardware run: the same ShardedKlt objects and the same step loop over gloo, around a stand-in for the
        # device tracker that writes a known function of the GLOBAL feature index (rehearsal scaffolding of this flag only).
        total, cap_global = args.shard_total, args.shard_total - args.shard_total // 7  # a cap that cuts into the last blocks

        idx = torch.arange(total, dtype=torch.float32)
        d_ref = torch.stack([idx, -idx], dim=1).contiguous()
        d_in, d_st = d_ref + 0.5, torch.full((total,), 9, dtype=torch.uint8)
        slots = [FD.ShardedKlt(_DryTracker(cap_global), total, "cpu", world, rank) for _ in range(2)]
        sharded_steps(slots, d_ref, d_in, d_st, max(2, args.steps), True)
        results = [FD.unpack_gathered(sl.gathered, total, world) for sl in slots]
        want_uv, want_st = torch.empty_like(d_ref), torch.empty_like(d_st)
        _DryTracker(cap_global).track(d_ref, d_in, d_st, want_uv, want_st, None, max_track_points=cap_global)
        bounds = [FD.shard_bounds(total, world, r) for r in range(world)]
        sharded = {"total": total, "global_cap": cap_global, "capacity": FD.shard_capacity(total, world), "packed_bytes": FD.packed_bytes(FD.shard_capacity(total, world)),
                   "blocks_cover_the_list": bounds[0][0] == 0 and bounds[-1][1] == total and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1)),
                   "gathered_equals_unsharded": all(bool(torch.equal(uv, want_uv) and torch.equal(st, want_st)) for uv, st in results)}
        flag = torch.tensor([1 if (sharded["gathered_equals_unsharded"] and sharded["blocks_cover_the_list"]) else 0], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # every rank holds every rank's result: all must agree
        sharded["all_ranks_agree"] = bool(flag.item() == 1)
    # The config5_sharded object of an N > 1 run (BASELINE configs[4]), through the SAME function the GPU run calls — step loop, result
    # slots, timing reductions, the gathered == unsharded check, the schema — around the stand-in tracker.
    config5 = None
    if not args.no_config5_leg:
        def make_dry(total):
            idx = torch.arange(total, dtype=torch.float32)
            d_ref = torch.stack([idx, -idx], dim=1).contiguous()
            d_in, d_st = d_ref + 0.5, torch.full((total,), 9, dtype=torch.uint8)
            trk = _DryTracker(total)
            return trk, d_ref, d_in, d_st, (lambda o_uv, o_st: trk.track(d_ref, d_in, d_st, o_uv, o_st, None))

        config5 = config5_sharded_leg(max(2, min(args.steps, 5)), 1, world, rank, "cpu", make_dry, None, total=args.config5_total)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "rccl_ranks": int(dist.get_world_size()), "ranks_counted": int(ones.item()),
                          "backend": "gloo", "all_gather_ok": shards_ok, "self_launched": os.environ.get("FTK_BENCH_SELF_LAUNCHED") == "1",
                          "steps": args.steps, "warmup": args.warmup, "sharded": sharded, "config5_sharded": config5}), flush=True)
    dist.destroy_process_group()
    if not shards_ok or int(ones.item()) != world or (sharded is not None and not sharded["all_ranks_agree"]):
        raise SystemExit(4)
    if config5 is not None and not config5["torch"]["gathered_equals_unsharded_bitwise"]:
        raise SystemExit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config2", help="feature_tracker_amd.synth.CONFIGS key")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tree-leg", action="store_true", help="skip the throughput_mode measurement (N = 1 only; it runs after the timed region)")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the with_pyramid_upload measurement (N = 1 only; it runs after the timed region)")
    ap.add_argument("--no-configs-leg", action="store_true", help="skip the `configs` object (the other BASELINE configurations, N = 1 only; after the timed region)")
    ap.add_argument("--no-host-call-leg", action="store_true", help="skip the `host_call` object (one synchronous host-vector TrackFeatures call; N = 1 only)")
    ap.add_argument("--no-real-images-leg", action="store_true", help="skip the `real_images` object (the reference's example pair, every variant; N = 1 only)")
    ap.add_argument("--shard-total", type=int, default=0,
                    help="strong-scaling variant (BASELINE.json configs[4] style): this many features in total, block-sharded over the "
                         "ranks with feature_tracker_amd.dist.ShardedKlt; 0 = the contractual weak-scaling workload")
    ap.add_argument("--features", type=int, default=0, help="experiment knob: override the workload's feature count (the reported config says so)")
    ap.add_argument("--native-comm", action="store_true",
                    help="N > 1 (or N = 1 for a rehearsal): every step through ftk_klt_track_sharded_device — RCCL issued by the C ABI — instead "
                         "of torch.distributed's all_gather_into_tensor; same workload, same metric ($FTK_BENCH_NATIVE_COMM=1 selects it too)")
    ap.add_argument("--prewarm-seconds", type=float, default=0.0,
                    help="experiment knob: keep the device busy with untimed steps for this long before the W warmup steps (clock ramp study)")
    ap.add_argument("--no-config5-leg", action="store_true", help="N > 1 (or FTK_BENCH_FORCE_DIST=1): skip the `config5_sharded` object")
    ap.add_argument("--config5-total", type=int, default=CONFIG5_TOTAL, help="features in total of the config5_sharded leg (BASELINE configs[4]: 200 000)")
    ap.add_argument("--config5-timeout", type=float, default=300.0, help="seconds the config5_sharded leg may take before the line is printed without it")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rank plumbing only: gloo on the CPU, no device work (CPU test of the N-rank launch)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched N > 1 runs: seconds before the ranks are ended")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # Who starts the ranks.  Under a launcher (torchrun: WORLD_SIZE is set) this process IS one rank.  Without one,
    # `--gpus N > 1` starts its own N ranks as children BEFORE anything touches torch or HIP, so that the plain
    # `python bench.py --gpus 8` measures eight GPUs.  Any other disagreement between --gpus and the world is an error:
    # a run must never print a line whose n_gpus differs from what was asked for.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], args.launch_timeout))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run (the reported n_gpus would be wrong)")
    if args.dry_launch:
        return dry_launch(args, world, rank)

    import torch
    import torch.distributed as dist

    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import dist as FD
    from feature_tracker_amd import synth

    if torch.cuda.device_count() < world:  # counting devices does not initialise the GPU
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} HIP device(s) visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("FTK_BENCH_FORCE_DIST") == "1"  # the latter: exercise the RCCL path at world size 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = dict(synth.CONFIGS[args.workload])
    if args.features > 0:
        cfg["n"] = args.features
    if args.shard_total > 0:
        return run_sharded(args, cfg, world, rank, local_rank, dev, use_dist)
    if args.native_comm or os.environ.get("FTK_BENCH_NATIVE_COMM") == "1":
        return run_native_comm(args, cfg, world, rank, local_rank, dev)
    n, w, h, levels, half = cfg["n"], cfg["width"], cfg["height"], cfg["levels"], cfg["half"]
    if cfg["model"] == "basic":
        ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1))
    else:
        ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1), rotation_deg=1.5, scale=1.02)
    ref_levels = synth.build_pyramid(ref_img, levels)
    cur_levels = synth.build_pyramid(cur_img, levels)
    # weak scaling: every rank tracks its own n features (different seeds), pyramids replicated
    uv = synth.make_features(n, w, h, seed=12345 + rank, half=half)

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, local_rank)
        ref_pyr = D.upload_pyramid(ref_levels, ctx, dev)
        cur_pyr = D.upload_pyramid(cur_levels, ctx, dev)
        opt = F.OpticalFlowOptions()
        opt.kMethod = cfg["method"]
        opt.kPatchRowHalfSize = opt.kPatchColHalfSize = half
        opt.kMaxTrackPointsNumber = n
        klt = D.DeviceKlt(cfg["model"], opt, ref_pyr, cur_pyr, ctx)

        d_ref = torch.from_numpy(uv).to(dev)
        d_cur_in = d_ref.clone()  # no prediction: cur = ref (optical_flow.cpp:12-14)
        d_st_in = torch.zeros(n, dtype=torch.uint8, device=dev)
        # two alternating result slots
        packed2 = [torch.zeros(FD.packed_bytes(n), dtype=torch.uint8, device=dev) for _ in range(2)]
        views2 = [FD.pack_views(pk, n) for pk in packed2]
        gathered2 = [torch.empty(FD.packed_bytes(n) * world, dtype=torch.uint8, device=dev) for _ in range(2)]
        packed = packed2[0]
        d_cur_out, d_st_out = views2[0]
        d_iters = torch.zeros(n, dtype=torch.int32, device=dev)

        def step(with_iters=False):
            klt.track(d_ref, d_cur_in, d_st_in, d_cur_out, d_st_out, d_iters if with_iters else None)
            if use_dist:
                return FD.all_gather_results(packed, world, force_collective=True)
            return packed

        step(with_iters=True)
        stream.synchronize()
        iters = d_iters.cpu().numpy().astype(np.uint32)
        status = d_st_out.cpu().numpy()
        if args.prewarm_seconds > 0:
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < args.prewarm_seconds:
                for _ in range(100):
                    step()
                stream.synchronize()
        for _ in range(args.warmup):
            step()
        stream.synchronize()

        first_uv, first_st = d_cur_out.cpu().numpy().copy(), status.copy()  # the first step's result, for the parity object
        launches = [klt.bind(d_ref, d_cur_in, d_st_in, views2[slot][0], views2[slot][1], None) for slot in range(2)]
        # kernel-only duration (roofline objects): HIP events on the launch stream around the same launches on the same
        # data in a SEPARATE pass after the timed region — an event pair costs about a tenth of this kernel in queue
        # time, so none is recorded between the K timed steps at any step count.
        n_ev = max(10, min(args.steps, 50))
        ev0 = {k: torch.cuda.Event(enable_timing=True) for k in range(n_ev)}
        ev1 = {k: torch.cuda.Event(enable_timing=True) for k in range(n_ev)}

        # N > 1: every step is the tracker kernel followed by the RCCL all-gather of its packed result shard.  The
        # K steps are captured ONCE into a HIP graph in which the gather of step k runs on a side stream beside the
        # kernel of step k + 1 (two result slots; the kernel of step k + 2 waits for the gather of step k), and the
        # timed region replays that graph: the same K kernels and K collectives, without a Python round trip per
        # step.  All ranks must agree on the mode (a collective inside a graph on one rank and outside on another
        # would deadlock), so capture success is all-reduced; if any rank cannot capture, every rank runs the plain loop.
        graph = None
        if use_dist and args.steps >= 2 and os.environ.get("FTK_BENCH_NO_GRAPH") != "1":
            ok = 1
            try:
                # quiesce first: every collective issued so far has completed AND the process group's watchdog has had time to
                # retire their work objects, so that it holds nothing to poll while the capture runs
                torch.cuda.synchronize()
                time.sleep(0.5)
                side = torch.cuda.Stream(device=dev)
                g = torch.cuda.CUDAGraph()
                # thread_local: the process group's watchdog thread polls its events with hipEventQuery while we capture; in the
                # default (global) mode such a call from ANY thread invalidates the capture — an intermittent failure that grows
                # with the length of the capture
                with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                    gather_done = {}
                    for k in range(args.steps):
                        slot = k & 1
                        if k >= 2:
                            stream.wait_event(gather_done[k - 2])  # the slot's previous gather has read packed2[slot]
                        launches[slot]()
                        kernel_done = torch.cuda.Event()
                        kernel_done.record(stream)
                        side.wait_event(kernel_done)
                        with torch.cuda.stream(side):
                            dist.all_gather_into_tensor(gathered2[slot], packed2[slot])
                            gather_done[k] = torch.cuda.Event()
                            gather_done[k].record(side)
                    stream.wait_stream(side)
                graph = g
            except Exception as exc:  # capture is not available for this collective / runtime: plain loop
                ok = 0
                graph = None
                if rank == 0:
                    print(f"bench.py: graph capture unavailable ({type(exc).__name__}: {exc}); using the per-step loop", file=sys.stderr)
            torch.cuda.synchronize()
            agree = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) == 0:
                graph = None
            if graph is not None:
                # one untimed replay (first-use initialisation of the instantiated graph) under a watchdog: collectives
                # inside a graph that never complete would otherwise hold the node until the caller's own timeout
                import threading

                def _abort():
                    print("bench.py: the graph replay did not finish within 180 s; aborting this rank", file=sys.stderr, flush=True)
                    os._exit(3)
                guard = threading.Timer(180.0, _abort)
                guard.daemon = True
                guard.start()
                graph.replay()
                torch.cuda.synchronize()
                guard.cancel()

        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if graph is not None:
            graph.replay()
        else:
            for k in range(args.steps):
                slot = k & 1
                launches[slot]()
                if use_dist:
                    FD.all_gather_results(packed2[slot], world, force_collective=True, out=gathered2[slot])
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        # kernel duration for the roofline objects: a short eager pass right after the timed region (same launches, same data)
        for k in ev0:
            ev0[k].record(stream)
            launches[k & 1]()
            ev1[k].record(stream)
        torch.cuda.synchronize()
        # The kernel's average launch duration for the roofline objects: ONE event pair around a batch of back-to-back launches
        # (no event between them), divided by the batch size.  It contains the inter-launch gap, so it can only OVERSTATE the
        # kernel time (an event pair around a single launch adds ~5 us of marker cost: that figure is given beside it, and
        # subtracting a separately measured "empty pair" over-corrects — it came out 10 % BELOW the rocprofv3 trace average).
        batch0, batch1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        batch_n = max(20, min(args.steps, 200))
        batch0.record(stream)
        for k in range(batch_n):
            launches[k & 1]()
        batch1.record(stream)
        torch.cuda.synchronize()
        kernel_ms_batch = batch0.elapsed_time(batch1) / batch_n
        # Throughput mode (ftk_set_reduction_mode(TREE)): the same launches with butterfly sums instead of the exact-order chains.
        # Reported beside the contract path — what bit-exactness costs — never asserted and never the headline.
        tree = None
        if world == 1 and not args.no_tree_leg:
            ctx.set_reduction("tree")
            try:
                tree_out = (torch.empty_like(views2[1][0]), torch.empty_like(views2[1][1]))
                tree_launch = klt.bind(d_ref, d_cur_in, d_st_in, tree_out[0], tree_out[1], None)
                for _ in range(max(3, min(args.warmup, 20))):
                    tree_launch()
                stream.synchronize()
                t_tree = time.perf_counter()
                for _ in range(args.steps):
                    tree_launch()
                stream.synchronize()
                tree_elapsed = time.perf_counter() - t_tree
                tree = {"elapsed": tree_elapsed, "uv": tree_out[0].cpu().numpy(), "st": tree_out[1].cpu().numpy()}
            finally:
                ctx.set_reduction("exact")
        upload = None
        if world == 1 and args.features == 0 and not args.no_upload_leg:
            upload = with_pyramid_upload(args, cfg, ctx, klt, opt, ref_img, cur_img, d_ref, d_cur_in, d_st_in, n, levels, stream, views2[1])
        configs_leg, host_call = None, None
        if world == 1 and args.features == 0 and not args.no_configs_leg:
            from feature_tracker_amd import _native as NL0
            src_hash = NL0.build_info().get("source_hash")
            k_cfg = max(20, min(args.steps, 100))
            configs_leg = {}
            t_cfg = time.perf_counter()
            for name in ("config1", "config3", "config4", "config5_shard"):
                if name == args.workload:
                    continue
                configs_leg[name if name != "config4" else "config4_tracker"] = tracker_config_leg(name, dict(synth.CONFIGS[name]), k_cfg, ctx, dev, stream, src_hash)
            configs_leg["config4_matcher"] = matcher_config_leg(k_cfg, ctx, dev, stream)
            configs_leg["config4_tracker_luminance"] = tracker_config_leg("config4_luminance", dict(synth.CONFIGS["config4"], luminance=True), k_cfg, ctx, dev, stream, src_hash)
            configs_leg["basic_fast_2000_13x13"] = tracker_config_leg("basic_fast", dict(synth.CONFIGS["config2"], half=6, method="fast"), k_cfg, ctx, dev, stream, src_hash)
            # the reference's default method (kFast) of the other two trackers, at sizes their one-wave kernels serve
            configs_leg["affine_fast_5000_13x13"] = tracker_config_leg("affine_fast", dict(synth.CONFIGS["config3"], width=640, height=480, levels=4, method="fast"), k_cfg, ctx, dev, stream, src_hash)
            configs_leg["seconds_spent"] = time.perf_counter() - t_cfg
        if world == 1 and args.features == 0 and not args.no_host_call_leg:
            host_call = host_call_leg(cfg, ref_levels, cur_levels, uv, first_uv, first_st)
        real_images = None
        if world == 1 and args.features == 0 and not args.no_real_images_leg:
            try:
                real_images = real_images_leg(max(20, min(args.steps, 100)), ctx, dev, stream)
            except (ImportError, FileNotFoundError) as exc:  # no PIL / no data files on this machine: say so in the line
                real_images = {"skipped": f"{type(exc).__name__}: {exc}"}
        if use_dist:
            # every rank must now hold every rank's result shard: spot-check the own shard inside the gathered buffer
            per = FD.packed_bytes(n)
            for slot in range(min(2, args.steps)):
                assert torch.equal(gathered2[slot][rank * per:(rank + 1) * per], packed2[slot]), "all-gather did not return this rank's shard"

    rccl_ranks = int(dist.get_world_size()) if use_dist else 1  # ranks of the process group the per-step all-gather ran over
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms_bracketed = float(np.mean([ev0[k].elapsed_time(ev1[k]) for k in ev0]))
    kernel_ms = float(kernel_ms_batch)  # the ONE kernel time both roofline objects use
    # the LAST launch's result as well: calls of >= 4096 features go through the longest-first launch order from the third call on
    # (ftk_klt_track_device), so the first step alone would not show that the ordered launches return the same bits
    last_uv, last_st = views2[0][0].cpu().numpy().copy(), views2[0][1].cpu().numpy().copy()
    if rank == 0:
        total_features = n * world * args.steps
        value = total_features / elapsed
        algo = algorithmic_bytes(iters, levels, half, cfg["method"])
        achieved = algo / (kernel_ms * 1e-3) / 1e9
        from feature_tracker_amd import _native as NL
        build = NL.build_info()
        pmc, pmc_refused = pmc_profile(args.workload, build.get("source_hash")) if args.features == 0 else (None, "--features overrides the workload")
        pmc = pmc or {}
        kernel_name = "klt_basic_inverse_pipelined_kernel" if (cfg["model"], cfg["method"]) == ("basic", "inverse") else f"klt_track_kernel<{cfg['model']}, {cfg['method']}>"
        cpu_uv, cpu_st, cpu_it, _ = oracle_once(cfg, ref_levels, cur_levels, uv)
        out = {
            "metric": "tracked features/sec (2000 pts, 640x480, 4-lvl pyr)", "value": value, "unit": "tracked features/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['model']} KLT {cfg['method']}, {n} features/GPU, {w}x{h}, {levels}-level pyramid, "
                                   f"{2 * half + 1}x{2 * half + 1} patch", "parallelism": (f"features sharded x{world}, pyramids replicated, one RCCL all-gather of packed (uv,status) per step"
                                       + (", steps captured in one HIP graph (gather k overlaps kernel k+1)" if graph is not None else ""))
                       if use_dist else "single GPU",
                       "tracked_fraction": float((status == 1).mean()), "mean_iterations_per_feature": float(iters.mean())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc.get("bytes_per_launch"), "traffic_source": pmc.get("source"), "traffic_refused": pmc_refused,
                         "kernel": kernel_name, "kernel_ms": kernel_ms, "kernel_launches_timed": len(ev0),
                         "kernel_ms_method": f"one HIP event pair on the launch stream around {batch_n} back-to-back launches in a separate pass after the "
                                             "timed region, divided by the count (includes the inter-launch gap: an upper bound of the kernel time)",
                         "kernel_ms_single_launch_event_pair": kernel_ms_bracketed,
                         "kernel_ms_rocprofv3_trace_this_build": (pmc.get("trace_average_ns") or 0) * 1e-6 or None,
                         "algorithmic_bytes_per_launch": algo},
            "build": build,
            "parity": dict(parity_report(first_uv, first_st, cpu_uv, cpu_st), iteration_counts_equal=bool(np.array_equal(iters, cpu_it)),
                           last_launch_bit_identical=bool(np.array_equal(last_uv.view(np.uint32), cpu_uv.view(np.uint32)) and np.array_equal(last_st, cpu_st))),
        }
        if pmc.get("valu_insts_per_launch"):
            valu = pmc["valu_insts_per_launch"] / (kernel_ms * 1e-3)
            out["roofline_valu"] = {"bound": "valu_issue", "achieved": valu, "peak": VALU_PEAK_DATASHEET, "unit": "wave64 VALU instructions/s",
                                    "frac": valu / VALU_PEAK_DATASHEET, "valu_insts_per_launch": pmc["valu_insts_per_launch"],
                                    "peak_is": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md cycle constants; the figure of rounds 1-2)",
                                    "frac_of_measured_peak": valu / VALU_PEAK_MEASURED,
                                    "measured_peak_is": f"1024 SIMDs x 2.4 GHz / {VALU_CYCLES_PER_WAVE_INST_MEASURED} cycles, the rate a saturated SIMD sustains "
                                                        "(scripts/microbench/int_valu_rate.hip, profiles/r2_microbench_int_valu.txt)",
                                    "lds_bank_conflict_frac": pmc.get("lds_bank_conflict_frac"), "source": pmc.get("source")}
        else:
            out["roofline_valu"] = None  # no counters of THIS build: see roofline.traffic_refused
        if tree is not None:
            rep = parity_report(tree["uv"], tree["st"], cpu_uv, cpu_st)
            out["throughput_mode"] = {
                "value": n * args.steps / tree["elapsed"], "unit": "tracked features/s", "ms_per_step": tree["elapsed"] / args.steps * 1e3,
                "speedup_over_exact": (elapsed / args.steps) / (tree["elapsed"] / args.steps),
                "max_px": rep["max_px"], "p99_px": rep["p99_px"], "frac_gt_1e-3": rep["frac_gt_1e-3"], "status_mismatches": rep["status_mismatches"],
                "bit_identical": rep["bit_identical"],
                "what": "ftk_set_reduction_mode(FTK_REDUCTION_TREE): the same per-pixel products summed by per-lane partials + a butterfly instead of in the "
                        "reference's pixel order; px-error against the CPU oracle on the timed inputs; reported, not the contract, never the default",
            }
        if upload is not None:
            up_uv, up_st = upload.pop("result_uv").cpu().numpy(), upload.pop("result_st").cpu().numpy()
            upload["bit_identical_to_resident_path"] = bool(np.array_equal(up_uv.view(np.uint32), first_uv.view(np.uint32)) and np.array_equal(up_st, first_st))
            out["with_pyramid_upload"] = upload
        if host_call is not None:
            out["host_call"] = host_call
        if real_images is not None:
            out["real_images"] = real_images
        if configs_leg is not None:
            # the headline workload in the same shape, so that the object covers all five BASELINE configurations
            configs_leg[args.workload] = {
                "workload": out["config"]["workload"], "ms_per_step": out["ms_per_step"], "features_per_s": value, "steps": args.steps,
                "regime": "n/a (below 4096 features: list order)" if n < 4096 else "warm", "cold_ms": None, "roofline_frac": out["roofline"]["frac"],
                "algorithmic_bytes_per_launch": algo, "valu_frac": (out.get("roofline_valu") or {}).get("frac"),
                "bit_identical": bool(out["parity"]["bit_identical"] and out["parity"]["last_launch_bit_identical"] and out["parity"]["iteration_counts_equal"]),
                "mean_iterations_per_feature": float(iters.mean())}
            out["configs"] = configs_leg
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, ref_levels, cur_levels, uv)
    else:
        out = None
    if use_dist and not args.no_config5_leg:
        # BASELINE.json configs[4] (200 000 features sharded over the ranks): measured by every N > 1 run, after the weak-scaling region
        leg = run_config5_leg(args, world, rank, local_rank, dev, ctx, stream, out)
        if out is not None:
            out["config5_sharded"] = leg
    if out is not None:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
