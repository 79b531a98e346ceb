#!/usr/bin/env python3
"""Experiment (GPU box): how long does a launch of only the K LONGEST features of a workload take?  With K <= the number of features
the chip holds at once this is the latency of the workload's slowest feature — the floor of the full launch, whatever its order.
    python scripts/longest_features.py config3 [K=64] [only]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import feature_tracker_amd as F  # noqa: E402
from feature_tracker_amd import device as D, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = dict(synth.CONFIGS[name])
n, w, h, levels, half = cfg["n"], cfg["width"], cfg["height"], cfg["levels"], cfg["half"]
if cfg["model"] == "basic":
    ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1))
else:
    ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1), rotation_deg=1.5, scale=1.02)
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
uv = synth.make_features(n, w, h, seed=12345, half=half)
with torch.cuda.stream(stream):
    ctx = D.context_on_stream(stream, 0)
    ref_pyr = D.upload_pyramid(synth.build_pyramid(ref_img, levels), ctx, dev)
    cur_pyr = D.upload_pyramid(synth.build_pyramid(cur_img, levels), ctx, dev)

    def run(points, reps):
        m = len(points)
        opt = F.OpticalFlowOptions()
        opt.kMethod = cfg["method"]
        opt.kPatchRowHalfSize = opt.kPatchColHalfSize = half
        opt.kMaxTrackPointsNumber = m
        klt = D.DeviceKlt(cfg["model"], opt, ref_pyr, cur_pyr, ctx)
        d_ref = torch.from_numpy(np.ascontiguousarray(points)).to(dev)
        d_in, d_st = d_ref.clone(), torch.zeros(m, dtype=torch.uint8, device=dev)
        d_out, d_so, d_it = torch.empty_like(d_ref), torch.empty_like(d_st), torch.zeros(m, dtype=torch.int32, device=dev)
        klt.track(d_ref, d_in, d_st, d_out, d_so, d_it)
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            klt.track(d_ref, d_in, d_st, d_out, d_so, None)
        e0.record(stream)
        for _ in range(reps):
            klt.track(d_ref, d_in, d_st, d_out, d_so, None)
        e1.record(stream)
        stream.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3, d_it.cpu().numpy()

    t_all, iters = run(uv, 30)
    order = np.argsort(-iters.astype(np.int64), kind="stable")
    print(f"{name}: {n} features {t_all:.1f} us per launch; iterations mean {iters.mean():.2f} max {iters.max()}")
    if len(sys.argv) > 3 and sys.argv[3] == "only":  # e.g. with the -DFTK_STAMPS build and FTK_STAMPS_DUMP: the last launch is the K longest
        t, it = run(uv[order[:K]], 3)
        print(f"  the {K:4d} longest features alone: {t:7.1f} us per launch (iterations {it.min()}..{it.max()})")
        sys.exit(0)
    for k in sorted({1, 8, K, 256}):
        t, it = run(uv[order[:k]], 30)
        print(f"  the {k:4d} longest features alone: {t:7.1f} us per launch (iterations {it.min()}..{it.max()})")
    for k in (1, 8, 64):  # the launch WITHOUT its k longest features: what the tail costs
        t, it = run(uv[np.sort(order[k:])], 30)
        print(f"  all but the {k:3d} longest features: {t:7.1f} us per launch (iterations {it.min()}..{it.max()})")
    t, it = run(uv[order[-256:]], 30)
    print(f"  the  256 shortest features alone: {t:7.1f} us per launch (iterations {it.min()}..{it.max()})")
