#!/usr/bin/env python3
"""Experiment (GPU box): how long does the launch-order block (klt_common.h klt_order_block, one extra workgroup of a tracker launch)
take by itself?  Every feature is passed through (incoming status = failed), so the launch is as long as its sort block.
    python scripts/sort_block_time.py [n=25000] [width height levels half]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import feature_tracker_amd as F  # noqa: E402
from feature_tracker_amd import device as D, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
w, h, levels, half = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1920, 1080, 4, 6)))
ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1))
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
uv = synth.make_features(n, w, h, seed=12345, half=half)
with torch.cuda.stream(stream):
    ctx = D.context_on_stream(stream, 0)
    ref_pyr = D.upload_pyramid(synth.build_pyramid(ref_img, levels), ctx, dev)
    cur_pyr = D.upload_pyramid(synth.build_pyramid(cur_img, levels), ctx, dev)
    opt = F.OpticalFlowOptions()
    opt.kMethod = "inverse"
    opt.kPatchRowHalfSize = opt.kPatchColHalfSize = half
    opt.kMaxTrackPointsNumber = n
    klt = D.DeviceKlt("basic", opt, ref_pyr, cur_pyr, ctx)
    d_ref = torch.from_numpy(uv).to(dev)
    d_out, d_so = torch.empty_like(d_ref), torch.empty(n, dtype=torch.uint8, device=dev)
    for name, status in (("all features passed through", 3), ("all features tracked", 0)):
        d_st = torch.full((n,), status, dtype=torch.uint8, device=dev)
        for _ in range(6):
            klt.track(d_ref, d_ref.clone(), d_st, d_out, d_so, None)
        stream.synchronize()
        times = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            klt.track(d_ref, d_ref, d_st, d_out, d_so, None)
            e1.record(stream)
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        print(f"n {n}: {name}: {np.median(times):.1f} us per isolated call (FTK_KLT_SCHED={os.environ.get('FTK_KLT_SCHED', '1')})")
