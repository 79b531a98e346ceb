#!/usr/bin/env python3
"""Experiment (GPU box): marginal kernel time of one pyramid level and of one Gauss-Newton iteration
on the config-2 workload, by timing the same launch with fewer levels / a lower kMaxIteration.
    python scripts/marginal_costs.py [n_features [levels max_iteration]]     (one combination only, for PMC runs)
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import feature_tracker_amd as F
from feature_tracker_amd import device as D, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
w, h, half = 640, 480, 10
model, method = "basic", "inverse"
if os.environ.get("FTK_MARGINAL_CONFIG"):  # e.g. config3 / config4: that configuration's model, method, image and patch
    cfg = synth.CONFIGS[os.environ["FTK_MARGINAL_CONFIG"]]
    w, h, half, model, method = cfg["width"], cfg["height"], cfg["half"], cfg["model"], cfg["method"]
if model == "basic":
    ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1))
else:
    ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1), rotation_deg=1.5, scale=1.02)
uv = synth.make_features(n, w, h, half=half)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = D.context_on_stream(stream, 0)
    only = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None
    for levels in (5, 4, 3, 2, 1):
        rl, cl = synth.build_pyramid(ref_img, levels), synth.build_pyramid(cur_img, levels)
        rp, cp = D.upload_pyramid(rl, ctx, dev), D.upload_pyramid(cl, ctx, dev)
        for max_it in (15, 1) if only is None else (only[1],):
            if only is not None and levels != only[0]:
                continue
            opt = F.OpticalFlowOptions()
            opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber, opt.kMaxIteration = method, half, half, n, max_it
            klt = D.DeviceKlt(model, opt, rp, cp, ctx)
            d_ref = torch.from_numpy(uv).to(dev); d_in = d_ref.clone()
            d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
            d_out = torch.empty_like(d_ref); d_sto = torch.empty_like(d_st); d_it = torch.zeros(n, dtype=torch.int32, device=dev)
            klt.track(d_ref, d_in, d_st, d_out, d_sto, d_it); stream.synchronize()
            ts = []
            for _ in range(60 if only is None else 8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream); klt.track(d_ref, d_in, d_st, d_out, d_sto, None); e1.record(stream); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            print(f"levels {levels} max_iteration {max_it:2d}: kernel {np.median(ts):6.1f} us, mean iterations {d_it.cpu().numpy().mean():.2f}")
