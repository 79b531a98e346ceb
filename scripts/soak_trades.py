#!/usr/bin/env python3
"""Experiment / stress (GPU box): position-keyed trades of launch slots under a list that is reshuffled before every call.  Every
call must return, feature by feature, what the first (list-order) call returned for the same feature — whichever slot ran it.
    python scripts/soak_trades.py [seconds=60] [seed=1]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import feature_tracker_amd as F  # noqa: E402
from feature_tracker_amd import device as D, synth  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
t_end = time.time() + budget
rounds = calls = mismatches = 0
with torch.cuda.stream(stream):
    ctx = D.context_on_stream(stream, 0)
    while time.time() < t_end:
        w, h = int(rs.choice([320, 640, 1280])), int(rs.choice([240, 480, 720]))
        levels = int(rs.randint(2, 5))
        model = str(rs.choice(["affine", "lssd", "basic"]))
        method = str(rs.choice(["inverse", "direct"])) if model != "basic" else "inverse"
        half = 6 if model != "basic" else int(rs.choice([8, 10]))  # multi-wave launches: 13 x 13 non-fast affine / LSSD, large Basic patches
        n = int(rs.randint(4096, 9000))
        ref_img, cur_img = synth.make_image_pair(w, h, (float(rs.uniform(-9, 9)), float(rs.uniform(-9, 9))), rotation_deg=float(rs.uniform(-3, 3)), scale=float(rs.uniform(0.97, 1.03)))
        uv = synth.make_features(n, w, h, seed=int(rs.randint(1 << 30)), half=half)
        opt = F.OpticalFlowOptions()
        opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, half, half, n
        klt = D.DeviceKlt(model, opt, D.upload_pyramid(synth.build_pyramid(ref_img, levels), ctx, dev), D.upload_pyramid(synth.build_pyramid(cur_img, levels), ctx, dev), ctx)
        d_st = torch.zeros(n, dtype=torch.uint8, device=dev)

        def run(points):
            d_ref = torch.from_numpy(np.ascontiguousarray(points)).to(dev)
            d_out, d_so = torch.empty_like(d_ref), torch.empty(n, dtype=torch.uint8, device=dev)
            klt.track(d_ref, d_ref.clone(), d_st, d_out, d_so)
            stream.synchronize()
            return d_out.cpu().numpy(), d_so.cpu().numpy()

        base_uv, base_st = run(uv)
        for _ in range(int(rs.randint(3, 8))):
            perm = rs.permutation(n)
            g_uv, g_st = run(uv[perm])
            calls += 1
            bad = (not np.array_equal(g_st, base_st[perm])) or (not np.array_equal(g_uv.view(np.uint32), base_uv[perm].view(np.uint32)))
            if bad:
                mismatches += 1
                print(f"MISMATCH: {model}/{method} n {n} {w}x{h} levels {levels}")
        rounds += 1
print(f"soak trades: {rounds} scenes, {calls} reshuffled calls in {budget:.0f} s, {mismatches} mismatches")
sys.exit(1 if mismatches else 0)
