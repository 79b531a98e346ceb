#!/usr/bin/env python3
"""Per-phase cycle totals of one tracker configuration with the stamps build (scripts/build_stamps.sh):
   FTK_LIB_PATH=feature_tracker_amd/csrc/diag/libftk_hip_stamps.so python scripts/stamps_fast.py basic fast 2000 6 [levels]
The library prints "[ftk stamps] memtime ticks/feature ..." (100 MHz ticks: x 24 = shader cycles at 2.4 GHz) for the first launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import feature_tracker_amd as F
from feature_tracker_amd import synth
model, method, n, half = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
levels = int(sys.argv[5]) if len(sys.argv) > 5 else 4
if model == "basic":
    ref, cur = synth.make_image_pair(640, 480, (3.3, -2.1))
else:
    ref, cur = synth.make_image_pair(640, 480, (3.3, -2.1), rotation_deg=1.5, scale=1.02)
rl, cl = synth.build_pyramid(ref, levels), synth.build_pyramid(cur, levels)
uv = synth.make_features(n, 640, 480, half=half)
klt = {"basic": F.OpticalFlowBasicKlt, "affine": F.OpticalFlowAffineKlt, "lssd": F.OpticalFlowLssdKlt}[model]()
o = klt.options()
o.kMethod, o.kPatchRowHalfSize, o.kPatchColHalfSize, o.kMaxTrackPointsNumber = method, half, half, n
if len(sys.argv) > 6 and model == "lssd":
    klt.consider_patch_luminance = bool(int(sys.argv[6]))
rp, cp = F.ImagePyramid.from_host_levels(rl), F.ImagePyramid.from_host_levels(cl)
for _ in range(3):
    ok, c, s = klt.TrackFeatures(rp, cp, uv)
print("tracked", float((s == 1).mean()), "mean iters", float(np.mean(klt.last_iterations)))
