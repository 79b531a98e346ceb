#!/usr/bin/env python3
"""Experiment helper: one tracker launch per pyramid depth so that rocprofv3 --pmc SQ_INSTS_VALU can
separate the per-level and the per-iteration instruction cost (total = L*S + I*T + C)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feature_tracker_amd as F
from feature_tracker_amd import device as D, synth

method = sys.argv[1] if len(sys.argv) > 1 else "inverse"
ref_img, cur_img = synth.make_image_pair(640, 480, (3.3, -2.1))
uv = synth.make_features(2000, 640, 480, half=10)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
out = []
with torch.cuda.stream(stream):
    ctx = D.context_on_stream(stream, 0)
    for levels in (1, 2, 3, 4):
        rl, cl = synth.build_pyramid(ref_img, levels), synth.build_pyramid(cur_img, levels)
        rp, cp = D.upload_pyramid(rl, ctx, dev), D.upload_pyramid(cl, ctx, dev)
        opt = F.OpticalFlowOptions()
        opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, 10, 10, 2000
        klt = D.DeviceKlt("basic", opt, rp, cp, ctx)
        d_ref = torch.from_numpy(uv).to(dev)
        d_st = torch.zeros(2000, dtype=torch.uint8, device=dev)
        d_out = torch.empty_like(d_ref); d_sto = torch.empty_like(d_st); d_it = torch.zeros(2000, dtype=torch.int32, device=dev)
        klt.track(d_ref, d_ref.clone(), d_st, d_out, d_sto, d_it)
        stream.synchronize()
        out.append({"levels": levels, "iters_total": int(d_it.sum().item())})
print(json.dumps(out))
