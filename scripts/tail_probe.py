#!/usr/bin/env python3
"""Experiment (GPU box): is a tracker launch bound by its slowest feature, and where does that feature spend its time?
    python scripts/tail_probe.py affine:fast:2000:6 [--size 640x480] [--levels 4]
Times the whole launch, the single longest feature alone, the 8 longest, all but the 8 longest and one median feature alone.
With the stamps build (scripts/build_stamps.sh; FTK_LIB_PATH=.../diag/libftk_hip_stamps.so FTK_STAMPS_DUMP=/tmp/st.bin) also prints
the per-phase s_memtime totals (us) of the longest and of the median feature when each runs alone."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import feature_tracker_amd as F
from feature_tracker_amd import device as D, synth

ap = argparse.ArgumentParser()
ap.add_argument("spec")
ap.add_argument("--size", default="640x480")
ap.add_argument("--levels", type=int, default=4)
args = ap.parse_args()
f = args.spec.split(":")
model, method, n, half = f[0], f[1], int(f[2]), int(f[3])
lum = len(f) > 4 and f[4] == "lum"
w, h = (int(x) for x in args.size.split("x"))
ref, cur = synth.make_image_pair(w, h, (3.3, -2.1)) if model == "basic" else synth.make_image_pair(w, h, (3.3, -2.1), rotation_deg=1.5, scale=1.02)
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
uv = synth.make_features(n, w, h, half=half)
dump = os.environ.get("FTK_STAMPS_DUMP")
with torch.cuda.stream(stream):
    ctx = D.context_on_stream(stream, 0)
    rp, cp = D.upload_pyramid(synth.build_pyramid(ref, args.levels), ctx, dev), D.upload_pyramid(synth.build_pyramid(cur, args.levels), ctx, dev)

    def run(points, reps=30):
        m = len(points)
        opt = F.OpticalFlowOptions()
        opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, half, half, m
        klt = D.DeviceKlt(model, opt, rp, cp, ctx, consider_luminance=lum)
        d_ref = torch.from_numpy(np.ascontiguousarray(points)).to(dev)
        d_in, d_st = d_ref.clone(), torch.zeros(m, dtype=torch.uint8, device=dev)
        d_out, d_so, d_it = torch.empty_like(d_ref), torch.empty_like(d_st), torch.zeros(m, dtype=torch.int32, device=dev)
        klt.track(d_ref, d_in, d_st, d_out, d_so, d_it)
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            klt.track(d_ref, d_in, d_st, d_out, d_so, None)
        e0.record(stream)
        for _ in range(reps):
            klt.track(d_ref, d_in, d_st, d_out, d_so, None)
        e1.record(stream)
        stream.synchronize()
        st = None
        if dump and os.path.exists(dump):
            st = np.fromfile(dump, dtype=np.uint64).reshape(-1, 8).astype(np.float64) * 0.01
        return e0.elapsed_time(e1) / reps * 1e3, d_it.cpu().numpy(), st

    t_all, iters, _ = run(uv)
    order = np.argsort(-iters.astype(np.int64), kind="stable")
    print(f"{args.spec}: {t_all:.1f} us per launch; iterations mean {iters.mean():.2f} max {iters.max()}; top 8: {iters[order[:8]].tolist()}")
    names = "stage extract setup sweep next chain solve total".split()
    for label, pts in (("longest alone", uv[order[:1]]), ("8 longest", uv[order[:8]]), ("all but the 8 longest", uv[np.sort(order[8:])]),
                       ("a median feature alone", uv[order[n // 2: n // 2 + 1]])):
        t, it, st = run(pts)
        line = f"  {label:24s} {t:7.1f} us (iterations {it.min()}..{it.max()})"
        if st is not None and len(pts) == 1:
            line += "  stamps[us]: " + " ".join(f"{k} {v:.1f}" for k, v in zip(names, st[0]))
        print(line)
