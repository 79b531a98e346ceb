#!/usr/bin/env python3
"""The sweep behind feature_tracker_amd/csrc/klt_wave_policy.inc (run on an MI355X; scripts/make_wave_policy.py turns its output into the table):
every tracker variant x patch size x feature count x waves per feature, on the synthetic scene (every feature done after a few
iterations) and on the reference's example pair (a few features never converge), back-to-back launches on device-resident buffers.
    python scripts/wave_policy_sweep.py > profiles/r5_wave_policy_sweep.jsonl
One process: the wave count is forced through FTK_KLT_WAVES, which the library reads once per context — refreshed per cell."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

VARIANTS = ["basic:inverse", "basic:direct", "basic:fast", "affine:inverse", "affine:direct", "affine:fast", "lssd:inverse", "lssd:direct", "lssd:fast", "lssd:fast:lum"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--halves", default="4,5,6,7,8,10")
    ap.add_argument("--features", default="300,800,1200,1700,2400,3500,5000,7000,10000,16000")
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--scenes", default="synthetic,real")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--budget-seconds", type=float, default=900.0)
    args = ap.parse_args()
    import torch
    from PIL import Image
    import bench
    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import synth
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    t_start = time.perf_counter()
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        scenes = {}
        for name in args.scenes.split(","):
            for kind in ("plain", "warped"):
                if name == "real":
                    ref = np.ascontiguousarray(np.array(Image.open(bench.REAL_PAIR[0]).convert("L"), dtype=np.uint8))
                    cur = np.ascontiguousarray(np.array(Image.open(bench.REAL_PAIR[1]).convert("L"), dtype=np.uint8))
                elif kind == "plain":
                    ref, cur = synth.make_image_pair(640, 480, (3.3, -2.1))
                else:
                    ref, cur = synth.make_image_pair(640, 480, (3.3, -2.1), rotation_deg=1.5, scale=1.02)
                rl, cl = synth.build_pyramid(ref, 4), synth.build_pyramid(cur, 4)
                scenes[(name, kind)] = (rl, D.upload_pyramid(rl, ctx, dev), D.upload_pyramid(cl, ctx, dev))
        done = 0
        for scene in args.scenes.split(","):
            for half in (int(x) for x in args.halves.split(",")):
                pixels = (2 * half + 1) ** 2
                for n in (int(x) for x in args.features.split(",")):
                    uv_cache = {}
                    for variant in args.variants.split(","):
                        f = variant.split(":")
                        model, method, lum = f[0], f[1], len(f) > 2
                        rl, rp, cp = scenes[(scene, "plain" if model == "basic" else "warped")]
                        if scene not in uv_cache:
                            uv_cache[scene] = bench.real_image_features(rl[0], n, half, ctx)[0] if scene == "real" else synth.make_features(n, 640, 480, half=half)
                        uv = uv_cache[scene]
                        opt = F.OpticalFlowOptions()
                        opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, half, half, n
                        d_ref = torch.from_numpy(uv).to(dev)
                        d_in, d_st = d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev)
                        outs = [(torch.empty_like(d_ref), torch.empty_like(d_st)) for _ in range(2)]
                        d_it = torch.zeros(n, dtype=torch.int32, device=dev)
                        for waves in range(1, min(4, (pixels + 63) // 64 + 1) + 1):
                            os.environ["FTK_KLT_WAVES"] = str(waves)
                            ctx.refresh_env()
                            klt = D.DeviceKlt(model, opt, rp, cp, ctx, consider_luminance=lum)
                            klt.track(d_ref, d_in, d_st, outs[0][0], outs[0][1], d_it)
                            launches = [klt.bind(d_ref, d_in, d_st, o[0], o[1], None) for o in outs]
                            for k in range(8):  # the launch order and the tail class settle over a few calls
                                launches[k & 1]()
                                if k % 3 == 2:
                                    stream.synchronize()
                            stream.synchronize()
                            best = None
                            for _ in range(2):
                                t0 = time.perf_counter()
                                for k in range(args.steps):
                                    launches[k & 1]()
                                stream.synchronize()
                                us = (time.perf_counter() - t0) / args.steps * 1e6
                                best = us if best is None else min(best, us)
                            print(json.dumps({"variant": variant, "scene": scene, "half": half, "pixels": pixels, "n": n, "waves": waves, "us": round(best, 2),
                                              "max_iters": int(d_it.max().item()), "mean_iters": round(float(d_it.float().mean().item()), 2)}), flush=True)
                            done += 1
                    if time.perf_counter() - t_start > args.budget_seconds:
                        print(json.dumps({"stopped": "budget", "cells": done}), flush=True)
                        return
        os.environ.pop("FTK_KLT_WAVES", None)
    print(json.dumps({"finished": True, "cells": done, "seconds": round(time.perf_counter() - t_start, 1)}), flush=True)


if __name__ == "__main__":
    main()
