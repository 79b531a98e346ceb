#!/usr/bin/env python3
"""Back-to-back launch time of tracker variants on device-resident inputs (bench.py's timed-region method: K launches, one
synchronisation on each side, wall clock), checked against the oracle:
    python scripts/time_variant.py basic:fast:2000:6 lssd:fast:10000:6:lum affine:fast:2000:6 [--steps 50] [--size 640x480] [--levels 4]
Each spec is model:method:n:half[:lum].  --real: the reference's example pair (tests/data/optical_flow, 752x480) with Harris corners topped up
to n (bench.py real_image_features) instead of the synthetic scene."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("specs", nargs="+")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--size", default="640x480")
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--fixed-iterations", action="store_true", help="never converge, never stop on large steps: kMaxIteration iterations on every level (what ONE iteration costs)")
    ap.add_argument("--max-iteration", type=int, default=15)
    ap.add_argument("--real", action="store_true", help="the reference's example image pair + Harris corners instead of the synthetic scene")
    args = ap.parse_args()
    import torch
    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import synth
    from tests import oracle_lib
    w, h = (int(x) for x in args.size.split("x"))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    scenes = {}
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        for spec in args.specs:
            f = spec.split(":")
            model, method, n, half = f[0], f[1], int(f[2]), int(f[3])
            lum = len(f) > 4 and f[4] == "lum"
            key = "real" if args.real else model == "basic"
            if key not in scenes:
                if args.real:
                    from PIL import Image
                    import bench
                    ref = np.ascontiguousarray(np.array(Image.open(bench.REAL_PAIR[0]).convert("L"), dtype=np.uint8))
                    cur = np.ascontiguousarray(np.array(Image.open(bench.REAL_PAIR[1]).convert("L"), dtype=np.uint8))
                else:
                    ref, cur = synth.make_image_pair(w, h, (3.3, -2.1)) if key else synth.make_image_pair(w, h, (3.3, -2.1), rotation_deg=1.5, scale=1.02)
                rl, cl = synth.build_pyramid(ref, args.levels), synth.build_pyramid(cur, args.levels)
                scenes[key] = (rl, cl, D.upload_pyramid(rl, ctx, dev), D.upload_pyramid(cl, ctx, dev))
            rl, cl, rp, cp = scenes[key]
            if args.real:
                import bench
                uv, _ = bench.real_image_features(rl[0], n, half, ctx)
            else:
                uv = synth.make_features(n, w, h, half=half)
            opt = F.OpticalFlowOptions()
            opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, half, half, n
            opt.kMaxIteration = args.max_iteration
            if args.fixed_iterations:
                opt.kMaxConvergeStep, opt.kMaxToleranceLargeStep = 0.0, 1 << 30
            klt = D.DeviceKlt(model, opt, rp, cp, ctx, consider_luminance=lum)
            d_ref = torch.from_numpy(uv).to(dev)
            d_in, d_st = d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev)
            outs = [(torch.empty_like(d_ref), torch.empty_like(d_st)) for _ in range(2)]
            d_it = torch.zeros(n, dtype=torch.int32, device=dev)
            klt.track(d_ref, d_in, d_st, outs[0][0], outs[0][1], d_it)
            stream.synchronize()
            launches = [klt.bind(d_ref, d_in, d_st, o[0], o[1], None) for o in outs]
            for k in range(6):
                launches[k & 1]()
            stream.synchronize()
            best = []
            for _ in range(3):
                t0 = time.perf_counter()
                for k in range(args.steps):
                    launches[k & 1]()
                stream.synchronize()
                best.append((time.perf_counter() - t0) / args.steps * 1e6)
            out = {"spec": spec, "us_per_step": round(min(best), 2), "us_runs": [round(x, 2) for x in best], "mean_iters": float(d_it.float().mean().item()), "max_iters": int(d_it.max().item())}
            if not args.no_oracle and not args.fixed_iterations:
                t0 = time.perf_counter()
                ok, cuv, cst, cit = oracle_lib.klt_track_pyramid(model, rl, cl, uv, method=method, half=half, max_points=n, consider_luminance=lum)
                out["cpu_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
                g_uv, g_st = outs[(args.steps - 1) & 1][0].cpu().numpy(), outs[(args.steps - 1) & 1][1].cpu().numpy()
                out["bit_identical"] = bool(np.array_equal(g_uv.view(np.uint32), cuv.view(np.uint32)) and np.array_equal(g_st, cst) and np.array_equal(d_it.cpu().numpy().astype(np.uint32), cit))
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
