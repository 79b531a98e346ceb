#!/usr/bin/env python3
"""Experiment helper (GPU box): one launch of a batch of DirectMethod pose problems (300 points, 13 x 13, 4 levels), spread over the chip
(default) against one workgroup per problem (FTK_DIRECT_SPREAD=0).   python scripts/direct_batch_time.py 1 2 6 8 12 16 24 30 64"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import torch
    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import synth
    FX, FY, CX, CY = 400.0, 410.0, 321.5, 238.25
    ref, cur = synth.make_image_pair(640, 480, (3.3, -2.1))
    rl, cl = synth.build_pyramid(ref, 4), synth.build_pyramid(cur, 4)
    uv = synth.make_features(300, 640, 480, half=6)
    z = (5.0 * np.random.RandomState(1).uniform(0.8, 1.25, len(uv))).astype(np.float32)
    pts = np.stack([(uv[:, 0] - CX) / FX * z, (uv[:, 1] - CY) / FY * z, z], axis=1).astype(np.float32)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        rp, cp = D.upload_pyramid(rl, ctx, dev), D.upload_pyramid(cl, ctx, dev)
        spread_env = os.environ.get("FTK_DIRECT_SPREAD")  # (=n: that many producer workgroups per problem, if they fit)
        for count in (int(a) for a in sys.argv[1:]):
            row = {}
            for mode in ("spread", "one-workgroup"):
                if mode == "spread":
                    os.environ.pop("FTK_DIRECT_SPREAD", None)
                    if spread_env is not None:
                        os.environ["FTK_DIRECT_SPREAD"] = spread_env
                else:
                    os.environ["FTK_DIRECT_SPREAD"] = "0"
                ctx.refresh_env()
                times = []
                for rep in range(4):
                    problems = [dict(ref=rp, cur=cp, K=[FX, FY, CX, CY], p_c_in_ref=torch.from_numpy(pts).to(dev), ref_uv=torch.from_numpy(uv).to(dev),
                                     cur_uv=torch.from_numpy(uv.copy()).to(dev), pose=torch.tensor([1, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=dev),
                                     status=torch.zeros(300, dtype=torch.uint8, device=dev), status_valid=False,
                                     iterations=torch.zeros(1, dtype=torch.int32, device=dev)) for _ in range(count)]
                    batch = D.DeviceDirectBatch(F.DirectMethodOptions(), problems, ctx)
                    stream.synchronize()
                    t0 = time.perf_counter()
                    batch.track()
                    stream.synchronize()
                    times.append((time.perf_counter() - t0) * 1e3)
                row[mode] = min(times[1:])
            print(f"{count:3d} problems: spread {row['spread']:.2f} ms, one workgroup each {row['one-workgroup']:.2f} ms")


if __name__ == "__main__":
    main()
