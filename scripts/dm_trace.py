#!/usr/bin/env python3
"""Experiment helper (GPU box; a build with docs/experiments/direct_spread_consumer_variants.diff.txt applied and -DFTK_DM_TRACE, through FTK_LIB_PATH):
when do the rounds of the spread direct-method kernel's consumer arrive and how long does each take to chain?  Prints, for the third
iteration of a single 300-point problem, the gaps between consecutive stamps (s_memrealtime, 100 MHz): wait-for-round / chain-round."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import torch
    import feature_tracker_amd as F
    from feature_tracker_amd import _native as N
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
        poses = []
        spread_env = os.environ.get("FTK_DIRECT_SPREAD")
        for rep in range(3):
            problems = [dict(ref=rp, cur=cp, K=[FX, FY, CX, CY], p_c_in_ref=torch.from_numpy(pts).to(dev), ref_uv=torch.from_numpy(uv).to(dev),
                             cur_uv=torch.from_numpy(uv.copy()).to(dev), pose=torch.tensor([1, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=dev),
                             status=torch.zeros(300, dtype=torch.uint8, device=dev), status_valid=False, iterations=torch.zeros(1, dtype=torch.int32, device=dev))]
            D.DeviceDirectBatch(F.DirectMethodOptions(), problems, ctx).track()
            stream.synchronize()
            poses.append(problems[0]["pose"].cpu().numpy().tobytes())
            if rep == 1:
                os.environ["FTK_DIRECT_SPREAD"] = "0"  # the third run: the one-workgroup kernel, whose pose the spread kernel must reproduce bit for bit
                ctx.refresh_env()
    print("pose of the spread kernel == pose of the one-workgroup kernel, bitwise:", poses[0] == poses[2] and poses[1] == poses[2])
    buf = np.zeros(8192, dtype=np.uint64)
    lib = N.lib()
    lib.ftk_debug_dm_trace.argtypes = [C.c_void_p, C.c_int]
    assert lib.ftk_debug_dm_trace(buf.ctypes.data_as(C.c_void_p), 8192) == 0
    n = int(buf[8191])
    t = buf[:n].astype(np.int64)
    t = (t - t[0]) * 10  # ns
    arrive, chained = t[1::2], t[2::2]
    m = min(len(arrive), len(chained))
    wait = arrive[:m] - np.concatenate([[0], chained[:m - 1]])
    chain = chained[:m] - arrive[:m]
    print(f"{m} rounds, iteration {t[-1] / 1e3:.1f} us; waiting for a round {wait.sum() / 1e3:.1f} us (first round {wait[0] / 1e3:.2f}), chaining {chain.sum() / 1e3:.1f} us")
    print("per round, ns: wait median %.0f  p90 %.0f  max %.0f | chain median %.0f  p90 %.0f  max %.0f" % (np.median(wait[1:]), np.percentile(wait[1:], 90), wait[1:].max(),
          np.median(chain), np.percentile(chain, 90), chain.max()))
    lt = buf[4096:4096 + 4 * m].astype(np.int64).reshape(m, 4)
    if lt.any():
        # a loader wave's own stamps (wave 7): round start | flag seen | loads back | LDS written
        lt = (lt - int(buf[0])) * 10
        a, b, c, d = lt[8:, 0], lt[8:, 1], lt[8:, 2], lt[8:, 3]
        print("loader wave 7, medians over rounds 8.., ns: poll %.0f | loads %.0f | LDS stores %.0f | then waits at the barrier for %.0f (its round %.0f)" % (
            np.median(b - a), np.median(c - b), np.median(d - c), np.median(a[1:] - d[:-1]), np.median(a[1:] - a[:-1])))
    if buf[8190]:
        print("shader clock over the iteration's stream: %.2f GHz (s_memtime ticks per s_memrealtime tick x 100 MHz)" % ((int(buf[8189]) - int(buf[8190])) / (t[-1] / 10) * 0.1))
    print("first 12 rounds (wait, chain) ns:", [(int(a), int(b)) for a, b in zip(wait[:12], chain[:12])])


if __name__ == "__main__":
    main()
