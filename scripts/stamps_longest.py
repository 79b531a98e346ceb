#!/usr/bin/env python3
"""Experiment helper (GPU box, a -DFTK_STAMPS build through FTK_LIB_PATH): where do the shader ticks of the LONGEST features of a
real-image call go?  Prints, for the few features with the most iterations, the per-slot s_memtime ticks (klt_*_kernels.hip FTK_STAMP_END
slots) and what is left outside every slot, per iteration.
    FTK_LIB_PATH=feature_tracker_amd/csrc/diag/libftk_hip_stamps.so python scripts/stamps_longest.py basic:inverse:300:6"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import torch
    from PIL import Image
    import bench
    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    from feature_tracker_amd import synth
    dump = os.path.join(tempfile.gettempdir(), "stamps_longest.bin")
    os.environ["FTK_STAMPS_DUMP"] = dump
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ref = np.ascontiguousarray(np.array(Image.open(bench.REAL_PAIR[0]).convert("L"), dtype=np.uint8))
    cur = np.ascontiguousarray(np.array(Image.open(bench.REAL_PAIR[1]).convert("L"), dtype=np.uint8))
    rl, cl = synth.build_pyramid(ref, 4), synth.build_pyramid(cur, 4)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        rp, cp = D.upload_pyramid(rl, ctx, dev), D.upload_pyramid(cl, ctx, dev)
        for spec in sys.argv[1:]:
            model, method, n, half = spec.split(":")[:4]
            n, half = int(n), int(half)
            uv, _ = bench.real_image_features(rl[0], n, half, ctx)
            opt = F.OpticalFlowOptions()
            opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = method, half, half, n
            klt = D.DeviceKlt(model, opt, rp, cp, ctx)
            d_ref = torch.from_numpy(uv).to(dev)
            d_in, d_st = d_ref.clone(), torch.zeros(n, dtype=torch.uint8, device=dev)
            o_uv, o_st = torch.empty_like(d_ref), torch.empty_like(d_st)
            d_it = torch.zeros(n, dtype=torch.int32, device=dev)
            for _ in range(3):
                klt.track(d_ref, d_in, d_st, o_uv, o_st, d_it)
                stream.synchronize()
            it = d_it.cpu().numpy()
            a = np.fromfile(dump, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
            print(f"== {spec}: mean iterations {it.mean():.1f}, max {it.max()}")
            for i in np.argsort(-it)[:3]:
                s = a[i]
                generic = not (model == "basic" and method in ("inverse",))
                if generic:
                    print(f"  feature {i}: {it[i]} iterations, total {s[7]:.0f} ticks; slots (ref_stage setup cur_stage phaseA count chain solve): " + " ".join(f"{x:.0f}" for x in s[:7]) +
                          f"; per iteration: phaseA {s[3] / it[i]:.0f} count {s[4] / it[i]:.0f} chain {s[5] / it[i]:.0f} solve {s[6] / it[i]:.0f} window {s[2] / it[i]:.0f} rest {(s[7] - s[:7].sum()) / it[i]:.0f} total {s[7] / it[i]:.0f}")
                    continue
                inside = s[0] + s[1] + s[2] + s[3] + s[5]
                print(f"  feature {i}: {it[i]} iterations, total {s[7]:.0f} ticks; slots 0..5: {s[0]:.0f} {s[1]:.0f} {s[2]:.0f} {s[3]:.0f} [{s[4]:.0f}] {s[5]:.0f}; outside the slots {s[7] - inside:.0f}; "
                      f"per iteration: slot2 {s[2] / it[i]:.0f} slot3 {s[3] / it[i]:.0f} slot5 {s[5] / it[i]:.0f} rest {(s[7] - inside) / it[i]:.0f} total {s[7] / it[i]:.0f}")


if __name__ == "__main__":
    main()
