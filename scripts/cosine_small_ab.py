#!/usr/bin/env python3
"""Float (cosine) matcher, small calls (GPU box): the one-launch exact form (float_matcher_kernels.hip cosine_match_small_kernel)
against the clear + prep + contraction + recheck pipeline, per shape; indices compared with each other and with the oracle.
    python scripts/cosine_small_ab.py [dim ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from feature_tracker_amd import device as D, synth
    from tests import oracle_lib
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    dims = [int(x) for x in sys.argv[1:]] or [256, 128]
    shapes = [(100, 100), (300, 300), (600, 600), (1000, 1000), (2000, 2000), (300, 3000), (3000, 300)]
    rs = np.random.RandomState(5)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        for dim in dims:
            for n_ref, n_cur in shapes:
                ref, cur, _ = synth.make_float_descriptors(n_ref, n_cur, dim=dim)
                d_ref, d_cur = torch.from_numpy(ref).to(dev), torch.from_numpy(cur).to(dev)
                cur_uv_h = rs.uniform(0, 640, (n_cur, 2)).astype(np.float32)
                pred_uv_h = rs.uniform(0, 640, (n_ref, 2)).astype(np.float32)
                cur_uv, pred_uv = torch.from_numpy(cur_uv_h).to(dev), torch.from_numpy(pred_uv_h).to(dev)
                for nearby in (False, True):
                    out, want = {}, None
                    for small in ("0", "1"):
                        os.environ["FTK_COSINE_SMALL"] = small
                        os.environ["FTK_COSINE_SMALL_ANY"] = "1"
                        ctx.refresh_env()  # the switches are read once per context
                        d_idx = torch.full((n_ref,), -1, dtype=torch.int32, device=dev)
                        args = dict(pred_uv=pred_uv if nearby else None, cur_uv=cur_uv if nearby else None, max_col=60, max_row=60)
                        for _ in range(3):
                            D.cosine_match_device(ctx, d_ref, d_cur, 0.2, d_idx, **args)
                        stream.synchronize()
                        times = []
                        for _ in range(20):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(stream)
                            D.cosine_match_device(ctx, d_ref, d_cur, 0.2, d_idx, **args)
                            e1.record(stream)
                            e1.synchronize()
                            times.append(e0.elapsed_time(e1) * 1e3)
                        got = d_idx.cpu().numpy()
                        want = got if want is None else want
                        out[small] = (float(np.median(times)), bool(np.array_equal(got, want)))
                    if n_ref * n_cur <= 400000:
                        ok, cpu = oracle_lib.match_float(ref, cur, 0.2, pred_uv_h if nearby else None, cur_uv_h if nearby else None, 60, 60)
                        oracle_same = bool(np.array_equal(cpu, want))
                    else:
                        oracle_same = None
                    print(f"{n_ref:6d} x {n_cur:6d} x {dim:3d} {'nearby' if nearby else 'force ':6s}  pipeline {out['0'][0]:7.1f} us   one launch {out['1'][0]:7.1f} us"
                          f"   same indices {out['1'][1]}   oracle {oracle_same}   matched {(want >= 0).sum()}", flush=True)


if __name__ == "__main__":
    main()
