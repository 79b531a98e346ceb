#!/usr/bin/env python3
"""Hamming matcher, small calls (GPU box): the one-launch form (matcher_kernels.hip hamming_match_small_kernel) against the
boxes + scan + epilogue launches, per shape; indices compared.   python scripts/match_small_ab.py [bits ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import feature_tracker_amd as F
    from feature_tracker_amd import device as D
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    widths = [int(x) for x in sys.argv[1:]] or [256]
    shapes = [(100, 100), (300, 300), (1000, 1000), (2000, 2000), (3000, 3000), (300, 3000), (3000, 300), (64, 60000), (5000, 5000)]
    rs = np.random.RandomState(5)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        for bits in widths:
            for n_ref, n_cur in shapes:
                cur = rs.randint(0, 2, (n_cur, bits)).astype(np.uint8)
                ref = cur[rs.randint(0, n_cur, n_ref)].copy()
                ref ^= (rs.rand(n_ref, bits) < (20.0 / bits)).astype(np.uint8)
                d_ref = torch.from_numpy(F.pack_brief(ref).view(np.int32)).to(dev)
                d_cur = torch.from_numpy(F.pack_brief(cur).view(np.int32)).to(dev)
                cur_uv = torch.from_numpy(rs.uniform(0, 640, (n_cur, 2)).astype(np.float32)).to(dev)
                pred_uv = torch.from_numpy(rs.uniform(0, 640, (n_ref, 2)).astype(np.float32)).to(dev)
                for nearby in (False, True):
                    out, want = {}, None
                    for small in ("0", "1"):
                        os.environ["FTK_MATCH_SMALL"] = small
                        ctx.refresh_env()  # the switches are read once per context
                        d_idx = torch.full((n_ref,), -1, dtype=torch.int32, device=dev)
                        args = dict(pred_uv=pred_uv if nearby else None, cur_uv=cur_uv if nearby else None, max_col=60, max_row=60)
                        for _ in range(3):
                            D.hamming_match_device(ctx, d_ref, d_cur, bits, 60.0, d_idx, **args)
                        stream.synchronize()
                        times = []
                        for _ in range(30):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(stream)
                            D.hamming_match_device(ctx, d_ref, d_cur, bits, 60.0, d_idx, **args)
                            e1.record(stream)
                            e1.synchronize()
                            times.append(e0.elapsed_time(e1) * 1e3)
                        got = d_idx.cpu().numpy()
                        want = got if want is None else want
                        out[small] = (float(np.median(times)), bool(np.array_equal(got, want)))
                    print(f"{n_ref:6d} x {n_cur:6d} x {bits:3d} {'nearby' if nearby else 'force ':6s}  launches {out['0'][0]:7.1f} us   one launch {out['1'][0]:7.1f} us"
                          f"   same indices {out['1'][1]}   matched {(want >= 0).sum()}", flush=True)


if __name__ == "__main__":
    main()
