#!/usr/bin/env python3
"""Hamming matcher per-call time by problem shape and scan kernel (GPU box): where the matrix-core scan starts to pay.
    python scripts/match_shapes.py            -> one line per (n_ref, n_cur, bits, mode) with the time under each FTK_MATCH_KERNEL
Every timed result is compared with the popcount scan's indices (must be identical)."""
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
    shapes = [(300, 300, 256), (1000, 1000, 256), (2000, 2000, 256), (4000, 4000, 256), (10000, 10000, 256), (200, 20000, 256), (20000, 200, 256),
              (2000, 2000, 512), (10000, 10000, 512)]
    rs = np.random.RandomState(5)
    with torch.cuda.stream(stream):
        ctx = D.context_on_stream(stream, 0)
        for n_ref, n_cur, bits in shapes:
            cur = rs.randint(0, 2, (n_cur, bits)).astype(np.uint8)
            src = rs.randint(0, n_cur, n_ref)
            ref = cur[src].copy()
            flip = rs.rand(n_ref, bits) < (20.0 / bits)
            ref ^= flip.astype(np.uint8)
            d_ref = torch.from_numpy(F.pack_brief(ref).view(np.int32)).to(dev)
            d_cur = torch.from_numpy(F.pack_brief(cur).view(np.int32)).to(dev)
            cur_uv = torch.from_numpy(rs.uniform(0, 640, (n_cur, 2)).astype(np.float32)).to(dev)
            pred_uv = torch.from_numpy(rs.uniform(0, 640, (n_ref, 2)).astype(np.float32)).to(dev)
            for nearby in (False, True):
                out = {}
                want = None
                for kernel in ("scalar", "mfma"):
                    os.environ["FTK_MATCH_KERNEL"] = kernel
                    F.refresh_env_switches()  # the switches are read once per context
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
                    if want is None:
                        want = got
                    out[kernel] = (float(np.median(times)), bool(np.array_equal(got, want)))
                print(f"{n_ref:6d} x {n_cur:6d} x {bits:3d} {'nearby' if nearby else 'force ':6s}  scalar {out['scalar'][0]:7.1f} us   mfma {out['mfma'][0]:7.1f} us"
                      f"   same indices {out['mfma'][1]}   matched {(want >= 0).sum()}", flush=True)


if __name__ == "__main__":
    main()
