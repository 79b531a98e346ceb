#!/usr/bin/env python3
"""Where a workgroup of the matrix-core Hamming scan spends its cycles (-DFTK_MATCH_STAMPS build: scripts/build_variant.sh mstamps
"-DFTK_MATCH_STAMPS"; wave 0 of every workgroup stamps its phases with s_memtime).
    FTK_LIB_PATH=feature_tracker_amd/csrc/diag/libftk_hip_mstamps.so FTK_MATCH_KERNEL=mfma python scripts/match_mfma_stamps.py"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import feature_tracker_amd as F
    from feature_tracker_amd import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    ref, cur, _ = synth.make_descriptors(n, n)
    m = F.BriefMatcher()
    m.options().kMaxValidDescriptorDistance = 60
    dump = os.path.join(tempfile.gettempdir(), "match_stamps.bin")
    os.environ["FTK_MATCH_STAMPS_DUMP"] = dump
    F.refresh_env_switches()  # the switches are read once per context
    for _ in range(3):
        m.ForceMatch(ref, cur)
    st = np.fromfile(dump, dtype=np.uint64).reshape(-1, 8)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    start, end = (st[:, 0] - t0) * 0.01, (st[:, 1] - t0) * 0.01
    print(f"workgroups {len(st)}; span {end.max():.1f} us; start median {np.median(start):.1f} max {start.max():.1f} us; life median {np.median(end - start):.1f} max {(end - start).max():.1f} us")
    clock = st[:, 7].astype(np.float64) / np.maximum(end - start, 1e-9)
    print(f"shader clock over workgroup lives: median {np.median(clock):.0f} MHz")
    full = st[st[:, 7] >= np.percentile(st[:, 7], 50)]
    names = ["prologue", "tile loop", "  of it: key updates", "final reduction"]
    tot = np.median(full[:, 7])
    print(f"median of the longer half of the workgroups: {tot:.0f} ticks total")
    for name, col in zip(names, (2, 3, 4, 6)):
        v = np.median(full[:, col])
        print(f"  {name:34s} {v:9.0f} ticks  {100 * v / tot:5.1f} %")
    print(f"  key updates per wave: median {np.median(full[:, 5]):.0f}, ticks each {np.median(full[:, 4] / np.maximum(full[:, 5], 1)):.0f}")


if __name__ == "__main__":
    main()
