#!/usr/bin/env python3
"""Timeline of one Hamming ForceMatch launch from the -DFTK_MATCH_STAMPS build (scripts/build_variant.sh mstamps "-DFTK_MATCH_STAMPS"):
per-workgroup start / descriptors-loaded / end (s_memrealtime, 100 MHz) and the CU each ran on.
    FTK_LIB_PATH=feature_tracker_amd/csrc/diag/libftk_hip_mstamps.so python scripts/match_stamps.py"""
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
    st = np.fromfile(dump, dtype=np.uint64).reshape(-1, 4)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    start, end = (st[:, 0] - t0) * 0.01, (st[:, 2] - t0) * 0.01
    loaded = start  # slot 1 now carries shader-clock ticks over the workgroup's life
    clock_mhz = st[:, 1].astype(np.float64) / np.maximum(end - start, 1e-9)
    print(f"shader clock over workgroup lives: median {np.median(clock_mhz):.0f} MHz, p10 {np.percentile(clock_mhz, 10):.0f}, p90 {np.percentile(clock_mhz, 90):.0f}")
    hw = st[:, 3]
    args_us = (hw >> 40).astype(np.float64) * 0.01
    print(f"kernel arguments available after: median {np.median(args_us):.2f} us, p90 {np.percentile(args_us, 90):.2f}, max {args_us.max():.2f}")
    xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    cu_key = xcc * 1000 + se * 100 + sh * 10 + cu
    print(f"workgroups {len(st)}, span {end.max():.1f} us; start: median {np.median(start):.1f}, p90 {np.percentile(start, 90):.1f}, max {start.max():.1f} us")
    print(f"life: median {np.median(end - start):.1f} us (min {np.min(end - start):.1f}, max {np.max(end - start):.1f}); descriptor load: median {np.median(loaded - start):.2f} us")
    print(f"distinct CUs seen {len(np.unique(cu_key))}; workgroups per CU: min {np.bincount(np.unique(cu_key, return_inverse=True)[1]).min()}, max {np.bincount(np.unique(cu_key, return_inverse=True)[1]).max()}")
    early = start < 2.0
    print(f"first-round workgroups ({int(early.sum())}): load median {np.median((loaded - start)[early]):.2f} us, p10 {np.percentile((loaded - start)[early], 10):.2f}, p90 "
          f"{np.percentile((loaded - start)[early], 90):.2f}; compute (loaded -> end) median {np.median((end - loaded)[early]):.1f} us")
    late = ~early
    if late.any():
        print(f"later workgroups ({int(late.sum())}): load median {np.median((loaded - start)[late]):.2f} us, p90 {np.percentile((loaded - start)[late], 90):.2f}; "
              f"compute median {np.median((end - loaded)[late]):.1f} us")
    # resident workgroups over time
    ts = np.linspace(0, end.max(), 25)
    print("t(us): resident workgroups")
    for t in ts:
        print(f"  {t:6.1f}: {int(((start <= t) & (end > t)).sum())}")
    # per CU finish time
    fin = {}
    for k, e in zip(cu_key, end):
        fin[k] = max(fin.get(k, 0), e)
    f = np.array(list(fin.values()))
    print(f"per-CU last finish: min {f.min():.1f}, median {np.median(f):.1f}, max {f.max():.1f} us")


if __name__ == "__main__":
    main()
