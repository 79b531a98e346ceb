#!/usr/bin/env python3
"""Per-phase cycle stamps of the float matcher's register-stationary walk (GPU box): needs the timing build
    scripts/build_variant.sh rrstamps "-DFTK_RR_STAMPS"
    FTK_LIB_PATH=feature_tracker_amd/csrc/diag/libftk_hip_rrstamps.so python scripts/rr_stamps.py [dim] [nearby]
The kernel prints, for one early and one late wave of one workgroup, the s_memtime ticks spent waiting at the barrier,
issuing the transfers, in the MFMA block and in the epilogue."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_tracker_amd import device as D  # noqa: E402
from feature_tracker_amd import synth  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nearby = len(sys.argv) > 2 and sys.argv[2] == "nearby"
n = 10000
ref, cur, _ = synth.make_float_descriptors(n, n, dim=dim, noise=0.2)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = D.context_on_stream(stream, 0)
    d_ref, d_cur = torch.from_numpy(ref).to(dev), torch.from_numpy(cur).to(dev)
    idx = torch.full((n,), -1, dtype=torch.int32, device=dev)
    rs = np.random.RandomState(1)
    kw = {}
    if nearby:
        kw = dict(pred_uv=torch.from_numpy(rs.uniform(0, 640, (n, 2)).astype(np.float32)).to(dev),
                  cur_uv=torch.from_numpy(rs.uniform(0, 640, (n, 2)).astype(np.float32)).to(dev), max_col=50, max_row=50)
    for _ in range(2):
        D.cosine_match_device(ctx, d_ref, d_cur, 0.1, idx, **kw)
        stream.synchronize()
