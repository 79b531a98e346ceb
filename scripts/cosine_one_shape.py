#!/usr/bin/env python3
"""One float-matcher shape, a few calls (for rocprofv3 --kernel-trace: scripts/trace_cosine_shape.sh): n_ref n_cur dim [nearby]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from feature_tracker_amd import device as D, synth
n_ref, n_cur, dim = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nearby = len(sys.argv) > 4 and sys.argv[4] == "nearby"
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
rs = np.random.RandomState(5)
with torch.cuda.stream(stream):
    ctx = D.context_on_stream(stream, 0)
    ref, cur, _ = synth.make_float_descriptors(n_ref, n_cur, dim=dim)
    d_ref, d_cur = torch.from_numpy(ref).to(dev), torch.from_numpy(cur).to(dev)
    cur_uv = torch.from_numpy(rs.uniform(0, 640, (n_cur, 2)).astype(np.float32)).to(dev)
    pred_uv = torch.from_numpy(rs.uniform(0, 640, (n_ref, 2)).astype(np.float32)).to(dev)
    d_idx = torch.full((n_ref,), -1, dtype=torch.int32, device=dev)
    for _ in range(6):
        D.cosine_match_device(ctx, d_ref, d_cur, 0.2, d_idx, pred_uv=pred_uv if nearby else None, cur_uv=cur_uv if nearby else None, max_col=60, max_row=60)
    stream.synchronize()
    print("matched", int((d_idx >= 0).sum().item()))
