#!/usr/bin/env python3
"""Experiment (GPU box): what the pieces of bench.py's with_pyramid_upload call cost, stream-ordered, K calls per measurement:
two H2D frame copies alone; + the pyramid builds (ftk_pyramid_update); + the tracker launch.
    python scripts/upload_leg_pieces.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import feature_tracker_amd as F  # noqa: E402
from feature_tracker_amd import device as D, synth  # noqa: E402

cfg = dict(synth.CONFIGS["config2"])
n, w, h, levels, half = cfg["n"], cfg["width"], cfg["height"], cfg["levels"], cfg["half"]
ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1))
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
uv = synth.make_features(n, w, h, seed=12345, half=half)
K = 200
with torch.cuda.stream(stream):
    ctx = D.context_on_stream(stream, 0)
    pr, pc = F.ImagePyramid.build(ref_img, levels, ctx), F.ImagePyramid.build(cur_img, levels, ctx)
    opt = F.OpticalFlowOptions()
    opt.kMethod, opt.kPatchRowHalfSize, opt.kPatchColHalfSize, opt.kMaxTrackPointsNumber = "inverse", half, half, n
    klt = D.DeviceKlt("basic", opt, pr, pc, ctx)
    d_ref = torch.from_numpy(uv).to(dev)
    d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_out, d_so = torch.empty_like(d_ref), torch.empty_like(d_st)
    p_ref = torch.from_numpy(np.ascontiguousarray(ref_img)).pin_memory()
    p_cur = torch.from_numpy(np.ascontiguousarray(cur_img)).pin_memory()
    d_img = torch.empty_like(torch.from_numpy(ref_img), device=dev)
    d_img2 = torch.empty_like(d_img)

    def timed(fn):
        for _ in range(10):
            fn()
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        stream.synchronize()
        return (time.perf_counter() - t0) / K * 1e6

    def copies():
        d_img.copy_(p_ref, non_blocking=True)
        d_img2.copy_(p_cur, non_blocking=True)

    def updates():
        pr.update(p_ref.data_ptr(), "host_async")
        pc.update(p_cur.data_ptr(), "host_async")

    def updates_device():
        pr.update(d_img.data_ptr(), "device")
        pc.update(d_img2.data_ptr(), "device")

    def track():
        klt.track(d_ref, d_ref, d_st, d_out, d_so, None)

    def all_of_it():
        updates()
        track()

    print(f"two H2D copies of {w}x{h} (torch, pinned):      {timed(copies):7.1f} us per call")
    print(f"two ftk_pyramid_update from pinned host memory:  {timed(updates):7.1f} us")
    print(f"two ftk_pyramid_update from device memory:       {timed(updates_device):7.1f} us")
    print(f"tracker launch alone:                            {timed(track):7.1f} us")
    print(f"updates + tracker (the bench leg):               {timed(all_of_it):7.1f} us")

    # host-side cost of one update call (launches are asynchronous: the loop below measures the caller's thread, not the GPU)
    stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        pr.update(p_ref.data_ptr(), "host_async")
    t1 = time.perf_counter()
    stream.synchronize()
    print(f"host thread per ftk_pyramid_update(host_async) call: {(t1 - t0) / 50 * 1e6:5.1f} us")
