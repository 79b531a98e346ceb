"""Diagnostic (GPU box): bench.py's N > 1 step — tracker launch, RCCL all-gather on a side stream, two result slots — captured into a
HIP graph in isolation.  MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 python scripts/capture_debug.py [full|nocoll|nowait] [K] [thread_local]"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import feature_tracker_amd as F
from feature_tracker_amd import device as D, dist as FD, synth
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = dict(synth.CONFIGS["config2"]); n, w, h, levels, half = cfg["n"], cfg["width"], cfg["height"], cfg["levels"], cfg["half"]
ref_img, cur_img = synth.make_image_pair(w, h, (3.3, -2.1))
uv = synth.make_features(n, w, h, seed=12345, half=half)
stream = torch.cuda.Stream(device=dev)
def status(tag):
    try:
        st = torch.cuda.is_current_stream_capturing()
    except Exception as e:
        st = "EXC " + str(e)[:80]
    print(tag, "capturing:", st, flush=True)
with torch.cuda.stream(stream):
    ctx = D.context_on_stream(stream, 0)
    ref_pyr = D.upload_pyramid(synth.build_pyramid(ref_img, levels), ctx, dev)
    cur_pyr = D.upload_pyramid(synth.build_pyramid(cur_img, levels), ctx, dev)
    opt = F.OpticalFlowOptions(); opt.kMethod = cfg["method"]; opt.kPatchRowHalfSize = opt.kPatchColHalfSize = half; opt.kMaxTrackPointsNumber = n
    klt = D.DeviceKlt(cfg["model"], opt, ref_pyr, cur_pyr, ctx)
    d_ref = torch.from_numpy(uv).to(dev); d_in = d_ref.clone(); d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
    packed2 = [torch.zeros(FD.packed_bytes(n), dtype=torch.uint8, device=dev) for _ in range(2)]
    views2 = [FD.pack_views(pk, n) for pk in packed2]
    gathered2 = [torch.empty(FD.packed_bytes(n), dtype=torch.uint8, device=dev) for _ in range(2)]
    launches = [klt.bind(d_ref, d_in, d_st, views2[s][0], views2[s][1], None) for s in range(2)]
    for s in range(2):
        launches[s](); FD.all_gather_results(packed2[s], 1, force_collective=True, out=gathered2[s])
    stream.synchronize()
    mode = sys.argv[1] if len(sys.argv) > 1 else "full"
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    kw = {"capture_error_mode": sys.argv[3]} if len(sys.argv) > 3 else {}
    try:
        with torch.cuda.graph(g, stream=stream, **kw):
            gd = {}
            for k in range(K):
                slot = k & 1
                if k >= 2 and mode != "nowait":
                    stream.wait_event(gd[k - 2]); pass
                launches[slot]()
                kd = torch.cuda.Event(); kd.record(stream); side.wait_event(kd)
                with torch.cuda.stream(side):
                    if mode != "nocoll":
                        dist.all_gather_into_tensor(gathered2[slot], packed2[slot])
                    else:
                        gathered2[slot].copy_(packed2[slot])
                    gd[k] = torch.cuda.Event(); gd[k].record(side)
            stream.wait_stream(side); status("join")
        g.replay(); torch.cuda.synchronize(); print("OK", mode)
    except Exception as e:
        print("FAILED", mode, type(e).__name__, str(e)[:160])
