import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import feature_tracker_amd as F
from feature_tracker_amd import synth
for name in ("config1", "config2"):
    cfg = synth.CONFIGS[name]
    ref, cur = synth.make_image_pair(cfg["width"], cfg["height"], (3.3, -2.1))
    rl, cl = synth.build_pyramid(ref, cfg["levels"]), synth.build_pyramid(cur, cfg["levels"])
    for n in (300, cfg["n"]):
        uv = synth.make_features(n, cfg["width"], cfg["height"], half=cfg["half"])
        klt = F.OpticalFlowBasicKlt(); o = klt.options(); o.kMethod = "inverse"; o.kPatchRowHalfSize = o.kPatchColHalfSize = cfg["half"]; o.kMaxTrackPointsNumber = n
        rp, cp = F.ImagePyramid.from_host_levels(rl), F.ImagePyramid.from_host_levels(cl)
        for _ in range(5): klt.TrackFeatures(rp, cp, uv)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); ok, c, s = klt.TrackFeatures(rp, cp, uv); ts.append(time.perf_counter() - t0)
        print(name, "n", n, "host call median us", round(np.median(ts) * 1e6, 1), "p10", round(np.percentile(ts, 10) * 1e6, 1), "tracked", float((s == 1).mean()),
              "checksum", int(np.ascontiguousarray(c).view(np.uint32).astype(np.uint64).sum()))
