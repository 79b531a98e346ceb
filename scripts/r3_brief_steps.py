"""Where does the reference's BRIEF test spend its 1.7 ms?  Times each host call of its flow (fresh process, after warm-up)."""
import time
import numpy as np
import feature_tracker_amd as F
from feature_tracker_amd import synth

ctx = F.Context(0)
ctx.warmup()
img, img2 = synth.make_image_pair(752, 480, (2.0, 1.0))
rs = np.random.RandomState(1)
uv = np.stack([rs.uniform(30, 720, 159), rs.uniform(30, 450, 159)], 1).astype(np.float32)


def t(label, fn, reps=1):
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    print(f"{label:40s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms")
    return r


for rnd in range(3):
    print("round", rnd)
    p = t("pyramid upload 1 level", lambda: F.ImagePyramid.from_host_levels([img], ctx))
    d = F.BriefDescriptor(ctx)
    w = t("brief compute_packed (incl upload)", lambda: d.compute_packed(img, uv))
    w2 = t("brief compute_packed (incl upload) #2", lambda: d.compute_packed(img2, uv))
    t("pyramid close", lambda: p.close())
    m = F.BriefMatcher(ctx)
    m.options().kMaxValidDescriptorDistance = 60
    m.options().kMaxValidPredictRowDistance = m.options().kMaxValidPredictColDistance = 50
    t("NearbyMatch 159x159", lambda: m.NearbyMatch(w, w2, uv, uv))
    t("ForceMatch 159x159", lambda: m.ForceMatch(w, w2))
