#!/bin/bash
# first-use costs: the reference's programs (what they print) + ftk_warmup itself in a fresh process
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash scripts/run_dropin_programs.sh 2>&1 | head -24
python - <<'PY'
import time, feature_tracker_amd as F
t0=time.perf_counter(); ctx=F.Context(0); t1=time.perf_counter()
for m in (1,2,4,8,16,31):
    t=time.perf_counter(); ctx.warmup(m); print("warmup mask",m,"%.2f ms"%((time.perf_counter()-t)*1e3))
print("context create %.2f ms"%((t1-t0)*1e3))
PY
FTK_NO_WARMUP=1 bash scripts/run_dropin_programs.sh 2>&1 | head -12
