#!/bin/bash
# Per-kernel durations of the float-descriptor matcher by grid (GPU box): scripts/trace_cosine.sh <tag>
set -u
TAG=${1:-t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/cos_kt_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/scripts/bench_configs.py --only cosine --quick > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,collections,statistics,sys,re
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    m=re.search(r'(cosine_\w+(<[^>]*>)?)',n)
    if m:
        d[(m.group(1),r.get("Grid_Size_X") or r.get("Grid_Size"),r.get("Grid_Size_Y"),r.get("LDS_Block_Size") or r.get("LDS_Block_Size_v",""))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items()): print(k, len(v), "median %.1f us  min %.1f"%(statistics.median(v),min(v)))
PY
