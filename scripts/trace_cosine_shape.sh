#!/bin/bash
# Per-kernel durations of ONE float-matcher shape (GPU box): scripts/trace_cosine_shape.sh <n_ref> <n_cur> <dim> [nearby]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/cos_shape_$1x$2x$3${4:-}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 $ROOT/scripts/cosine_one_shape.py "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,collections,statistics,sys,re
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m=re.search(r'(cosine_\w+(<[^>]*>)?)',r["Kernel_Name"])
    if m:
        d[(m.group(1),r.get("Grid_Size_X") or r.get("Grid_Size"),r.get("Workgroup_Size_X") or "")].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items()): print(k, len(v), "median %.1f us  min %.1f  max %.1f"%(statistics.median(v),min(v),max(v)))
PY
