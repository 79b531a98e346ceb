#!/bin/bash
# Quick instruction-mix PMC pass for bench.py (GPU box): scripts/pmc_quick.sh <tag> [bench args...]
set -u
TAG=${1:-q}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline $*"
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/${name}.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,statistics,collections
vals=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'klt_track' in r['Kernel_Name']: vals[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(vals.items()): print(f"{k:24s} {statistics.median(v):14.0f}")
PY
