#!/bin/bash
# Instruction mix + LDS bank behaviour of one bench.py workload (GPU box): scripts/pmc_config.sh <workload> [extra bench args]
set -u
W=${1:-config3}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcw_$W
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-upload-leg --no-tree-leg $*"
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/${name}.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,statistics,collections
vals=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'klt_' in r['Kernel_Name']: vals[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(vals.items()): print(f"{k:24s} {statistics.median(v):14.0f}")
PY
