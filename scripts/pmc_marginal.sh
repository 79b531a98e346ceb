#!/bin/bash
# Instruction counts of one tracker launch for a few (levels, kMaxIteration) combinations, so that the
# cost of one level entry and of one Gauss-Newton iteration can be read off as differences (GPU box):
#   scripts/pmc_marginal.sh <tag>
set -u
TAG=${1:-m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcm_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for combo in "4 15" "4 1" "1 1" "1 2"; do
  set -- $combo
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/l$1_i$2" -- python3 $ROOT/scripts/marginal_costs.py 2000 $1 $2 > "$OUT/l$1_i$2.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,statistics,collections,os
for d in sorted(glob.glob(sys.argv[1]+'/l*_i*/')):
    vals=collections.defaultdict(list)
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'klt_' in r['Kernel_Name']: vals[r['Counter_Name']].append(float(r['Counter_Value']))
    print(os.path.basename(d.rstrip('/')), {k:int(statistics.median(v)) for k,v in sorted(vals.items())})
PY
