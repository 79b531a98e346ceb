#!/bin/bash
# Experiment (GPU box): FETCH_SIZE / TCC hit counters of one workload's dominant kernel for the default library and diag variants.
#   scripts/fetch_ab.sh config5_shard base morton
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
W=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "" "$@"; do
  if [ -n "$v" ]; then export FTK_LIB_PATH=$ROOT/feature_tracker_amd/csrc/diag/libftk_hip_$v.so; else unset FTK_LIB_PATH; fi
  D=/tmp/fetch_ab_${v:-default}; rm -rf $D
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rocprofv3 --pmc $grp --output-format csv -d $D/$(echo $grp | cut -d' ' -f1) -- python3 $ROOT/bench.py --workload $W --steps 30 --warmup 3 --no-cpu-baseline --no-upload-leg --no-tree-leg > /dev/null 2>&1
  done
  python3 - "$D" "${v:-default}" <<'PY'
import sys, glob, csv, statistics
d, tag = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "klt_" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(tag, {k: statistics.median(v) for k, v in vals.items()})
PY
done
