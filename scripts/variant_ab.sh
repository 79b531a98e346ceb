#!/bin/bash
# Experiment (GPU box): library variants (scripts/build_variant.sh <tag> "<flags>") against the default build, per workload.
#   variant_ab.sh "<tag> <tag> ..." [workloads...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAGS=$1; shift
for w in ${*:-config3 config4}; do
  for t in default $TAGS; do
    echo "== $w $t"
    if [ $t = default ]; then unset FTK_LIB_PATH; else export FTK_LIB_PATH=$ROOT/feature_tracker_amd/csrc/diag/libftk_hip_$t.so; fi
    python3 $ROOT/bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('    ms_per_step %.4f  bit_identical %s' % (d['ms_per_step'], d['parity']['bit_identical']))"
  done
done
