#!/bin/bash
# Experiment (GPU box): bench.py kernel time of several workloads for the default library and for diag variants.
#   scripts/ab.sh "config2 config5_shard" chainprio other ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WL=$1; shift
for v in "" "$@"; do
  if [ -n "$v" ]; then export FTK_LIB_PATH=$ROOT/feature_tracker_amd/csrc/diag/libftk_hip_$v.so; else unset FTK_LIB_PATH; fi
  for w in $WL; do
    python3 $ROOT/bench.py --workload $w --no-cpu-baseline --no-upload-leg --no-tree-leg --steps 100 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-12s %-14s step %7.2f us  kernel %7.2f us  bit_identical %s' % ('${v:-default}', '$w', d['ms_per_step'] * 1e3, d['roofline']['kernel_ms'] * 1e3, d['parity']['bit_identical']))"
  done
done
