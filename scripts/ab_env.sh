#!/bin/bash
# Experiment (GPU box): bench.py kernel time per workload under different environment settings.
#   scripts/ab_env.sh "config5_shard config2" "FTK_KLT_GROUP=1" "FTK_KLT_GROUP=2" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WL=$1; shift
for e in "$@"; do
  for w in $WL; do
    env $e python3 $ROOT/bench.py --workload $w --no-cpu-baseline --no-upload-leg --no-tree-leg --steps 100 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-28s %-14s step %7.2f us  kernel %7.2f us  bit_identical %s' % ('$e', '$w', d['ms_per_step'] * 1e3, d['roofline']['kernel_ms'] * 1e3, d['parity']['bit_identical']))"
  done
done
