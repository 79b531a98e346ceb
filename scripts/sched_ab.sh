#!/bin/bash
# Experiment (GPU box): tracker launch order — longest first by the previous call's iteration counts (default) vs list order
# (FTK_KLT_SCHED=0), per BASELINE configuration.   sched_ab.sh [workloads...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for w in ${*:-config3 config4 config5_shard config2}; do
  for s in 1 0; do
    echo "== $w FTK_KLT_SCHED=$s"
    FTK_KLT_SCHED=$s python3 $ROOT/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('    ms_per_step %.4f  kernel_ms %.4f  bit_identical %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['bit_identical']))"
  done
done
