#!/bin/bash
# Experiment (GPU box): Hamming matcher 10 000 x 10 000 BRIEF-256 per-call time by scan kernel (FTK_MATCH_KERNEL) and switches.
#   match_ab.sh [quick]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "== $*"; env "$@" python3 $ROOT/scripts/bench_configs.py --only match --quick 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('   ', d['case'], 'gpu_kernel_ms %.4f' % d['gpu_kernel_ms'], 'exact', d['indices_bit_exact_on_sample'])"; }
run FTK_MATCH_KERNEL=mfma

run FTK_MATCH_KERNEL=scalar
[ "$1" = quick ] && exit 0
for wgs in 1024 1536 3072 4096; do run FTK_MATCH_KERNEL=mfma FTK_MATCH_WGS=$wgs; done
