#!/bin/bash
# Experiment (GPU box): Hamming matcher 10 000 x 10 000 BRIEF-256 kernel time vs workgroup count / library variant.
#   match_sweep.sh [variant tags built by build_variant.sh ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "== $*"; env "$@" python3 $ROOT/scripts/bench_configs.py --only match --quick 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('   ', d['case'], 'gpu_kernel_ms %.4f' % d['gpu_kernel_ms'], 'exact', d['indices_bit_exact_on_sample'])"; }
run FTK_NOP=1
for wgs in 1024 1536 2048 3072 6144 8192; do run FTK_MATCH_WGS=$wgs; done
run FTK_MATCH_WGS=2048 FTK_MATCH_ANY_PER=1
for extra in "$@"; do run FTK_LIB_PATH=$ROOT/feature_tracker_amd/csrc/diag/libftk_hip_$extra.so; run FTK_LIB_PATH=$ROOT/feature_tracker_amd/csrc/diag/libftk_hip_$extra.so FTK_MATCH_WGS=2048; done
