#!/bin/bash
# Experiment (GPU box): with the longest-first launch order in place, do more resident waves pay?  (packing / waves per feature)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { w=$1; shift; echo "== $w $*"; env "$@" python3 $ROOT/bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('    ms_per_step %.4f  bit_identical %s' % (d['ms_per_step'], d['parity']['bit_identical']))"; }
for wv in 1 2 3 4; do run config4 FTK_KLT_WAVES=$wv; done
for g in 2 4; do run config4 FTK_KLT_GROUP=$g; done
run config4 FTK_LSSD_CHUNKED=0
for wv in 1 2 3 4; do run config3 FTK_KLT_WAVES=$wv; done
