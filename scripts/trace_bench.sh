#!/bin/bash
# GPU box: per-kernel durations of bench.py for one workload (rocprofv3 --kernel-trace --stats); env assignments after the workload
#   trace_bench.sh config4 FTK_KLT_SCHED_MODE=3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
W=$1; shift
for a in "$@"; do export "$a"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trb && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trb -- python3 $ROOT/bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline > /tmp/trb.log 2>&1
f=$(find /tmp/trb -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'klt' in r['Name'] or 'order' in r['Name']:
        print('%-100s calls %5s avg %9.1f ns  min %8s max %8s' % (r['Name'][:100], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
PY
