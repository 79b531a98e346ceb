#!/bin/bash
# GPU box: per-kernel durations of the Hamming matcher bench (rocprofv3 --kernel-trace --stats), optional env assignments as arguments
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for a in "$@"; do export "$a"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trm && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trm -- python3 $ROOT/scripts/bench_configs.py --only match --quick > /tmp/trm.log 2>&1
f=$(find /tmp/trm -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-110s calls %5s avg %9.1f ns  min %8s max %8s' % (r['Name'][:110], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
PY
