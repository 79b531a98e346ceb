#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats over scripts/bench_configs.py (every tracker variant, producers, both matchers, the direct
# method) -> gpurun_out/all_kernels_stats.csv with one line per ftk:: kernel (calls, average / min / max duration).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tak && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tak -- python3 $ROOT/scripts/bench_configs.py --quick > /tmp/tak.log 2>&1
f=$(find /tmp/tak -name '*kernel_stats.csv' | head -1)
mkdir -p $ROOT/gpurun_out
python3 - "$f" "$ROOT/gpurun_out/all_kernels_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'ftk::' in r['Name']]
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
with open(sys.argv[2], 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'StdDev'])
    for r in rows:
        w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['MinNs'], r['MaxNs'], r['StdDev']])
print(len(rows), 'kernels')
PY
