#!/bin/bash
# Experiment (GPU box): time of a launch with FEW features (all resident: the launch is one feature's latency) by waves per feature.
#   latency_by_waves.sh <workload> <features>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for wv in 1 2 3 4; do
  FTK_KLT_WAVES=$wv python3 $ROOT/bench.py --workload ${1:-config3} --features ${2:-256} --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('waves per feature $wv: %.1f us per launch, mean iterations %s' % (d['ms_per_step'] * 1e3, d['config'].get('mean_iterations_per_feature')))"
done
