#!/bin/bash
# Round-3 measurement pass on the GPU box (through gpurun, from the repo root).  Part A: tests, bench lines, drop-in programs,
# all configurations, longest features; part B: rocprofv3 trace + PMC passes, soaks.  Outputs under gpurun_out/r3/, condensed
# into profiles/ afterwards (scripts/summarize_profile.py r3 + copies).
set -u
PART=${1:-A}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3
mkdir -p $O
cd $ROOT
if [ "$PART" = A ]; then
  python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"
  python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench20 rc=$?"
  python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  bash scripts/run_dropin_programs.sh > $O/dropin_programs.txt 2>&1; echo "dropin rc=$?"
  python scripts/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "configs rc=$?"
  (for w in config3 config2 config4 config5_shard; do for k in 1 64 256; do python scripts/longest_features.py $w $k only 2>/dev/null | tail -1; done; done) > $O/longest_features.txt 2>&1; echo "longest rc=$?"
  PYTHONPATH=. python scripts/host_call_latency.py > $O/host_call_latency.txt 2>&1; echo "host latency rc=$?"
elif [ "$PART" = B ]; then
  bash scripts/profile_all.sh r3 > $O/profile_all.log 2>&1; echo "profile rc=$?"
  bash scripts/trace_all_kernels.sh > $O/trace_all.log 2>&1; cp gpurun_out/all_kernels_stats.csv $O/ 2>/dev/null; echo "trace all rc=$?"
elif [ "$PART" = C ]; then
  python scripts/soak_parity.py 600 2027 > $O/soak_parity.txt 2>&1; echo "soak rc=$?"
  python scripts/soak_parity.py 120 31 tree > $O/soak_tree.txt 2>&1; echo "soak tree rc=$?"
  python scripts/soak_parity.py 120 41 matcher > $O/soak_matcher.txt 2>&1; echo "soak matcher rc=$?"
fi
