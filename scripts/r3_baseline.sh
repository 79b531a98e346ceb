#!/bin/bash
# Round-3 first GPU call: GPU suite, bench at the driver's shape and at 200 steps, the reference's own programs, all configs.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out
cd $ROOT
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log
tail -3 $O/gpu_tests.log
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench20 rc=$?"
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
bash scripts/run_dropin_programs.sh > $O/dropin_programs.txt 2>&1; echo "dropin rc=$?"
python scripts/bench_configs.py --quick > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "configs rc=$?"
