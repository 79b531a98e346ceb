#!/bin/bash
# Round-5 measurement pass on the GPU box (through gpurun, from the repo root).  Part A: tests, bench lines, all configurations, the
# variant / small-call tables; part B: rocprofv3 trace + PMC passes; part C: soaks, drop-in programs.  Outputs under gpurun_out/r5/,
# condensed into profiles/ afterwards (scripts/summarize_profile.py r5 + copies).
set -u
PART=${1:-A}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r5
mkdir -p $O
cd $ROOT
if [ "$PART" = A ]; then
  timeout -k 10 600 python -m pytest tests -q -m gpu --timeout 300 > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"
  python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench20 rc=$?"
  python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  python scripts/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "configs rc=$?"
  python scripts/time_variant.py basic:fast:2000:6 basic:fast:10000:6 affine:fast:300:6 affine:fast:2000:6 affine:fast:5000:6 lssd:fast:2000:6 lssd:fast:10000:6 lssd:fast:10000:6:lum \
      basic:inverse:2000:6 basic:direct:2000:6 affine:inverse:2000:6 affine:direct:2000:6 lssd:inverse:2000:6 lssd:direct:2000:6 --steps 50 > $O/variants.jsonl 2>&1; echo "variants rc=$?"
  python scripts/time_variant.py basic:inverse:300:6 basic:direct:300:6 basic:fast:300:6 affine:inverse:300:6 affine:direct:300:6 affine:fast:300:6 lssd:inverse:300:6 lssd:direct:300:6 lssd:fast:300:6 \
      basic:inverse:2000:6 basic:direct:2000:6 basic:fast:2000:6 affine:inverse:2000:6 affine:direct:2000:6 affine:fast:2000:6 lssd:inverse:2000:6 lssd:direct:2000:6 lssd:fast:2000:6 \
      --real --steps 100 > $O/variants_real.jsonl 2>&1; echo "variants real rc=$?"
  python scripts/match_small_ab.py 256 > $O/match_small_ab.txt 2>&1; echo "match small rc=$?"
  python scripts/cosine_small_ab.py 256 128 > $O/cosine_small_ab.txt 2>&1; echo "cosine small rc=$?"
elif [ "$PART" = B ]; then
  bash scripts/profile_all.sh r5 > $O/profile_all.log 2>&1; echo "profile rc=$?"
  bash scripts/trace_all_kernels.sh > $O/trace_all.log 2>&1; cp gpurun_out/all_kernels_stats.csv $O/ 2>/dev/null; echo "trace all rc=$?"
elif [ "$PART" = C ]; then
  python scripts/soak_parity.py 360 2029 > $O/soak_parity.txt 2>&1; echo "soak rc=$?"
  python scripts/soak_parity.py 100 41 matcher > $O/soak_matcher.txt 2>&1; echo "soak matcher rc=$?"
  python scripts/soak_parity.py 200 4242 direct > $O/soak_direct_batches.txt 2>&1; echo "soak direct batches rc=$?"
  bash scripts/run_dropin_programs.sh > $O/dropin_programs.txt 2>&1; echo "dropin rc=$?"
  PYTHONPATH=. python scripts/host_call_latency.py > $O/host_call_latency.txt 2>&1; echo "host latency rc=$?"
fi
