#!/bin/bash
# rocprofv3 evidence for every hot kernel of the path (GPU box, through gpurun from the repo root):
#   scripts/profile_all.sh <tag> [workloads...]     -> gpurun_out/prof_<tag>/<workload>/{trace,pmc_*}/...
# Workloads: config2 (headline, klt_basic_inverse_pipelined_kernel), config3 (klt_track_kernel<affine, inverse>),
# config4 (klt_track_kernel<lssd, fast>), config5_shard, hamming (hamming_match_mfma_kernel, 10 000 x 10 000 BRIEF-256).
# One run with --kernel-trace --stats; every PMC group in a run of its own (never combined with tracing).
# scripts/summarize_profile.py <tag> condenses the result into profiles/.
set -u
TAG=${1:-r2}; shift || true
WORKLOADS=${*:-config2 config3 config4 config5_shard hamming}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
# which build these counters belong to: bench.py refuses committed PMC figures whose source hash is not the running library's
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from feature_tracker_amd import _native as N; import json; print(json.dumps(N.build_info()))" > "$OUT/build_info.json"
cd /tmp && export TMPDIR=/tmp
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS"
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_BUSY_CU_CYCLES")
for W in $WORKLOADS; do
  D=$OUT/$W; rm -rf "$D"; mkdir -p "$D"
  if [ "$W" = hamming ]; then
    CMD="python3 $ROOT/scripts/bench_configs.py --only match --quick"
  else
    # (the legs that launch OTHER workloads or isolated calls are off: the trace average of the workload's kernel is then the back-to-back figure bench.py times)
    CMD="python3 $ROOT/bench.py --workload $W --steps 30 --warmup 3 --no-cpu-baseline --no-upload-leg --no-tree-leg --no-configs-leg --no-host-call-leg --no-real-images-leg"
  fi
  echo "== $W: trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- $CMD > "$D/trace_stdout.log" 2>&1 || echo "trace failed: $W" >> "$OUT/errors.log"
  i=0
  for grp in "${GROUPS_[@]}"; do
    echo "== $W: pmc $grp"
    rocprofv3 --pmc $grp --output-format csv -d "$D/pmc_$i" -- $CMD > "$D/pmc_${i}_stdout.log" 2>&1 || echo "pmc group failed: $W: $grp" >> "$OUT/errors.log"
    i=$((i+1))
  done
done
# keep the box->repo merge small: drop everything but csv/log
find "$OUT" -type f ! -name '*.csv' ! -name '*.log' ! -name 'build_info.json' -delete
du -sh "$OUT"
