#!/bin/bash
# Experiment (GPU box): arrival trace + launch time of spread direct-method consumer variants (docs/experiments/direct_spread_consumer_variants.diff.txt)
D=feature_tracker_amd/csrc/diag
for v in "$@"; do
  echo "=== $v"
  FTK_LIB_PATH=$D/libftk_hip_$v.so timeout -k 10 120 python scripts/dm_trace.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 || exit 1
  FTK_LIB_PATH=$D/libftk_hip_$v.so timeout -k 10 120 python scripts/direct_batch_time.py 1 2 6 2>&1 | grep -v amdgpu.ids || exit 1
done
