#!/bin/bash
# Builds the stamps variant and runs scripts/stamps_fast.py on the GPU box for each "model method n half" line given as arguments:
#   scripts/stamps_run.sh "basic fast 2000 6" "basic fast 2000 10"
cd "$(dirname "$0")/.." || exit 1
EXTRA="" bash scripts/build_stamps.sh 2>&1 | grep -E "error" && exit 1
cmd='export FTK_LIB_PATH=$PWD/feature_tracker_amd/csrc/diag/libftk_hip_stamps.so'
for c in "$@"; do cmd="$cmd; echo \"== $c\"; python scripts/stamps_fast.py $c 2>&1 | tail -2"; done
/usr/local/graft/bin/gpurun --timeout 600 -- "$cmd" 2>&1 | grep -v "amdgpu.ids\|^\[gpurun\] sending"
