#!/bin/bash
# Diagnostic build with per-phase s_memtime stamps: feature_tracker_amd/csrc/diag/libftk_hip_stamps.so
# Use:  FTK_LIB_PATH=$PWD/feature_tracker_amd/csrc/diag/libftk_hip_stamps.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline
exec "$(dirname "$0")/build_variant.sh" stamps "-DFTK_STAMPS $EXTRA"
