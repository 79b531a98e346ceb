#!/bin/bash
# Diagnostic build with per-phase s_memtime stamps: feature_tracker_amd/csrc/diag/libftk_hip_stamps.so
# Use:  FTK_LIB_PATH=$PWD/feature_tracker_amd/csrc/diag/libftk_hip_stamps.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline
set -e
cd "$(dirname "$0")/../feature_tracker_amd/csrc"
T=$(mktemp -d)
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -DFTK_STAMPS $EXTRA"
for f in klt_kernels klt_basic_kernels matcher_kernels float_matcher_kernels direct_kernels pyramid_kernels feature_kernels; do hipcc $F -c -o $T/$f.o $f.hip & done
hipcc $F -x hip -c -o $T/ftk_api.o ftk_api.cpp
wait
mkdir -p diag
hipcc -shared -fPIC --offload-arch=gfx950 -o diag/libftk_hip_stamps.so $T/*.o
rm -rf $T
