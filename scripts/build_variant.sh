#!/bin/bash
# Experiment build: feature_tracker_amd/csrc/diag/libftk_hip_<tag>.so with extra compiler flags.
#   scripts/build_variant.sh <tag> "<extra flags>"     use with FTK_LIB_PATH=.../diag/libftk_hip_<tag>.so
set -e
TAG=$1; EXTRA=${2:-}
cd "$(dirname "$0")/../feature_tracker_amd/csrc"
T=$(mktemp -d)
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero $EXTRA"
pids=""
for f in float_matcher_kernels pyramid_kernels feature_kernels; do hipcc $F -c -o $T/$f.o $f.hip & pids="$pids $!"; done
for f in klt_kernels klt_basic_kernels klt_fast_kernels; do hipcc $F -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -structurizecfg-skip-uniform-regions=1 -mllvm -disable-lsr -c -o $T/$f.o $f.hip & pids="$pids $!"; done  # as in the Makefile
hipcc $F -mllvm -amdgpu-sched-strategy=max-ilp -c -o $T/direct_kernels.o direct_kernels.hip & pids="$pids $!"
hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -c -o $T/matcher_kernels.o matcher_kernels.hip & pids="$pids $!"  # as in the Makefile
hipcc $F -x hip -c -o $T/ftk_api.o ftk_api.cpp & pids="$pids $!"
hipcc $F -x hip -c -o $T/ftk_comm.o ftk_comm.cpp
make -s ftk_build_info.inc && hipcc -O2 -std=c++17 -fPIC -c -o $T/ftk_build_info.o ftk_build_info.cpp  # the hash is the tree's; the tag names the variant
for p in $pids; do wait $p; done  # a failed compile fails the script (set -e)
mkdir -p diag
hipcc -shared -fPIC --offload-arch=gfx950 -o diag/libftk_hip_$TAG.so $T/*.o -ldl
rm -rf $T
