#!/bin/bash
# Register / spill / scratch use of every kernel of one translation unit, compiled for the device only with the library's flags:
#   scripts/kernel_resources.sh klt_fast_kernels.hip [extra hipcc flags]
cd "$(dirname "$0")/../feature_tracker_amd/csrc" || exit 1
src=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -structurizecfg-skip-uniform-regions=1 -mllvm -disable-lsr \
  --cuda-device-only -S -o /tmp/kernel_resources.s "$@" "$src" 2>/dev/null || exit 1
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size)|\.name:" /tmp/kernel_resources.s | paste - - - - - - | \
  sed -E 's/\s+\.name:\s+/ /; s/\.private_segment_fixed_size:/scratch/; s/\.sgpr_count:/sgpr/; s/\.sgpr_spill_count:/sspill/; s/\.vgpr_count:/vgpr/; s/\.vgpr_spill_count:/vspill/' | tr -s ' \t' ' '
