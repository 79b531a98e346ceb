#!/bin/bash
# Pipelined Basic-inverse kernel: the ring pitch (64 + pad floats) under the quad chain — time and LDS bank conflicts, config 2, same box
D=feature_tracker_amd/csrc/diag
for lib in $D/libftk_hip_pad4.so $D/libftk_hip_pad8.so feature_tracker_amd/csrc/libftk_hip.so $D/libftk_hip_pad24.so; do
  echo "=== $lib"
  FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:2000:10 basic:inverse:2000:6 --steps 200 --no-oracle 2>&1 | grep spec
  export FTK_LIB_PATH=$lib
  bash scripts/pmc_variant.sh pad basic:inverse:2000:10 --steps 20 2>/dev/null | grep -E "pipelined|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_WAIT_ANY |SQ_WAVE_CYCLES|SQ_INSTS_LDS|SQ_INSTS_VALU "
  unset FTK_LIB_PATH
done
