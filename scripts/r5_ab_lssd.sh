#!/bin/bash
# A / B, same box: diag/base (the committed build) against the working tree's library, LSSD / generic-kernel variants
D=feature_tracker_amd/csrc/diag
V=${V:-"lssd:inverse lssd:direct"}
for rep in 1 2; do
for lib in $D/libftk_hip_base.so feature_tracker_amd/csrc/libftk_hip.so; do
  echo "=== $lib"
  S=""; R3=""; R2=""
  for v in $V; do S="$S $v:2000:6 $v:5000:6"; R3="$R3 $v:300:6"; R2="$R2 $v:2000:6"; done
  FTK_LIB_PATH=$lib python scripts/time_variant.py $S --steps 100 || exit 1
  echo "--- real"; FTK_LIB_PATH=$lib python scripts/time_variant.py $R3 $R2 --real --steps 100 || exit 1
done
done
