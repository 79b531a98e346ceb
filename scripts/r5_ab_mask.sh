#!/bin/bash
# A / B, same box: quad chains with all 64 lanes reading (diag/allq, the round's evidence build) against whole active quads only (default)
D=feature_tracker_amd/csrc/diag
V="basic:inverse basic:direct basic:fast affine:fast lssd:inverse lssd:direct lssd:fast"
for lib in $D/libftk_hip_allq.so feature_tracker_amd/csrc/libftk_hip.so; do
  echo "=== $lib"
  S=""; R3=""; R2=""
  for v in $V; do S="$S $v:2000:6"; R3="$R3 $v:300:6"; R2="$R2 $v:2000:6"; done
  FTK_LIB_PATH=$lib python scripts/time_variant.py $S lssd:fast:2000:6:lum basic:inverse:2000:10 basic:inverse:200:5 --steps 100 || exit 1
  echo "--- real 300"; FTK_LIB_PATH=$lib python scripts/time_variant.py $R3 --real --steps 100 || exit 1
  echo "--- real 2000"; FTK_LIB_PATH=$lib python scripts/time_variant.py $R2 --real --steps 100 || exit 1
  echo "--- config4 quad0"; FTK_KLT_QUAD=0 FTK_LIB_PATH=$lib python scripts/time_variant.py lssd:fast:10000:6 lssd:fast:10000:6:lum --steps 100 || exit 1
  echo "--- config4 quad1"; FTK_KLT_QUAD=1 FTK_LIB_PATH=$lib python scripts/time_variant.py lssd:fast:10000:6 lssd:fast:10000:6:lum --steps 100 || exit 1
  echo "--- config5"; FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:25000:6 --size 1920x1080 --steps 50 || exit 1
done
