#!/bin/bash
# A / B of the quad (DPP) chains, every tracker variant, same box: round-4 chains (diag/libftk_hip_${BASE:-r4chains}.so) against the default build.
# Synthetic scene (2 000 features, 13 x 13) and the reference's example pair (300 / 2 000 features).
D=feature_tracker_amd/csrc/diag
V="basic:inverse basic:direct basic:fast affine:inverse affine:direct affine:fast lssd:inverse lssd:direct lssd:fast"
for lib in $D/libftk_hip_${BASE:-r4chains}.so feature_tracker_amd/csrc/libftk_hip.so; do
  echo "=== $lib"
  S=""; R3=""; R2=""
  for v in $V; do S="$S $v:2000:6"; R3="$R3 $v:300:6"; R2="$R2 $v:2000:6"; done
  FTK_LIB_PATH=$lib python scripts/time_variant.py $S lssd:fast:2000:6:lum --steps 100 || exit 1
  echo "--- real 300"; FTK_LIB_PATH=$lib python scripts/time_variant.py $R3 --real --steps 100 || exit 1
  echo "--- real 2000"; FTK_LIB_PATH=$lib python scripts/time_variant.py $R2 --real --steps 100 || exit 1
  echo "--- configs"; FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:2000:10 basic:inverse:200:5 lssd:fast:10000:6 lssd:fast:10000:6:lum --steps 100 || exit 1
  FTK_LIB_PATH=$lib python scripts/time_variant.py affine:inverse:5000:6 --size 1280x720 --levels 5 --steps 100 || exit 1
  FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:25000:6 --size 1920x1080 --steps 50 || exit 1
done
