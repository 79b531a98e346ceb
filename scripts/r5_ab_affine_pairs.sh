#!/bin/bash
# A / B, same box: the non-fast affine trackers' 24 sums with one lane per sum (diag/libftk_hip_nopair.so) against a pair of lanes per sum (default)
D=feature_tracker_amd/csrc/diag
for rep in 1 2; do
for lib in $D/libftk_hip_nopair.so feature_tracker_amd/csrc/libftk_hip.so; do
  echo "=== $lib"
  FTK_LIB_PATH=$lib python scripts/time_variant.py affine:inverse:2000:6 affine:direct:2000:6 affine:inverse:300:6 affine:direct:300:6 affine:inverse:2000:10 --steps 100 || exit 1
  echo "--- real"; FTK_LIB_PATH=$lib python scripts/time_variant.py affine:inverse:300:6 affine:direct:300:6 affine:inverse:2000:6 affine:direct:2000:6 --real --steps 100 || exit 1
  FTK_LIB_PATH=$lib python scripts/time_variant.py affine:inverse:5000:6 --size 1280x720 --levels 5 --steps 100 || exit 1
done
done
