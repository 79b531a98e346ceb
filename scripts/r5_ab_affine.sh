#!/bin/bash
# A / B, same box: the non-fast affine trackers' 24 sums as quad chains on two waves (default build) against one lane per sum (diag/noaq)
D=feature_tracker_amd/csrc/diag
for lib in $D/libftk_hip_noaq.so feature_tracker_amd/csrc/libftk_hip.so; do
  echo "=== $lib"
  FTK_LIB_PATH=$lib python scripts/time_variant.py affine:inverse:300:6 affine:direct:300:6 affine:inverse:2000:6 affine:direct:2000:6 --real --steps 100 || exit 1
  FTK_LIB_PATH=$lib python scripts/time_variant.py affine:inverse:2000:6 affine:direct:2000:6 affine:inverse:2000:7 --steps 100 || exit 1
  FTK_LIB_PATH=$lib python scripts/time_variant.py affine:inverse:5000:6 --size 1280x720 --levels 5 --steps 100 || exit 1
done
