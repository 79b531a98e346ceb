#!/bin/bash
# A / B, same box: the pipelined Basic kernel's consumer chunk by chunk (diag/libftk_hip_${BASE}.so) against the default build
D=feature_tracker_amd/csrc/diag
for rep in 1 2; do
for lib in $D/libftk_hip_${BASE:-noahead}.so feature_tracker_amd/csrc/libftk_hip.so; do
  echo "=== $lib"
  FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:2000:10 basic:inverse:200:5 basic:inverse:2000:6 basic:direct:2000:6 basic:inverse:300:6 basic:direct:300:6 --steps 200 || exit 1
  echo "--- real"; FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:300:6 basic:direct:300:6 basic:inverse:2000:6 basic:direct:2000:6 --real --steps 200 || exit 1
  FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:25000:6 --size 1920x1080 --steps 50 || exit 1
done
done
