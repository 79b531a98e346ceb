#!/bin/bash
# A / B of the quad (DPP) chain in the pipelined Basic-inverse kernel, same box: round-4 form (qc0), quad chain at ring pitch 68 (pad4), default (pitch 80)
D=feature_tracker_amd/csrc/diag
for lib in $D/libftk_hip_qc0.so $D/libftk_hip_pad4.so feature_tracker_amd/csrc/libftk_hip.so; do
  echo "=== $lib"
  FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:2000:10 basic:inverse:200:5 basic:inverse:2000:6 basic:inverse:300:6 --steps 100 || exit 1
  FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:25000:6 --size 1920x1080 --steps 50 || exit 1
  FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:300:6 basic:inverse:2000:6 --real --steps 100 || exit 1
done
