D=feature_tracker_amd/csrc/diag
for rep in 1 2; do
for lib in $D/libftk_hip_extra.so feature_tracker_amd/csrc/libftk_hip.so; do
  echo "=== $lib"
  FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:2000:10 basic:inverse:2000:6 --steps 200 || exit 1
  FTK_LIB_PATH=$lib python scripts/time_variant.py basic:inverse:25000:6 --size 1920x1080 --steps 50 || exit 1
done
done
