#!/bin/bash
# A / B, same box: the spread direct-method kernel's consumer with one lane per sum (diag/sq0) against quad chains on two waves (default)
D=feature_tracker_amd/csrc/diag
for rep in 1 2; do
for lib in $D/libftk_hip_sq0.so feature_tracker_amd/csrc/libftk_hip.so; do
  echo "=== $lib"
  FTK_LIB_PATH=$lib python scripts/direct_batch_time.py 1 2 4 6 || exit 1
  FTK_LIB_PATH=$lib python scripts/bench_configs.py --only direct --quick 2>/dev/null | grep -a "direct_method" | cut -c1-300
done
done
