#!/bin/bash
# GPU box: runs the reference's own test programs (compiled unchanged by scripts/check_dropin.sh) on the reference's example
# images and prints the times THEY report; FTK_HOST_PYRAMID=1 shows the same with the host-side pyramid loop for comparison.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d); mkdir -p $T/example $T/build
ln -s $ROOT/tests/data/optical_flow $T/example/optical_flow
ln -s $ROOT/tests/data/direct_method $T/example/direct_method
cd $T/build
for mode in device host; do
  [ $mode = host ] && export FTK_HOST_PYRAMID=1 || unset FTK_HOST_PYRAMID
  echo "=== pyramids built on the $mode"
  for p in test_optical_flow test_descriptor_matcher_brief test_descriptor_matcher_superpoint test_direct_method; do
    echo "--- $p"; $ROOT/feature_tracker_amd/host/build/dropin/$p 2>&1 | grep -a -i "cost time\|track\|match" | sed 's/\x1b\[[0-9;]*m//g' | head -8
  done
  # the tracker program's spans are a few hundred microseconds of host + device work: three more runs show the spread
  for i in 2 3 4; do
    echo "--- test_optical_flow, run $i"; $ROOT/feature_tracker_amd/host/build/dropin/test_optical_flow 2>&1 | grep -a -i "cost time" | sed 's/\x1b\[[0-9;]*m//g'
  done
done
echo "=== bench_cli n=300 / n=2000"
$ROOT/feature_tracker_amd/host/build/bench_cli $ROOT/tests/data/optical_flow/ref_image.png $ROOT/tests/data/optical_flow/cur_image.png 300 4 6 200 2>&1 | tail -12
$ROOT/feature_tracker_amd/host/build/bench_cli $ROOT/tests/data/optical_flow/ref_image.png $ROOT/tests/data/optical_flow/cur_image.png 2000 4 6 200 2>&1 | tail -12
