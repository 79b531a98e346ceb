#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d); mkdir -p $T/example $T/build
ln -s $ROOT/tests/data/optical_flow $T/example/optical_flow
cd $T/build
for i in 1 2; do
FTK_TRACE=1 $ROOT/feature_tracker_amd/host/build/dropin/test_descriptor_matcher_brief 2>&1 | grep -a "ftk trace\|cost time\|Detect\|Compute" | sed 's/\x1b\[[0-9;]*m//g'
echo ======
done
