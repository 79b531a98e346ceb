#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d); mkdir -p $T/example $T/build
ln -s $ROOT/tests/data/optical_flow $T/example/optical_flow
cd $T/build
FTK_TRACE=1 $ROOT/feature_tracker_amd/host/build/dropin/test_optical_flow 2>&1 | grep -a "ftk trace\|cost time" | sed 's/\x1b\[[0-9;]*m//g' | head -40
