#!/bin/bash
# Drop-in check (only where /root/reference is mounted, i.e. in the build container): compiles and
# links the reference's OWN caller programs, unchanged and in place, against this repo's headers
# and libraries.  Nothing from /root/reference is copied; the binaries land in
# feature_tracker_amd/host/build/dropin/ (git-ignored).  They need a GPU to run.
set -euo pipefail
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")/.." && pwd)
HOST=$HERE/feature_tracker_amd/host
[ -d "$REF/test" ] || { echo "reference not mounted at $REF: nothing to check"; exit 0; }
make -s -C "$HOST" -j4
mkdir -p "$HOST/build/dropin"
INC="-I$HERE/include -I$HOST/compat -I$HOST/src -I$HOST/src/optical_flow_tracker -I$HOST/src/optical_flow_tracker/basic_klt \
     -I$HOST/src/optical_flow_tracker/affine_klt -I$HOST/src/optical_flow_tracker/lssd_klt -I$HOST/src/descriptor_matcher -I$HOST/src/direct_method_tracker"
LIBS="$HOST/build/liblib_optical_flow_tracker.a $HOST/build/liblib_descriptor_matcher.a $HOST/build/liblib_direct_method_tracker.a $HOST/build/liblib_substrate.a \
      -L$HERE/feature_tracker_amd/csrc -lftk_hip -Wl,-rpath,$HERE/feature_tracker_amd/csrc -Wl,-rpath,/opt/rocm/lib -lz -lpthread"
for t in test_optical_flow test_descriptor_matcher_brief test_descriptor_matcher_superpoint test_descriptor_matcher_disk test_direct_method; do
  g++ -std=c++17 -O3 -Wall -Wno-unused-parameter $INC -o "$HOST/build/dropin/$t" "$REF/test/$t.cpp" $LIBS
  echo "linked unchanged: $t"
done
