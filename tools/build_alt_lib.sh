#!/bin/bash
# Alternate library for A/B runs: the given source files taken from a git revision, everything else from the working tree's objects.
#   tools/build_alt_lib.sh <rev> file.hip [file.hip ...]   -> ttts_amd/libttts_hip_alt.so (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")/.."
REV=$1; shift
T=$(mktemp -d); trap 'rm -rf "$T"' EXIT
OBJS=""
for o in ttts_amd/csrc/build/*.o; do
  b=$(basename $o .o)
  skip=0
  for f in "$@"; do [ "$b.hip" == "$f" ] && skip=1; done
  [ $skip == 0 ] && OBJS="$OBJS $o"
done
for f in "$@"; do
  git show $REV:ttts_amd/csrc/$f > $T/$f
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Iinclude -Ittts_amd/csrc -c $T/$f -o $T/${f%.hip}.o
  OBJS="$OBJS $T/${f%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ttts_amd/libttts_hip_alt.so $OBJS
ls -la ttts_amd/libttts_hip_alt.so
