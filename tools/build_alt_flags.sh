#!/bin/bash
# Alternate library for A/B runs: the given working-tree source file rebuilt with extra compiler flags, everything else from the
# working tree's objects.   tools/build_alt_flags.sh file.hip -DNAME[=V] ...   -> ttts_amd/libttts_hip_alt.so
set -e
cd "$(dirname "$0")/.."
F=$1; shift
T=$(mktemp -d); trap 'rm -rf "$T"' EXIT
OBJS=""
for o in ttts_amd/csrc/build/*.o; do
  [ "$(basename $o .o).hip" == "$F" ] || OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC "$@" -c ttts_amd/csrc/$F -o $T/alt.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ttts_amd/libttts_hip_alt.so $OBJS $T/alt.o
ls -la ttts_amd/libttts_hip_alt.so
