#!/bin/bash
# Experiment library: one source file of the working tree compiled with extra -D flags, everything else from the in-tree objects.
#   tools/build_exp_lib.sh <out.so> file.hip -DFLAG [...]      (build AFTER the in-tree library: lib.build() relinks what is newer)
set -e
cd "$(dirname "$0")/.."
OUT=$1; F=$2; shift; shift
T=$(mktemp -d); trap 'rm -rf "$T"' EXIT
OBJS=""
for o in ttts_amd/csrc/build/*.o; do
  [ "$(basename $o .o).hip" == "$F" ] || OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Iinclude -Ittts_amd/csrc "$@" -c ttts_amd/csrc/$F -o $T/x.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS $T/x.o
ls -la $OUT
