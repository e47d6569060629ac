#!/bin/bash
# Builds tools/ubench/conv_variants and one small library per build of the convolution family into tools/ubench/bin/ (git-ignored;
# travels with gpurun):  conv_base.so = csrc/conv*.hip of CONV_BASE_REV (default HEAD), conv_work.so = the working tree,
# conv_<NAME>.so = the working tree with -D<MACRO> for every "NAME=MACRO[=VALUE]" word of CONV_VARIANTS.
#   CONV_VARIANTS="nbs4=CONV_NBS=4" bash tools/ubench/conv_variants_build.sh     then on the box:  bash tools/ubench/conv_variants_run.sh
set -e
cd "$(dirname "$0")/../.."
B=tools/ubench/bin
mkdir -p $B
rm -f $B/conv_*
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Iinclude -Ittts_amd/csrc"
SRC="conv.hip conv_mfma.hip conv_grouped.hip conv_thin.hip"
lib() {  # name, source dir, extra flags
  local name=$1 src=$2; shift 2
  ( objs=""
    for f in $SRC; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $src/$f -o $T/$name.${f%.hip}.o & done
    /opt/rocm/bin/hipcc $FLAGS -c ttts_amd/csrc/lib.hip -o $T/$name.lib.o &
    wait
    for f in $SRC; do objs="$objs $T/$name.${f%.hip}.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/conv_$name.so $objs $T/$name.lib.o ) &
}
mkdir -p $T/base
for f in $SRC; do git show ${CONV_BASE_REV:-HEAD}:ttts_amd/csrc/$f > $T/base/$f; done
lib base $T/base
lib work ttts_amd/csrc
for v in $CONV_VARIANTS; do lib ${v%%=*} ttts_amd/csrc -D${v#*=}; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $B/conv_variants tools/ubench/conv_variants.cpp -ldl &   # (it has a device kernel of its own: the fill)
wait
ls -la $B | grep conv_
