#!/bin/bash
# Builds tools/ubench/attn_variants and one small library per build of the attention kernels into tools/ubench/bin/
# (git-ignored; travels with gpurun):  attn_base.so = csrc/attn*.hip of ATTN_BASE_REV (default HEAD) with the product flags,
# attn_work.so = the working tree, attn_<name>.so for every "name:extra flags for attn_dh64.hip" entry of ATTN_VARIANTS
# (separated by ';'; the entry "name:-" drops the product's per-file flags instead of adding to them).
#   ATTN_VARIANTS="slp:-;o2:-O2" bash tools/ubench/attn_variants_build.sh     then on the box:  bash tools/ubench/attn_variants_run.sh
set -e
cd "$(dirname "$0")/../.."
B=tools/ubench/bin
mkdir -p $B
rm -f $B/attn_*
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Iinclude -Ittts_amd/csrc"
DH64="-fno-honor-nans -fno-slp-vectorize"      # ttts_amd/lib.py EXTRA_FLAGS["attn_dh64.hip"]
lib() {  # name, source dir, flags for attn_dh64.hip
  local name=$1 src=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS -c $src/attn.hip -o $T/$name.attn.o &&
    /opt/rocm/bin/hipcc $FLAGS "$@" -c $src/attn_dh64.hip -o $T/$name.dh64.o &&
    /opt/rocm/bin/hipcc $FLAGS -c ttts_amd/csrc/lib.hip -o $T/$name.lib.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/attn_$name.so $T/$name.attn.o $T/$name.dh64.o $T/$name.lib.o ) &
}
mkdir -p $T/base
for f in attn.hip attn_dh64.hip; do git show ${ATTN_BASE_REV:-HEAD}:ttts_amd/csrc/$f > $T/base/$f; done
lib base $T/base $DH64
lib work ttts_amd/csrc $DH64
IFS=';' read -ra VS <<< "$ATTN_VARIANTS"
for v in "${VS[@]}"; do
  [ -z "$v" ] && continue
  name=${v%%:*}; fl=${v#*:}
  if [ "$fl" = "-" ]; then lib $name ttts_amd/csrc; else lib $name ttts_amd/csrc $DH64 $fl; fi
done
/opt/rocm/bin/hipcc -O2 -o $B/attn_variants tools/ubench/attn_variants.cpp -ldl &
wait
ls -la $B | grep attn_
