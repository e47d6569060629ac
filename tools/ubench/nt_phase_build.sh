#!/bin/bash
# Builds the harness and one small library per NT-GEMM variant into tools/ubench/bin/ (git-ignored; travels with gpurun).
#   [NT_VARIANTS="MACRO ..."] bash tools/ubench/nt_phase_build.sh     then on the box:  bash tools/ubench/nt_phase_run.sh
# nt_base.so = gemm.hip of NT_BASE_REV (default HEAD), nt_work.so = the working tree, nt_<MACRO_VALUE>.so = the working tree with -D<MACRO=VALUE>
set -e
cd "$(dirname "$0")/../.."
B=tools/ubench/bin
mkdir -p $B
rm -f $B/nt_*
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Iinclude -Ittts_amd/csrc"
build() {  # name, -D flags
  local name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -shared -o $B/nt_$name.so ttts_amd/csrc/gemm.hip ttts_amd/csrc/lib.hip &
}
# the baseline is compiled from a scratch copy outside the source tree (NT_BASE_REV: the commit to compare against, default HEAD)
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
git show ${NT_BASE_REV:-HEAD}:ttts_amd/csrc/gemm.hip > $T/gemm_base.hip
/opt/rocm/bin/hipcc $FLAGS -shared -o $B/nt_base.so $T/gemm_base.hip ttts_amd/csrc/lib.hip &
build work
for v in $NT_VARIANTS; do build ${v//=/_} -D$v; done
/opt/rocm/bin/hipcc -O2 -o $B/nt_phase tools/ubench/nt_phase.cpp -ldl &
wait
ls -la $B
