#!/bin/bash
# Round 3, GPU call C: the two remaining round-2 candidates (conv weight-gradient LATE, VQ nearest UNGUARDED) vs the default build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3c
mkdir -p $O
cp ttts_amd/libttts_hip.so /tmp/lib_default.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-variable -Iinclude"
swap() { cp $1 ttts_amd/libttts_hip.so; touch ttts_amd/csrc/build/*.o; sleep 0.1; touch ttts_amd/libttts_hip.so; }
timeout 200 python tools/vqvae_bench.py 32 6 1 > $O/vq_default.log 2>&1; tail -2 $O/vq_default.log
timeout 100 python tools/hbm_bench.py > $O/hbm_default.log 2>&1; grep -i "vq_nearest" $O/hbm_default.log | head -3
hipcc $FLAGS -DTTTS_WGRAD_LATE=1 -c ttts_amd/csrc/conv_mfma.hip -o /tmp/conv_late.o 2> $O/wgrad_late_build.err \
 && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_wlate.so $(ls ttts_amd/csrc/build/*.o | grep -v "build/conv_mfma.o") /tmp/conv_late.o && swap /tmp/lib_wlate.so \
 && { timeout 300 python -m pytest tests/test_gpu_vqvae.py -q -k "wgrad or conv or step" -p no:cacheprovider > $O/wlate_tests.log 2>&1; echo "WGRAD_LATE TESTS rc=$?"; tail -2 $O/wlate_tests.log; \
      timeout 200 python tools/vqvae_bench.py 32 6 1 > $O/vq_wlate.log 2>&1; tail -2 $O/vq_wlate.log; }
swap /tmp/lib_default.so
hipcc $FLAGS -DTTTS_VQ_UNGUARDED=1 -c ttts_amd/csrc/vq.hip -o /tmp/vq_un.o 2> $O/vq_un_build.err \
 && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_vqun.so $(ls ttts_amd/csrc/build/*.o | grep -v "build/vq.o") /tmp/vq_un.o && swap /tmp/lib_vqun.so \
 && { timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vqvae.py -q -k "vq" -p no:cacheprovider > $O/vqun_tests.log 2>&1; echo "VQ_UNGUARDED TESTS rc=$?"; tail -2 $O/vqun_tests.log; \
      timeout 100 python tools/hbm_bench.py > $O/hbm_vqun.log 2>&1; grep -i "vq_nearest" $O/hbm_vqun.log | head -3; }
swap /tmp/lib_default.so
