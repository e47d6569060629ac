#!/bin/bash
# First GPU call of round 3: run what round 2 left unmeasured.
#  1. gated parity tests of the surplus-tile-split NT GEMM (ttts_gemm_nt_split_bf16; written without GPU access)
#  2. GPT bench with TTTS_NT_SPLIT=0 / 1 (same box, same build): per-family times in roofline.all_kernels_ms_per_step
# Output: gpurun_out/r3first/{split_tests.log,b_split0.json,b_split1.json}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3first
mkdir -p $O
TTTS_RUN_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_gpu_kernels.py -q -k "surplus_split" > $O/split_tests.log 2>&1; echo "SPLIT TESTS rc=$?"; tail -5 $O/split_tests.log
for f in 0 1; do
  TTTS_NT_SPLIT=$f timeout 120 python bench.py --no-vqvae --no-cpu-baseline --steps 150 --warmup 10 > $O/b_split$f.json 2> $O/b_split$f.err; echo "bench TTTS_NT_SPLIT=$f rc=$?"
  python - $f <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r3first/b_split%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print("TTTS_NT_SPLIT=%s ms/step %s" % (sys.argv[1], d["ms_per_step"]), d["roofline"]["all_kernels_ms_per_step"])
except Exception as e:
    print("unreadable:", e)
PY
done
# 3. where the attention kernels wait (they are latency-bound: 16 % MFMA-busy, 37 % VALU-busy, 30 % of wave cycles waiting):
#    two counter passes over tools/gpt_step_once.py; read the attn_* blocks of the outputs
bash tools/gpt_pmc.sh r03_pmc_attn_wait1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
bash tools/gpt_pmc.sh r03_pmc_attn_wait2 SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
grep -A9 "attn_bwd_dkdv" gpurun_out/pmc/r03_pmc_attn_wait1.txt | head -10
grep -A9 "attn_bwd_dkdv" gpurun_out/pmc/r03_pmc_attn_wait2.txt | head -10
# 4. placement of a 3-per-CU kernel (attention dq / forward): does block b + 512 join blocks b and b + 256?
hipcc --offload-arch=gfx950 -O2 tools/exp/placement_probe.hip -o /tmp/pp && /tmp/pp 640 > $O/placement_640.txt && tail -1 $O/placement_640.txt
# 5. dK / dV kernel with zero-fill and statistics scaling at the LDS store instead of at the global load (-DTTTS_DKDV_LATE=1:
#    its ISA issues the tile prefetch back to back and waits after the MFMAs; the default waits four times in front of them).
#    Builds the variant library on the box, runs the attention + GPT parity tests and the bench with it, then restores.
cp ttts_amd/libttts_hip.so /tmp/lib_default.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-variable -Iinclude"
hipcc $FLAGS -DTTTS_DKDV_LATE=1 -c ttts_amd/csrc/attn.hip -o /tmp/attn_late.o 2> $O/late_build.err \
 && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_late.so $(ls ttts_amd/csrc/build/*.o | grep -v "build/attn.o") /tmp/attn_late.o \
 && cp /tmp/lib_late.so ttts_amd/libttts_hip.so && touch ttts_amd/csrc/build/*.o && sleep 0.1 && touch ttts_amd/libttts_hip.so \
 && { timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gpt.py -q -k "attn or attention or train_steps or tiny or full_config or dropout or grouped" > $O/late_tests.log 2>&1; echo "LATE TESTS rc=$?"; tail -3 $O/late_tests.log; \
      timeout 120 python bench.py --no-vqvae --no-cpu-baseline --steps 150 --warmup 10 > $O/b_late.json 2> $O/b_late.err; \
      python -c "import json; d=json.loads(open('$O/b_late.json').read().strip().splitlines()[-1]); print('DKDV_LATE ms/step', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"; }
#    5b. the same plus the key-block pairing order (-DTTTS_DKDV_PAIR=1: every CU gets 24 tile iterations instead of 30/26/22/18)
hipcc $FLAGS -DTTTS_DKDV_LATE=1 -DTTTS_DKDV_PAIR=1 -c ttts_amd/csrc/attn.hip -o /tmp/attn_lp.o 2> $O/lp_build.err \
 && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_lp.so $(ls ttts_amd/csrc/build/*.o | grep -v "build/attn.o") /tmp/attn_lp.o \
 && cp /tmp/lib_lp.so ttts_amd/libttts_hip.so && touch ttts_amd/csrc/build/*.o && sleep 0.1 && touch ttts_amd/libttts_hip.so \
 && { timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gpt.py -q -k "attn or attention or train_steps or tiny or full_config or dropout" > $O/lp_tests.log 2>&1; echo "LATE+PAIR TESTS rc=$?"; tail -3 $O/lp_tests.log; \
      timeout 120 python bench.py --no-vqvae --no-cpu-baseline --steps 150 --warmup 10 > $O/b_lp.json 2> $O/b_lp.err; \
      python -c "import json; d=json.loads(open('$O/b_lp.json').read().strip().splitlines()[-1]); print('DKDV_LATE+PAIR ms/step', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"; }
cp /tmp/lib_default.so ttts_amd/libttts_hip.so; touch ttts_amd/csrc/build/*.o; sleep 0.1; touch ttts_amd/libttts_hip.so
# 6. fused single-pass conv weight gradient with the bounds masks at the LDS store (-DTTTS_WGRAD_LATE=1): the default's
#    `cond ? loaded : 0` right behind each load makes its register prefetch synchronous (ISA).  VQ-VAE tests + step time.
hipcc $FLAGS -DTTTS_WGRAD_LATE=1 -c ttts_amd/csrc/conv_mfma.hip -o /tmp/conv_late.o 2> $O/wgrad_late_build.err \
 && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_wlate.so $(ls ttts_amd/csrc/build/*.o | grep -v "build/conv_mfma.o") /tmp/conv_late.o \
 && cp /tmp/lib_wlate.so ttts_amd/libttts_hip.so && touch ttts_amd/csrc/build/*.o && sleep 0.1 && touch ttts_amd/libttts_hip.so \
 && { timeout 300 python -m pytest tests/test_gpu_vqvae.py -q -k "wgrad or conv or step" > $O/wlate_tests.log 2>&1; echo "WGRAD_LATE TESTS rc=$?"; tail -3 $O/wlate_tests.log; \
      timeout 200 python tools/vqvae_bench.py 32 6 1 > $O/vq_wlate.log 2>&1; tail -3 $O/vq_wlate.log; }
cp /tmp/lib_default.so ttts_amd/libttts_hip.so; touch ttts_amd/csrc/build/*.o; sleep 0.1; touch ttts_amd/libttts_hip.so
timeout 200 python tools/vqvae_bench.py 32 6 1 > $O/vq_default.log 2>&1; tail -3 $O/vq_default.log
# 7. nearest-code search with unguarded (clamped-index) code-row loads + a branch-free full-group path (-DTTTS_VQ_UNGUARDED=1):
#    the default's guarded prefetch is waited for before the MFMAs it should overlap (ISA).  Indices must stay bit-exact.
hipcc $FLAGS -DTTTS_VQ_UNGUARDED=1 -c ttts_amd/csrc/vq.hip -o /tmp/vq_un.o 2> $O/vq_un_build.err \
 && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_vqun.so $(ls ttts_amd/csrc/build/*.o | grep -v "build/vq.o") /tmp/vq_un.o \
 && cp /tmp/lib_vqun.so ttts_amd/libttts_hip.so && touch ttts_amd/csrc/build/*.o && sleep 0.1 && touch ttts_amd/libttts_hip.so \
 && { timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vqvae.py -q -k "vq" > $O/vqun_tests.log 2>&1; echo "VQ_UNGUARDED TESTS rc=$?"; tail -3 $O/vqun_tests.log; \
      timeout 100 python tools/hbm_bench.py > $O/hbm_vqun.log 2>&1; grep -i "vq_nearest" $O/hbm_vqun.log | head -3; }
cp /tmp/lib_default.so ttts_amd/libttts_hip.so; touch ttts_amd/csrc/build/*.o; sleep 0.1; touch ttts_amd/libttts_hip.so
timeout 100 python tools/hbm_bench.py > $O/hbm_default.log 2>&1; grep -i "vq_nearest" $O/hbm_default.log | head -3
