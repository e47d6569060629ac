#!/bin/bash
# round 6, first call: the RCCL world-size-1 tests (TTTS_DP_FORCE=1) -- the nccl backend executing for the first time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6a; mkdir -p $O
export NCCL_DEBUG=WARN
timeout 300 env TTTS_DP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python tools/dp_world1_nccl.py gpt > $O/w1_gpt.log 2>&1; echo "gpt rc $?"; tail -5 $O/w1_gpt.log
timeout 600 env TTTS_DP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 python tools/dp_world1_nccl.py vqvae > $O/w1_vqvae.log 2>&1; echo "vqvae rc $?"; tail -5 $O/w1_vqvae.log
timeout 900 python -m pytest tests/test_gpu_gpt.py -q -p no:cacheprovider -x -k "world1 or rccl" 2>&1 | tail -15
