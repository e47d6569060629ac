#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
src = open("tests/test_gpu_vqvae.py").read()
w = src[src.index('VQ_WORKER = r"""') + len('VQ_WORKER = r"""'):]
w = w[:w.index('"""')]
open("/tmp/vq_worker.py", "w").write("ROOT = %r\n" % "/root/repo" + w)
PY
export TTTS_SHARE_GPU=1 MASTER_ADDR=127.0.0.1 PYTHONFAULTHANDLER=1
for cfg in "TTTS_BRANCH_STREAMS=0 TTTS_D_STREAMS=0" "TTTS_BRANCH_STREAMS=0 TTTS_D_STREAMS=3" "TTTS_BRANCH_STREAMS=3 TTTS_D_STREAMS=0" "TTTS_BRANCH_STREAMS=0 TTTS_D_STREAMS=0 TTTS_WGRAD_ARENA=0 TTTS_WSPLIT_CACHE=0"; do
  echo "== $cfg"
  env $cfg timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29931 /tmp/vq_worker.py 2>&1 | grep -v "^$" | grep "rank.-ok\|Fatal Python\|Error\|capture" | cut -c1-200 | head -8
done
