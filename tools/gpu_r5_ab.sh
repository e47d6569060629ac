#!/bin/bash
# Same-box A/Bs of round 5: (1) the VQ-VAE-GAN step of this tree against the round-4 tree (_r04/: `git archive` of 0536740, built
# here) -- did the boundary / ordering changes cost step time?  A python profile of the host side of one step goes to $O/host_profile.txt.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r5ab
mkdir -p $O
for rep in 1 2; do
  for tree in $R $R/_r04; do
    (cd $tree && timeout 300 python tools/vqvae_bench.py 32 6 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree=$tree', {k: d[k] for k in d if 'ms' in k or 'frames' in k})")
  done
done 2>&1 | tee $O/vqvae_ab.txt
for rep in 1 2; do
  (cd $R && TTTS_VQ_EXPIRE_SYNC=1 timeout 300 python tools/vqvae_bench.py 32 6 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree=here EXPIRE_SYNC=1', {k: d[k] for k in d if 'ms' in k})")
done 2>&1 | tee -a $O/vqvae_ab.txt
cat > /tmp/hostprof.py <<'PY'
import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.getcwd())
from ttts_amd.vqvae.train import SyntheticVqvaeBatches, VqvaeTrainer, get_hparams
tr = VqvaeTrainer(get_hparams())
cb = tr.net_g.quantizer.vq.layers[0]._codebook
with torch.no_grad():
    cb.inited.fill_(1); cb.embed.normal_(0, 0.3); cb.embed_avg.copy_(cb.embed * 4); cb.cluster_size.fill_(4.0)
data = next(iter(SyntheticVqvaeBatches(32, device=tr.device)))
for _ in range(2):
    tr.train_step(data)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile(); pr.enable()
tr.train_step(data); tr.train_step(data)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
PY
for tree in $R $R/_r04; do
  echo "== host profile in $tree"
  (cd $tree && python /tmp/hostprof.py 2>&1 | grep -v amdgpu.ids | head -36 | tail -28)
done 2>&1 | tee $O/host_profile.txt
