#!/bin/bash
# the single-pass "TF32-class" convolution mode: accuracy tests, then the VQ-VAE-GAN step in both modes (same box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "tf32class or split_bf16_conv_accuracy" 2>&1 | tail -4
run() { env "$@" timeout 300 python tools/vqvae_bench.py 32 6 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'], 2), 'host', round(d['host_issue_ms_in_loop'], 1), {k: round(v, 4) for k, v in d['losses'].items()})"; }
for rep in 1 2; do run TTTS_CONV_PRECISION=split_bf16; run TTTS_CONV_PRECISION=tf32class; done
