#!/bin/bash
# Round-4 counter passes: GPT step SQ / FETCH / WRITE (per kernel), VQ-VAE-GAN step kernel stats + conv-family traffic.
# Every rocprofv3 call is wrapped in `timeout`.  Output: gpurun_out/pmc/r04_*.txt, gpurun_out/prof/r04_vqvae_kernel_stats.csv,
# gpurun_out/vqvae_pmc_traffic.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
bash tools/gpt_pmc.sh r04_pmc_gpt_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE > /dev/null
bash tools/gpt_pmc.sh r04_pmc_gpt_fetch FETCH_SIZE > /dev/null
bash tools/gpt_pmc.sh r04_pmc_gpt_write WRITE_SIZE > /dev/null
grep -A9 "dh64\|wreg" gpurun_out/pmc/r04_pmc_gpt_sq.txt | head -60
mkdir -p gpurun_out/prof
rm -rf /tmp/vprof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vprof -o v -- python $R/tools/vqvae_bench.py 32 3 1 > $R/gpurun_out/vqvae_prof_stdout.txt 2>&1)
for f in $(find /tmp/vprof -name "*kernel_stats*.csv"); do cp $f gpurun_out/prof/r04_vqvae_kernel_stats.csv; done
head -12 gpurun_out/prof/r04_vqvae_kernel_stats.csv | cut -c1-150
bash tools/vqvae_pmc.sh 2 2>&1 | tail -20
