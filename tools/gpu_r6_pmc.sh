#!/bin/bash
# Round-6 counter / profile passes of the FINAL code -> gpurun_out/r6p/ (copied to profiles/r06_* afterwards):
#   GPT step SQ / FETCH / WRITE per kernel (attention MFMA-busy of this round), GPT bench kernel stats, VQ-VAE-GAN kernel stats
#   (both precisions, one stream) + conv-family HBM traffic, diffusion kernel stats (both precisions) + whole-step HBM traffic.
# Every rocprofv3 call: --kernel-trace (+ --stats | --pmc) only, wrapped in `timeout`.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r6p
mkdir -p $O
[ -n "$SKIP_GPT_PMC" ] || bash tools/gpt_pmc.sh r06_pmc_gpt_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE > /dev/null
[ -n "$SKIP_GPT_PMC" ] || bash tools/gpt_pmc.sh r06_pmc_gpt_fetch FETCH_SIZE > /dev/null
[ -n "$SKIP_GPT_PMC" ] || bash tools/gpt_pmc.sh r06_pmc_gpt_write WRITE_SIZE > /dev/null
cp gpurun_out/pmc/r06_pmc_gpt_*.txt $O/
grep -A9 "dh64" $O/r06_pmc_gpt_sq.txt | head -50
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gprof -o gpt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vqvae --no-diffusion > $O/prof_bench.json 2> $O/prof.err); echo "GPT PROF rc=$?"
find /tmp/gprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r06_bench_kernel_stats.csv
for prec in split_bf16 tf32class; do
  rm -rf /tmp/vprof
  (cd /tmp && TTTS_CONV_PRECISION=$prec TTTS_BRANCH_STREAMS=0 TTTS_D_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vprof -o v -- python $R/tools/vqvae_bench.py 32 3 1 > $O/vqvae_prof_$prec.txt 2>&1)
  f=$(find /tmp/vprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_vqvae_kernel_stats_$prec.csv
  tail -1 $O/vqvae_prof_$prec.txt | cut -c1-200
done
TTTS_BRANCH_STREAMS=0 TTTS_D_STREAMS=0 bash tools/vqvae_pmc.sh 1 > $O/vqvae_pmc.txt 2>&1; cp gpurun_out/vqvae_pmc_traffic.json $O/r06_vqvae_pmc_traffic.json; tail -12 $O/vqvae_pmc.txt | cut -c1-200
for mode in f32 fp8; do
  rm -rf /tmp/dprof_$mode
  (cd /tmp && TTTS_DIFFUSION_PRECISION=$mode DFB_STEPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof_$mode -o d -- python $R/tools/diffusion_bench.py > $O/diff_$mode.txt 2>&1)
  tail -1 $O/diff_$mode.txt | cut -c1-300
  f=$(find /tmp/dprof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_diffusion_${mode}_kernel_stats.csv
done
bash tools/diffusion_pmc.sh 3 > $O/diffusion_pmc.txt 2>&1; cp gpurun_out/diffusion_pmc_traffic.json $O/r06_diffusion_pmc_traffic.json; tail -8 $O/diffusion_pmc.txt | cut -c1-200
