cd $GRAFT_REPO_ROOT
timeout 60 python -m pytest tests/test_gpu_gpt.py -q -k "grouped" 2>&1 | tail -2
bash tools/gpt_pmc.sh r02_pmc_gpt_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
bash tools/gpt_pmc.sh r02_pmc_gpt_fetch FETCH_SIZE
bash tools/gpt_pmc.sh r02_pmc_gpt_write WRITE_SIZE
grep -A9 "grouped" gpurun_out/pmc/r02_pmc_gpt_sq.txt | head -12
grep -A1 "grouped" gpurun_out/pmc/r02_pmc_gpt_fetch.txt gpurun_out/pmc/r02_pmc_gpt_write.txt
