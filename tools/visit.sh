cd /root/repo
bash tools/pmc_pass.sh gpurun_out/fps_pmc_a.txt "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
bash tools/pmc_pass.sh gpurun_out/fps_pmc_b.txt "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
grep -h "fps_reg_kernel<256, 16" gpurun_out/fps_pmc_a.txt gpurun_out/fps_pmc_b.txt | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_extract.py -q -k "matches_plain_forward" 2>&1 | tail -2
