#!/bin/bash
# round-4 first visit: f16 PPT-Net counters (verdict item 2) + fresh 1-stream stage numbers at HEAD
mkdir -p gpurun_out
T=r04a
bash tools/pmc_pass.sh gpurun_out/${T}_ppt16_mfma.txt "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${T}_ppt16_fetch.txt "FETCH_SIZE" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${T}_ppt16_write.txt "WRITE_SIZE" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${T}_ppt16_lds.txt "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${T}_ppt16_tcp.txt "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" --model pptnet --mlp-dtype f16 --no-grouping
grep -E "chain16_kernel<2, 16, 3|sa_attn_kernel<64, 2>" gpurun_out/${T}_ppt16_*.txt | cut -c1-40,100-260
true
