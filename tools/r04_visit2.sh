#!/bin/bash
# round-4 visit: full GPU suite on the refactored build, A/B bench HEAD library vs working tree, f16 PPT-Net counters
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r04b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests.log
tail -15 gpurun_out/${T}_tests.log
bash tools/ab_env.sh "PA_LIB_PATH=patchaugnet_amd/csrc/ab/libpa_head.so" 2>&1 | tee gpurun_out/${T}_ab.txt
bash tools/pmc_pass.sh gpurun_out/${T}_ppt16_mfma.txt "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${T}_ppt16_fetch.txt "FETCH_SIZE" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${T}_ppt16_write.txt "WRITE_SIZE" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${T}_ppt16_lds.txt "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${T}_ppt16_tcp.txt "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" --model pptnet --mlp-dtype f16 --no-grouping
grep -E "chain16_kernel<2, 16, 3|sa_attn_kernel<64, 2>" gpurun_out/${T}_ppt16_*.txt | cut -c1-40,100-260
