#!/bin/bash
# SQ counter passes over tools/tgemm_target.py (the fp0-shape training GEMM): tools/pmc_tgemm.sh
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_tg$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_tg$i -o sq -- python $R/tools/tgemm_target.py > $R/gpurun_out/pmc_tg$i.log 2>&1
  echo "pass $i rc=$?"
  python $R/tools/pmc_summary.py $(ls $R/gpurun_out/pmc_tg$i/*results.db | head -1) > $R/gpurun_out/pmc_tg$i.txt 2>&1; rm -rf $R/gpurun_out/pmc_tg$i
done
