#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r04c
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_models.py tests/test_pointnet_vlad.py tests/test_gpu_ops.py tests/test_gpu_train_ops.py -m gpu -x -q 2>&1 | tail -15
A=patchaugnet_amd/csrc/ab
bash tools/ab_env.sh "PA_CHAIN_POOLED_NC4=1" "PA_LIB_PATH=$A/libpa_fp_minb2.so" "PA_LIB_PATH=$A/libpa_sa_minb3.so" "PA_LIB_PATH=$A/libpa_plain_minb3.so" "PA_LIB_PATH=$A/libpa_all_minb.so" 2>&1 | tee gpurun_out/${T}_ab.txt
