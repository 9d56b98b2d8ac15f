#!/bin/bash
export TMPDIR=/tmp
A=patchaugnet_amd/csrc/ab
bash tools/ab_env.sh "PA_LIB_PATH=$A/libpa_cprio1_fpx.so" "PA_LIB_PATH=$A/libpa_cprio2_fpx.so" "PA_LIB_PATH=$A/libpa_cprio1_all.so" "PA_LIB_PATH=$A/libpa_cprio2_all.so" 2>&1 | grep -E "===|^value|fp0.chain" | sed -E 's/sa0.fps.*sa0.chain=([0-9.]+).*sa1.chain=([0-9.]+).*sa2.chain=([0-9.]+).*fp2.chain=([0-9.]+).*fp1.chain=([0-9.]+).*fp0.chain=([0-9.]+).*/sa0 \1 sa1 \2 sa2 \3 fp2 \4 fp1 \5 fp0 \6/'
