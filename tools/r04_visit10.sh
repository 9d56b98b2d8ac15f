#!/bin/bash
export TMPDIR=/tmp
A=patchaugnet_amd/csrc/ab
bash tools/ab_env.sh "PA_LIB_PATH=$A/libpa_pd12.so" "PA_LIB_PATH=$A/libpa_pd16.so" 2>&1 | grep -v "^sa0.fps" 
for L in "" $A/libpa_pd16.so; do PA_LIB_PATH=$L python tools/probes/stage_b2b.py 9 | tr '\n' ' '; echo; done
