#!/bin/bash
# speed-of-light probe of the finest FP level: weight fragments read from LDS instead of per-wave buffer loads (wrong results; timing only)
for i in 1 2; do
python tools/stage_scaling.py 32 2>&1 | tail -1 | tr ' ' '\n' | grep -E "fp0.chain|fp1.chain"
PA_LIB_PATH=patchaugnet_amd/csrc/ab/libpa_fakeldsw.so python tools/stage_scaling.py 32 2>&1 | tail -1 | tr ' ' '\n' | grep -E "fp0.chain|fp1.chain"
done
