#!/bin/bash
export TMPDIR=/tmp
A=patchaugnet_amd/csrc/ab
timeout 300 python -m pytest tests/test_gpu_head.py tests/test_gpu_models.py -m gpu -q 2>&1 | tail -1
for L in "" $A/libpa_vlad_none.so $A/libpa_vlad_prio.so $A/libpa_vlad_il.so $A/libpa_vlad_dbg.so; do echo "== $L"; if [ -n "$L" ]; then export PA_LIB_PATH=$L; fi; timeout 300 python tools/vlad_time.py 2>&1 | grep -E "k=64|cycles"; done
