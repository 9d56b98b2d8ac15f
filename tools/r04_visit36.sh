#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4
( for seed in 2026 77; do FUZZ_SEED=$seed timeout 1500 python tests/fuzz_gpu.py 30; done ) > gpurun_out/r04_fuzz.log 2>&1
tail -40 gpurun_out/r04_fuzz.log
