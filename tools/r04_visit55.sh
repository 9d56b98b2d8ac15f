#!/bin/bash
cd /root/repo
FUZZ_SEED=4242 timeout 3300 python tests/fuzz_gpu.py 100 > gpurun_out/r04_fuzz_long.log 2>&1
tail -25 gpurun_out/r04_fuzz_long.log
