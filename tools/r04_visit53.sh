#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_models.py -q -s -k "split_fp16" 2>&1 | grep -E "relative max|max\|d|passed|failed|Error" | head
for p in 1 0; do echo "persist=$p"; PA_FPX3_PERSIST=$p timeout 300 python tools/fpx3_time.py 2>&1 | grep "split fp16"; done
