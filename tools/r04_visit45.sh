#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_chain.py -q -s -k split_fp16 2>&1 | grep -E "relative max|passed|failed|Error" | head
python tools/fpx3_time.py 2>&1 | grep -v amdgpu
