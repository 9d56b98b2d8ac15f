#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_f16.py -x -q -k finest 2>&1 | tail -2
for st in 0 1 2 3; do echo "== stagger $st"; PA_FPX16_STAGGER=$st timeout 300 python tools/fpx16_time.py 2>&1 | grep "fp16 table"; done
