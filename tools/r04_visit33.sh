#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -5
timeout 300 python tools/sa1_time.py
