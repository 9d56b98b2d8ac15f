#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_losses.py tests/test_gpu_train_full.py -m gpu -q 2>&1 | tail -3
timeout 300 python tools/emd_time.py 2>&1 | grep -v "resident"
