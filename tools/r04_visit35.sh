#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_train_glue.py tests/test_gpu_train_ops.py tests/test_gpu_train_full.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --config train --steps 40 --warmup 10 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'])"
