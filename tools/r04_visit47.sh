#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_models.py -q -s -k split_fp16 2>&1 | grep -E "max\|d|passed|failed|Error|assert" | head
timeout 300 python bench.py --mlp-dtype f32x3 --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['dtype'][:60]); print(d.get('roofline')); print({k: round(v*1e3,1) for k,v in d['kernels']['stages_ms'].items()})"
