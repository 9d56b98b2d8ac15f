#!/bin/bash
cd /root/repo
for dt in f32 f32x3 f32 f32x3; do
  timeout 600 python bench.py --mlp-dtype $dt --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$dt', round(d['value']), d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_extract.py -q 2>&1 | tail -2
