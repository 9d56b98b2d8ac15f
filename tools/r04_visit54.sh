#!/bin/bash
cd /root/repo
for dt in f32 f32x3 f32 f32x3; do
  timeout 600 python bench.py --model pptnet --mlp-dtype $dt --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('pptnet $dt', round(d['value']), d['ms_per_step'])"
done
