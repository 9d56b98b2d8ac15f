#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_f16.py -q -k netvlad 2>&1 | tail -2
for r in 256 512 256 512; do
  for m in pptnet patch_aug_net; do
  PA_VLAD16_ROWS=$r timeout 600 python bench.py --model $m --mlp-dtype f16 --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('rows=$r', '$m', round(d['value']), d['ms_per_step'])"
  done
done
