#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_f16.py tests/test_gpu_head.py tests/test_gpu_models.py -q 2>&1 | tail -3
for v in 0 1 0 1; do
  for m in pptnet patch_aug_net; do
  PA_ENGINE_VLAD_F16=$v timeout 600 python bench.py --model $m --mlp-dtype f16 --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('vlad_f16=$v', '$m', round(d['value']), d['ms_per_step'])"
  done
done
