#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_f16.py tests/test_gpu_models.py tests/test_gpu_extract.py -x -q 2>&1 | tail -3
for fx in 0 1; do
  echo "== PA_ENGINE_FPX16=$fx"
  for m in pptnet patch_aug_net; do
  PA_ENGINE_FPX16=$fx timeout 600 python bench.py --model $m --mlp-dtype f16 --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config'].get('workload'), d['value'], d['ms_per_step'])"
  done
done
