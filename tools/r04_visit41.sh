#!/bin/bash
cd /root/repo
for st in 4 6 8; do
  for m in "patch_aug_net f16" "pptnet f16" "patch_aug_net f32"; do
  set -- $m
  timeout 600 python bench.py --model $1 --mlp-dtype $2 --streams $st --steps 60 --warmup 12 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('streams=$st', '$1 $2', round(d['value']), d['ms_per_step'])"
  done
done
