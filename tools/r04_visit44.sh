#!/bin/bash
cd /root/repo
for st in 4 8 5 4 8 10; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --streams $st --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('streams=$st', round(d['value']), d['ms_per_step'], [round(x) for x in d['repetitions']['submaps_per_s']])"
done
