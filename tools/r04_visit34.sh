#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -2
python tools/sa1_time.py 2>&1 | grep sa_mid
PA_SA_MID_PREFETCH=1 python tools/sa1_time.py 2>&1 | grep sa_mid
for cfg in "PA_CHAIN_NO_MID=1" "PA_SA_MID_PREFETCH=0" "PA_SA_MID_PREFETCH=1" "PA_CHAIN_NO_MID=1" "PA_SA_MID_PREFETCH=0"; do
  for m in patch_aug_net pptnet; do
  env $cfg timeout 600 python bench.py --model $m --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', '$m', round(d['value']), d['ms_per_step'])"
  done
done
