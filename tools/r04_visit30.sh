#!/bin/bash
# fpx_f16.hip first contact: correctness, then stage times in the three modes
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_f16.py -x -q -k "finest_fp_level or pptnet_f16 or patch_aug_net_f16" 2>&1 | tail -15
for mode in 0 8 4; do
  echo "== PA_FPX16_LDS=$mode"
  PA_FPX16_LDS=$mode timeout 600 python bench.py --model pptnet --mlp-dtype f16 --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline'))"
  PA_FPX16_LDS=$mode timeout 600 python bench.py --model patch_aug_net --mlp-dtype f16 --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline'))"
done
