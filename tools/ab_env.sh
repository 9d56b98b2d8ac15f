#!/bin/bash
# In-run A/B of environment knobs: bash tools/ab_env.sh "VAR=1" "VAR2=x" ...  (each config and the default, interleaved twice)
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "" "$@"; do
  echo "=== rep $rep env: [$cfg]"
  env $cfg python bench.py --no-cpu-baseline --no-pmc --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['kernels']['stages_ms']
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],4), 'fp0.chain',round(st.get('fp0.chain',0),4), 'total1stream', round(st['total'],3), 'roofline', round(d['roofline']['frac'],4))
print(' '.join(f'{k}={v:.3f}' for k,v in st.items()))
"
done
done
