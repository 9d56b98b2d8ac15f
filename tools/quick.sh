#!/bin/bash
# Quick GPU visit: tools/quick.sh TAG "pytest args" ["ENV=1 ENV2=.." ...]  -> tests, then one bench line (+ stage times) per env set, always ending with the default env
TAG=${1:-q}; TESTS=${2:-}; shift 2
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -15; fi
for cfg in "$@" ""; do
  echo "=== env: [$cfg]"
  env $cfg timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-pmc 2> gpurun_out/${TAG}_bench.err | tee gpurun_out/${TAG}_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],4))
st=d['kernels']['stages_ms']
print(' '.join(f'{k}={v:.3f}' for k,v in st.items()))
print('roofline',round(d['roofline']['achieved'],2),round(d['roofline']['frac'],4))
" || tail -5 gpurun_out/${TAG}_bench.err
done
