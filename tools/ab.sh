#!/bin/bash
# A/B of env-var knobs on the GPU box: tools/ab.sh "VAR=1" -> prints stage times with and without
export TMPDIR=/tmp
python -m pytest tests/test_gpu_head.py tests/test_gpu_ops.py tests/test_gpu_chain.py tests/test_gpu_attention.py tests/test_gpu_models.py -m gpu -x -q 2>&1 | tail -3
for cfg in "$@" ""; do
  echo "=== env: [$cfg]"
  env $cfg python bench.py --no-cpu-baseline --steps 30 --warmup 6 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],3))
st=d['kernels']['stages_ms']
print(' '.join(f'{k}={v:.3f}' for k,v in st.items()))
print('roofline',d['roofline']['achieved'],d['roofline']['frac'])
"
done
