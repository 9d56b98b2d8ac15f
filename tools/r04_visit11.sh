#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_f16.py tests/test_gpu_chain.py -m gpu -q 2>&1 | tail -3
for S in "" "PA_CHAIN16_RT2=1"; do
env $S timeout 300 python bench.py --model pptnet --mlp-dtype f16 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pptnet f16 [$S]', round(d['value']), d['ms_per_step']); print(' '.join(f'{k}={v:.3f}' for k,v in d['kernels']['stages_ms'].items() if 'fp0' in k or 'attn' in k))
"
done
