#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f16.py tests/test_gpu_attention.py tests/test_gpu_models.py tests/test_gpu_extract.py -m gpu -q 2>&1 | tail -3
for M in "f16" "f32"; do for S in "" "PA_ATTN_F16_FUSE=0 PA_ATTN_FUSE=0"; do
env $S timeout 300 python bench.py --model pptnet --mlp-dtype $M --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pptnet $M [$S]', round(d['value']), d['ms_per_step']); print(' '.join(f'{k}={v:.3f}' for k,v in d['kernels']['stages_ms'].items() if 'attn' in k))
"
done; done
