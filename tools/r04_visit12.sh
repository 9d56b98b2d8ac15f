#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_f16.py tests/test_gpu_extract.py tests/test_gpu_models.py -m gpu -q 2>&1 | tail -3
for S in "" "PA_ATTN_F16_SPLIT=0"; do
env $S timeout 300 python bench.py --model pptnet --mlp-dtype f16 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pptnet f16 [$S]', round(d['value']), d['ms_per_step']); print(' '.join(f'{k}={v:.3f}' for k,v in d['kernels']['stages_ms'].items() if 'fp0' in k or 'attn' in k))
"
done
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('driver-protocol', round(d['value']), round(d['ms_per_step'],4), d['repetitions']['submaps_per_s'])
"; done
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('100-step', round(d['value']), round(d['ms_per_step'],4), d['repetitions']['submaps_per_s'], 'pcie', round(d['pcie_inclusive']['value']))
"
