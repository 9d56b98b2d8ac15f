#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r04d
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_f16.py tests/test_gpu_attention.py tests/test_pointnet_vlad.py tests/test_gpu_ops.py tests/test_gpu_train_ops.py -m gpu -q 2>&1 | tail -25
timeout 200 python tools/chain_phases.py fp0 fp1 fp2 sa0 sa1 sa2 2>&1 | tail -14
for S in "" "PA_ATTN_F16_SPLIT=0"; do
env $S timeout 300 python bench.py --model pptnet --mlp-dtype f16 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pptnet f16 [$S]', round(d['value']), d['ms_per_step']); print(' '.join(f'{k}={v:.3f}' for k,v in d['kernels']['stages_ms'].items()))
"
done
timeout 300 python bench.py --model pptnet --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pptnet f32', round(d['value']), d['ms_per_step']); print(' '.join(f'{k}={v:.3f}' for k,v in d['kernels']['stages_ms'].items()))
"
