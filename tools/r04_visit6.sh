#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r04f
for S in "" "PA_ATTN_F16_SPLIT=0" "PA_ATTN_F16_QV=1"; do
echo "== [$S]"; env $S timeout 300 python -m pytest tests/test_gpu_f16.py -m gpu -q 2>&1 | grep -E "assert np.float32|passed|failed|Error" | head -8
done
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests.log
tail -12 gpurun_out/${T}_tests.log
for S in "" "PA_ATTN_F16_SPLIT=0"; do
env $S timeout 300 python bench.py --model pptnet --mlp-dtype f16 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pptnet f16 [$S]', round(d['value']), d['ms_per_step']); print(' '.join(f'{k}={v:.3f}' for k,v in d['kernels']['stages_ms'].items()))
"
done
