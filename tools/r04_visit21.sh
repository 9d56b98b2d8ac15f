#!/bin/bash
export TMPDIR=/tmp
for C in 512 1024; do PA_FPS_NT=$C timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "fps" 2>&1 | tail -1; done
for S in 4; do for C in 256 512 1024; do
PA_FPS_NT=$C timeout 300 python bench.py --steps 40 --warmup 8 --streams $S --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $S nt $C', round(d['value']), round(d['ms_per_step'],4), 'sa0.fps', round(d['kernels']['stages_ms']['sa0.fps'],3), d['repetitions']['submaps_per_s'])
"
done; done
