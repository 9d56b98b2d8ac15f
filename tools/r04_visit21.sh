#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py -m gpu -q -k "fps or fuzz" 2>&1 | tail -1
timeout 300 python tools/fps_time.py 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rate', round(d['value']), round(d['ms_per_step'],4), 'sa0.fps', round(d['kernels']['stages_ms']['sa0.fps'],3), d['repetitions']['submaps_per_s'])
"
done
