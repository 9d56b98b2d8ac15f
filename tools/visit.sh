cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extract.py -x -q -m gpu -k "sampled_ahead or latency" 2>&1 | tail -3
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-kernel-pass --no-trace"
for i in 1 2 3; do for E in "PA_AHEAD=samplings" "PA_AHEAD=sampling"; do env $E $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver protocol $E', round(d['value']), d['repetitions']['submaps_per_s'])"; done; done
for E in "PA_AHEAD=samplings" "PA_AHEAD=sampling"; do for M in "--model pptnet --mlp-dtype f16" "--mlp-dtype f16"; do env $E $B $M 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E $M', round(d['value']), d['repetitions']['submaps_per_s'])"; done; done
