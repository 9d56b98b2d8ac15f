cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-kernel-pass --no-trace --only-steps"
for i in 1 2 3; do for E in "X=1" "PA_CHAIN_NO_FPX32=1"; do echo "driver protocol $E: $(env $E $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['repetitions']['submaps_per_s'])")"; done; done
