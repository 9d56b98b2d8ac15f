cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-kernel-pass --no-trace --only-steps"
show() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d['repetitions']['submaps_per_s'])"; }
for Q in 4 8 16; do for S in 4 6 8; do echo "== GPU_MAX_HW_QUEUES=$Q streams=$S"; GPU_MAX_HW_QUEUES=$Q $B --streams $S 2>/dev/null | show; done; done
for Q in 8 16; do echo "== fork probe GPU_MAX_HW_QUEUES=$Q"; for S in 4 3; do GPU_MAX_HW_QUEUES=$Q python tools/probes/fork_fps.py $S 20 2>&1 | grep -v amdgpu.ids | tail -1; done; done
