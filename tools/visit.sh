cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-kernel-pass --no-trace"
for i in 1 2 3; do $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver protocol', round(d['value']), d['repetitions']['submaps_per_s'], 'plain', round(d.get('plain_graph_pipeline',{}).get('value',0)))"; done
python bench.py --no-extras --no-cpu-baseline --no-pmc --no-kernel-pass --no-trace 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('100 steps', round(d['value']), d['repetitions']['submaps_per_s'], 'plain', round(d.get('plain_graph_pipeline',{}).get('value',0)))"
for G in 4 16; do $B --group $G 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group $G', round(d['value']), d['repetitions']['submaps_per_s'])"; done
