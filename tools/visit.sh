cd /root/repo
for i in 1 2 3; do for M in 1 2 4; do
PA_SA_TINY_GRID_MULT=$M python bench.py --steps 60 --reps 3 --no-cpu-baseline --no-kernel-pass --no-pmc --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('grid mult $M', round(d['value']), d['ms_per_step'])"
done; done
