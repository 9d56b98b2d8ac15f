cd /root/repo
for S in 4 8 4 8 7 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 --streams $S --no-cpu-baseline --no-kernel-pass --no-pmc --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('20 steps, streams $S', round(d['value']), d['ms_per_step'], [round(v) for v in d['repetitions']['submaps_per_s']])"
done
