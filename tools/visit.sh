cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_properties.py tests/test_gpu_f16.py tests/test_gpu_fuzz.py -x -q -k "fps or furthest or sampling or beside or fuzz" 2>&1 | tail -2
for i in 1 2 3; do python bench.py --steps 40 --reps 3 --no-pmc --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['frac'], d['roofline_latency']['ms_per_launch'], d['roofline_latency']['frac'])"; done
