cd /root/repo
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r05g_gpu_tests.log; tail -2 gpurun_out/r05g_gpu_tests.log
for i in 1 2; do python bench.py --config train --steps 50 --warmup 5 --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step ms', d['ms_per_step'], d['roofline']['frac'])"; done
python bench.py --config train > gpurun_out/r05g_bench_train.json 2>/dev/null; tail -1 gpurun_out/r05g_bench_train.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('train full', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1; done > gpurun_out/r05g_bench_driver_protocol.json
python - <<'PY'
import json
print("driver protocol", [round(json.loads(l)["value"]) for l in open("gpurun_out/r05g_bench_driver_protocol.json").read().strip().splitlines() if l.startswith("{")])
PY
