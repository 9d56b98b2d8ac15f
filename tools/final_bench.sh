export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/r02v_bench.json 2> gpurun_out/r02v_bench.err
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null; done > gpurun_out/r02v_bench_driver_protocol.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02v_bench.json").read().strip().splitlines()[-1]); print(round(d["value"]), d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
print([round(json.loads(l)["value"]) for l in open("gpurun_out/r02v_bench_driver_protocol.json").read().strip().splitlines() if l.startswith("{")])
PY
