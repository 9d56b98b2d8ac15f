cd /root/repo; export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_drv.json 2> gpurun_out/r05_bench_drv.err; echo "rc=$?"; tail -3 gpurun_out/r05_bench_drv.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_drv.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"))
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d["roofline_latency"].items() if k in ("us_per_round","model_us_per_round","frac","rounds")})
oc=d["other_configs"]
for k,v in oc.items():
    if isinstance(v,dict): print(k, {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","ms_per_submap","ms_64_iters","ms_1024_iters","error")})
print([ (p["batch"],p["input"],round(p["submaps_per_s"])) for p in oc["configs1_sweep"]["points"]])
print(oc["configs3_training_step"]["roofline"]["frac"], oc["configs3_training_step"]["roofline"].get("traffic"))
PY
