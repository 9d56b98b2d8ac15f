#!/bin/bash
# Round-5 measurement set: everything DESIGN.md section 5 cites, written under gpurun_out/r05f_* (copied into profiles/ by hand afterwards).
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
T=r05f
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${T}_gpu_tests.log; tail -1 gpurun_out/${T}_gpu_tests.log
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1; done > gpurun_out/${T}_bench_driver_protocol.json
timeout 900 python bench.py --config train > gpurun_out/${T}_bench_train.json 2> gpurun_out/${T}_bench_train.err; echo "train rc=$?"
timeout 600 python bench.py --model pptnet --mlp-dtype f16 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > gpurun_out/${T}_bench_pptnet_f16.json
timeout 600 python bench.py --model pptnet --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > gpurun_out/${T}_bench_pptnet_f32.json
timeout 600 python bench.py --mlp-dtype f16 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > gpurun_out/${T}_bench_patchaugnet_f16.json
python - <<'PY'
import json
def last(f):
    try: return json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1])
    except Exception as e: return {"error": str(e)}
d = last("gpurun_out/r05f_bench.json"); print("bench", round(d.get("value", 0)), d.get("ms_per_step"), d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("traffic"), d.get("roofline_latency", {}).get("frac"), d.get("cpu_baseline", {}).get("value"))
print("driver protocol", [round(json.loads(l)["value"]) for l in open("gpurun_out/r05f_bench_driver_protocol.json").read().strip().splitlines() if l.startswith("{")])
t = last("gpurun_out/r05f_bench_train.json"); print("train", t.get("ms_per_step"), t.get("roofline", {}).get("frac"), t.get("roofline", {}).get("traffic"))
for k in ("pptnet_f16", "pptnet_f32", "patchaugnet_f16"):
    x = last(f"gpurun_out/r05f_bench_{k}.json"); print(k, round(x.get("value", 0)), x.get("ms_per_step"))
PY
# kernel statistics: headline on one stream and on four, training step per graph replay
for S in 1 4; do
  rm -rf gpurun_out/${T}_prof
  timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -o ${T} -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-pmc --no-extras --streams $S > gpurun_out/${T}_prof.log 2>&1; echo "rocprof streams=$S rc=$?"
  python tools/rocprof_summary.py $(ls gpurun_out/${T}_prof/*results.db | head -1) gpurun_out/${T}_fused_${S}stream_kernel_stats.csv > /dev/null
  rm -rf gpurun_out/${T}_prof
done
bash tools/prof_train_diff.sh ${T} 10 50 2>&1 | head -3
bash tools/pmc_tgemm.sh > /dev/null 2>&1; for i in 1 2 3; do grep "tgemm_cm" gpurun_out/pmc_tg$i.txt; done > gpurun_out/${T}_tgemm_cm_pmc_raw.txt
python tools/train_gemm_shapes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_train_gemm_shapes_after.txt
