#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_group_modules.py tests/test_gpu_e2e_recall.py tests/test_gpu_losses.py tests/test_gpu_models.py tests/test_gpu_extract.py -m gpu -q -s 2>&1 | grep -E "flipped|passed|failed|Error|assert" | head -20
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04i_bench.json 2> gpurun_out/r04i_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r04i_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04i_bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), d["ms_per_step"])
print("step_mfma_frac", d.get("step_mfma_frac"))
print("reference_protocol", d.get("reference_protocol"))
for k,v in d.get("other_configs",{}).items():
    print(k, {kk:(vv if not isinstance(vv,dict) or kk.startswith("roofline") else "...") for kk,vv in v.items() if kk not in ("stages_ms","workload","losses_note")})
PY
