#!/bin/bash
# look-ahead pipeline: batches per sampling launch x form of the sampling launch over more clouds than CUs (PA_FPS_BIG_FORM: 0 = LDS cloud copy without the
# reserve (two workgroups per CU), -1 = with the reserve, 1 = 256-byte-LDS form 512 threads, 2 = 256-byte-LDS form 256 threads)
mkdir -p gpurun_out; rm -f gpurun_out/gs_*.json
for rep in 1 2; do
for cfg in "16 0" "16 1" "16 2" "32 1" "32 2" "24 1" "16 -1"; do
 set -- $cfg
  for steps in 20 100; do
   PA_FPS_BIG_FORM=$2 python bench.py --gpus 1 --steps $steps --warmup 5 --group $1 --no-trace --no-cpu-baseline --no-kernel-pass --no-pmc --no-extras > gpurun_out/gs_${1}_f${2}_${steps}_$rep.json 2> gpurun_out/gs.err || tail -3 gpurun_out/gs.err
  done
done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/gs_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"]), d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
P
