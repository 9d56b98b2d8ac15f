#!/bin/bash
# Round-6 measurement set: everything DESIGN.md section 5 cites, written under gpurun_out/r06f_* (copied into profiles/r06_* afterwards).
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
T=r06f
if [ "$1" != "noprof" ]; then
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${T}_gpu_tests.log; tail -1 gpurun_out/${T}_gpu_tests.log
fi
timeout 1200 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-trace --no-pmc 2>/dev/null | tail -1; done > gpurun_out/${T}_bench_driver_protocol.json
timeout 900 python bench.py --config train > gpurun_out/${T}_bench_train.json 2> gpurun_out/${T}_bench_train.err; echo "train rc=$?"
timeout 600 python bench.py --config oxford 2>/dev/null | tail -1 > gpurun_out/${T}_bench_oxford.json
timeout 600 python bench.py --model pptnet --mlp-dtype f16 --no-extras --no-cpu-baseline --no-pmc --no-trace 2>/dev/null | tail -1 > gpurun_out/${T}_bench_pptnet_f16.json
timeout 600 python bench.py --model pptnet --no-extras --no-cpu-baseline --no-pmc --no-trace 2>/dev/null | tail -1 > gpurun_out/${T}_bench_pptnet_f32.json
timeout 600 python bench.py --mlp-dtype f16 --no-extras --no-cpu-baseline --no-pmc --no-trace 2>/dev/null | tail -1 > gpurun_out/${T}_bench_patchaugnet_f16.json
python - <<'PY'
import json
def last(f):
    try: return json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1])
    except Exception as e: return {"error": str(e)}
d = last("gpurun_out/r06f_bench.json"); print("bench", round(d.get("value", 0)), d.get("ms_per_step"), d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("frac_in_pipeline"), d.get("roofline", {}).get("traffic"), d.get("cpu_baseline", {}).get("value"))
for r in d.get("stage_rooflines", {}).get("rows", []): print("   ", r.get("stage"), r.get("us_per_launch"), r.get("us_per_launch_in_pipeline"), r.get("frac"), r.get("frac_in_pipeline"), r.get("error"))
print("   inflation", d.get("stage_rooflines", {}).get("cross_stream_inflation"), d.get("stage_rooflines", {}).get("error"))
o = d.get("other_configs", {}).get("configs2_oxford_eval", {}); print("oxford in line", o.get("value"), o.get("extraction_submaps_per_s"), o.get("recall_delta_pp"), o.get("error"))
print("driver protocol", [round(json.loads(l)["value"]) for l in open("gpurun_out/r06f_bench_driver_protocol.json").read().strip().splitlines() if l.startswith("{")])
t = last("gpurun_out/r06f_bench_train.json"); print("train", t.get("ms_per_step"), t.get("roofline", {}).get("frac"), t.get("step_mfma_frac"))
for k in ("pptnet_f16", "pptnet_f32", "patchaugnet_f16", "oxford"):
    x = last(f"gpurun_out/r06f_bench_{k}.json"); print(k, round(x.get("value", 0)), x.get("ms_per_step"))
PY
[ "$1" = "noprof" ] && exit 0
# kernel statistics: headline and PPT-Net fp16 on one stream and on four, training step per graph replay
for M in "fused:" "pptnet_f16:--model pptnet --mlp-dtype f16"; do
  TAG=${M%%:*}; ARGS=${M#*:}
  for S in 1 4; do
    rm -rf gpurun_out/${T}_prof
    timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -o ${T} -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-pmc --no-extras --no-trace --only-steps --streams $S $ARGS > gpurun_out/${T}_prof.log 2>&1; echo "rocprof $TAG streams=$S rc=$?"
    SUF=${S}stream; [ $S = 4 ] && SUF=4streams
    python tools/rocprof_summary.py $(ls gpurun_out/${T}_prof/*results.db gpurun_out/${T}_prof/*/*results.db 2>/dev/null | head -1) gpurun_out/${T}_${TAG}_${SUF}_kernel_stats.csv > /dev/null
    rm -rf gpurun_out/${T}_prof
  done
done
bash tools/prof_train_diff.sh ${T} 10 50 2>&1 | head -3
# counters of the shipped build (separate passes, --kernel-trace only): MFMA busy, FETCH_SIZE, WRITE_SIZE -- f32 headline and PPT-Net fp16
for M in "pmc:" "pptnet_f16_pmc:--model pptnet --mlp-dtype f16 --no-grouping"; do
  TAG=${M%%:*}; ARGS=${M#*:}
  for C in "mfma_util:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT" "FETCH_SIZE:FETCH_SIZE" "WRITE_SIZE:WRITE_SIZE"; do
    NAME=${C%%:*}; CNT=${C#*:}
    D=gpurun_out/${T}_pmc_tmp; rm -rf $D
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $CNT -d /root/repo/$D -o pmc -- python /root/repo/tools/pmc_target.py $ARGS > /root/repo/$D.log 2>&1)
    DB=$(ls $D/*results.db $D/*/*results.db 2>/dev/null | head -1)
    if [ $NAME = mfma_util ]; then python tools/pmc_mfma_util.py $DB > gpurun_out/${T}_${TAG}_${NAME}.txt 2>&1; else python tools/pmc_summary.py $DB | grep -v "^columns" > gpurun_out/${T}_${TAG}_${NAME}.txt 2>&1; fi
    rm -rf $D $D.log
  done
done
python tools/stage_scaling.py 32 64 128 256 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_stage_scaling.txt
bash tools/pmc_tgemm.sh > /dev/null 2>&1; for i in 1 2 3; do grep "tgemm_cm" gpurun_out/pmc_tg$i.txt; done > gpurun_out/${T}_tgemm_cm_pmc_raw.txt
python tools/train_gemm_shapes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_train_gemm_shapes.txt
