#!/bin/bash
# One GPU-box visit: parity tests, default bench line, rocprofv3 kernel stats of the same bench command (+ a 1-stream trace whose
# per-kernel durations are free of cross-stream overlap), PMC traffic of the dominant kernels.
# usage: gpurun --timeout 1800 -- 'bash tools/gpu_check.sh TAG'
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
tail -5 gpurun_out/${TAG}_tests.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
cat gpurun_out/${TAG}_bench.json
# the driver's round-end command line (20 timed steps: pipeline fill / drain weigh ~5 %), three times
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2>> gpurun_out/${TAG}_bench.err; done > gpurun_out/${TAG}_bench_driver_protocol.json
python - <<'PY'
import json, glob
f = sorted(glob.glob("gpurun_out/*_bench_driver_protocol.json"))[-1]
print("driver protocol (--steps 20 --warmup 5):", [round(json.loads(l)["value"]) for l in open(f).read().strip().splitlines() if l.startswith("{")])
PY
# the other BASELINE configurations: PPT-Net (fp32 and the fp16 MLP path = configs[4]), PatchAugNet fp16 MLP path, section-8(d) sweep
timeout 300 python bench.py --model pptnet --no-cpu-baseline > gpurun_out/${TAG}_bench_pptnet_f32.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --model pptnet --mlp-dtype f16 --no-cpu-baseline > gpurun_out/${TAG}_bench_pptnet_f16.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --mlp-dtype f16 --no-cpu-baseline > gpurun_out/${TAG}_bench_patchaugnet_f16.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python tools/config_sweep.py > gpurun_out/${TAG}_config_sweep.json 2>> gpurun_out/${TAG}_bench.err
# BASELINE configs[3]: the training step (bench line + kernel stats of the same command)
timeout 300 python bench.py --config train --steps 30 --warmup 5 > gpurun_out/${TAG}_bench_train.json 2>> gpurun_out/${TAG}_bench.err
rm -rf gpurun_out/${TAG}_prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} -- python bench.py --config train --steps 10 --warmup 3 > gpurun_out/${TAG}_prof_train.log 2>&1; echo "rocprof(train) rc=$?"
python tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*results.db | head -1) gpurun_out/${TAG}_train_step_kernel_stats.csv
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/*_bench_*.json")):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(l["value"]), l["unit"])
    except Exception as ex:
        print(f, "unreadable", ex)
PY
rm -rf gpurun_out/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} -- python bench.py --model pptnet --mlp-dtype f16 --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-extras > gpurun_out/${TAG}_prof_ppt.log 2>&1; echo "rocprof(ppt) rc=$?"
python tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*results.db | head -1) gpurun_out/${TAG}_kernel_stats_pptnet_f16_streams_1.csv
for S in default 1; do
  rm -rf gpurun_out/${TAG}_prof
  EXTRA=""; [ "$S" = "1" ] && EXTRA="--streams 1"
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $EXTRA > gpurun_out/${TAG}_prof_$S.log 2>&1; echo "rocprof($S) rc=$?"
  python tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*results.db | head -1) gpurun_out/${TAG}_kernel_stats_streams_$S.csv
  rm -rf gpurun_out/${TAG}_prof
done
head -8 gpurun_out/${TAG}_kernel_stats_streams_1.csv | cut -c1-160
cd /tmp
# MFMA utilisation of the MFMA-bound kernels: SQ_VALU_MFMA_BUSY_CYCLES (sum over SIMDs) / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_target.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_mfma.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(ls $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma/*results.db | head -1) > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_mfma.txt 2>&1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
python - <<'PY'
import os, re, collections
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
tag = [f for f in os.listdir(root + "/gpurun_out") if f.endswith("_pmc_mfma.txt")][-1]
vals = collections.defaultdict(dict)
for line in open(root + "/gpurun_out/" + tag):
    m = re.match(r"(.*?)\s+(SQ_\w+|GRBM_\w+)\s+n=\s*\d+ avg=([0-9.]+) min=([0-9.]+)", line)
    if m:      # kernel cycles: the MINIMUM over the launches (the first launch of a kernel can be several times longer: cold caches, clock ramp)
        vals[m.group(1).strip()][m.group(2)] = float(m.group(4) if m.group(2) == "GRBM_GUI_ACTIVE" else m.group(3))
out = []
for k, v in vals.items():
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and v.get("GRBM_GUI_ACTIVE", 0) > 0:
        util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)
        out.append((util, k))
with open(root + "/gpurun_out/" + tag.replace("_pmc_mfma.txt", "_mfma_util.txt"), "w") as f:
    f.write("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); busy cycles averaged, kernel cycles the minimum over the launches of tools/pmc_target.py\n")
    for util, k in sorted(out, reverse=True):
        f.write(f"{util:6.1%}  {k[:150]}\n")
        print(f"{util:6.1%}  {k[:110]}")
PY
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_target.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$C.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$C/*results.db | head -1) > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$C.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$C
  grep -E "chain_kernel<1, 16, 3|group_lds_kernel<4>|knn_grid|vlad_accum_kernel<4>" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$C.txt | cut -c1-60,90-200
done

# round 4: counters of the fp16 PPT-Net step (configs[4]): MFMA-busy, L2 traffic, L1 accesses of the fp16 chain / attention kernels
cd $GRAFT_REPO_ROOT
bash tools/pmc_pass.sh gpurun_out/${TAG}_ppt16_pmc_mfma.txt "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${TAG}_ppt16_pmc_FETCH_SIZE.txt "FETCH_SIZE" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${TAG}_ppt16_pmc_WRITE_SIZE.txt "WRITE_SIZE" --model pptnet --mlp-dtype f16 --no-grouping
bash tools/pmc_pass.sh gpurun_out/${TAG}_ppt16_pmc_tcp.txt "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" --model pptnet --mlp-dtype f16 --no-grouping
timeout 300 python tools/emd_time.py > gpurun_out/${TAG}_emd_time.txt 2>&1
timeout 300 python tools/fpx16_time.py > gpurun_out/${TAG}_fpx16_time.txt 2>&1
timeout 300 python tools/sa1_time.py > gpurun_out/${TAG}_sa1_time.txt 2>&1
timeout 300 python tools/probes/stage_b2b.py 9 > gpurun_out/${TAG}_stage_b2b.txt 2>&1
timeout 300 python tools/chain_phases.py fp0 fp1 fp2 sa1 sa2 > gpurun_out/${TAG}_chain_phases.txt 2>&1
timeout 300 python tools/fps_time.py > gpurun_out/${TAG}_fps_time.txt 2>&1
timeout 300 python tools/knn_time.py > gpurun_out/${TAG}_knn_time.txt 2>&1
timeout 300 python tools/tnn_time.py > gpurun_out/${TAG}_tnn_time.txt 2>&1
