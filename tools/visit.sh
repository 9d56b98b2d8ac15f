cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_ops.py -m gpu -q -x -k "shared_resident or pairwise" 2>&1 | tail -5
timeout 300 python tools/train_gemm_shapes.py > gpurun_out/r05_train_gemm_shapes_before.txt 2>&1; cat gpurun_out/r05_train_gemm_shapes_before.txt
bash tools/pmc_pass.sh gpurun_out/r05_fps_pmc_sq_a.txt "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" --no-grouping
bash tools/pmc_pass.sh gpurun_out/r05_fps_pmc_sq_b.txt "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" --no-grouping
grep -h "fps_reg_kernel<256, 16" gpurun_out/r05_fps_pmc_sq_a.txt gpurun_out/r05_fps_pmc_sq_b.txt | cut -c1-40,150-260
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_v2.json 2> gpurun_out/r05_bench_v2.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_bench_v2.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_v2.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"]); print(json.dumps(d.get("roofline_latency"))[:900]); print(json.dumps(d["other_configs"].get("configs1_sweep"))[:1500])
PY
