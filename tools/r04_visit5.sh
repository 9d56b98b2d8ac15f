#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r04e
for S in "" "PA_ATTN_F16_SPLIT=0" "PA_ATTN_F16_TRANS=0" "PA_ATTN_F16_QV=1"; do
echo "== [$S]"; env $S timeout 300 python -m pytest tests/test_gpu_f16.py -m gpu -q -k pptnet 2>&1 | grep -E "assert np.float32|passed|failed" | head -5
done
timeout 200 python tools/chain_phases.py fp1 fp2 sa1 sa2 2>&1 | tail -14
rm -rf gpurun_out/${T}_prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -o ${T} -- python bench.py --model pptnet --mlp-dtype f16 --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-extras > gpurun_out/${T}_prof_ppt.log 2>&1; echo "rocprof(ppt) rc=$?"
python tools/rocprof_summary.py $(ls gpurun_out/${T}_prof/*results.db gpurun_out/${T}_prof/*/*results.db 2>/dev/null | head -1) gpurun_out/${T}_pptnet_f16_1stream_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
head -40 gpurun_out/${T}_pptnet_f16_1stream_kernel_stats.csv | cut -c1-170
