#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
for m in patch_aug_net pptnet; do
rm -rf gpurun_out/prof_f16
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_f16 -o f16 -- python bench.py --model $m --mlp-dtype f16 --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-extras --no-pmc > gpurun_out/prof_f16_$m.log 2>&1
python tools/rocprof_summary.py $(ls gpurun_out/prof_f16/*results.db gpurun_out/prof_f16/*/*results.db 2>/dev/null | head -1) gpurun_out/r04_${m}_f16_1stream_kernel_stats.csv
rm -rf gpurun_out/prof_f16
done
