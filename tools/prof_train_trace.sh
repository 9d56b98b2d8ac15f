#!/bin/bash
# per-dispatch launch sequence of ONE training step (eager launches, no prefetch): tools/prof_train_trace.sh TAG
TAG=${1:-tt}; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -o ${TAG} -- python bench.py --config train --steps 4 --warmup 2 --no-prefetch --no-graphs > gpurun_out/${TAG}_prof.log 2>&1; echo "rocprof rc=$?"
DB=$(ls gpurun_out/${TAG}_prof/*results.db | head -1)
python tools/rocprof_trace.py $DB > gpurun_out/${TAG}_train_trace.txt 2>&1
rm -rf gpurun_out/${TAG}_prof
wc -l gpurun_out/${TAG}_train_trace.txt
