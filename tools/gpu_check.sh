#!/bin/bash
# One GPU-box visit: parity tests, default bench line, rocprofv3 kernel stats of the same bench command.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_check.sh TAG'
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
tail -5 gpurun_out/${TAG}_tests.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
cat gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_prof.log 2>&1; echo "rocprof rc=$?"
python tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*results.db | head -1) gpurun_out/${TAG}_kernel_stats.csv
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
