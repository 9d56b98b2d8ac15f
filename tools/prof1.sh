#!/bin/bash
# 1-stream kernel stats + per-dispatch sequence of the bench step: tools/prof1.sh TAG [extra bench args]
TAG=${1:-p}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-pmc --no-extras --streams 1 "$@" > gpurun_out/${TAG}_prof.log 2>&1; echo "rocprof rc=$?"
DB=$(ls gpurun_out/${TAG}_prof/*results.db | head -1)
python tools/rocprof_summary.py $DB gpurun_out/${TAG}_kernel_stats_streams_1.csv
python tools/rocprof_trace.py $DB > gpurun_out/${TAG}_trace.txt 2>&1 || python tools/rocprof_trace.py $DB --schema > gpurun_out/${TAG}_schema.txt 2>&1
rm -rf gpurun_out/${TAG}_prof
head -34 gpurun_out/${TAG}_kernel_stats_streams_1.csv | cut -c1-200
