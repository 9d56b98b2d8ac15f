#!/bin/bash
export TMPDIR=/tmp
TAG=r04
mkdir -p gpurun_out
for S in 1 default; do
  rm -rf gpurun_out/${TAG}_prof
  EXTRA=""; [ "$S" = "1" ] && EXTRA="--streams 1"
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $EXTRA > gpurun_out/${TAG}_prof_$S.log 2>&1; echo "rocprof($S) rc=$?"
  python tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*results.db gpurun_out/${TAG}_prof/*/*results.db 2>/dev/null | head -1) gpurun_out/${TAG}_kernel_stats_streams_$S.csv
  rm -rf gpurun_out/${TAG}_prof
  grep "^{" gpurun_out/${TAG}_prof_$S.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench under rocprof', round(d['value']), d['roofline']['ms_per_launch'], d['roofline']['frac'])"
done
grep "chain_kernel<1, 16, 3" gpurun_out/${TAG}_kernel_stats_streams_1_by_grid.csv | cut -c1-200
