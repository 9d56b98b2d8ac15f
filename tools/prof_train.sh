#!/bin/bash
# kernel stats of the training step as bench.py runs it (one hipGraph replay per step; the trainer warm-up and the capture pass run the same kernels eagerly): tools/prof_train.sh TAG
TAG=${1:-t}; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} -- python bench.py --config train --steps 10 --warmup 3 --no-prefetch > gpurun_out/${TAG}_prof.log 2>&1; echo "rocprof rc=$?"
DB=$(ls gpurun_out/${TAG}_prof/*results.db | head -1)
python tools/rocprof_summary.py $DB gpurun_out/${TAG}_train_step_kernel_stats.csv
rm -rf gpurun_out/${TAG}_prof
python - <<PY
import csv
rows=list(csv.reader(open("gpurun_out/${TAG}_train_step_kernel_stats.csv")))[1:]
steps=10.0+3.0+3.0+1.0  # timed + bench warm-up + trainer warm-up + capture
tot=sum(float(r[2]) for r in rows)
nat=[r for r in rows if "at::native" in r[0] or "rocclr" in r[0]]
print("GPU us/step %.0f  native us/step %.0f (%.1f%%)  native launches/step %.0f  all launches/step %.0f" % (tot/steps, sum(float(r[2]) for r in nat)/steps, 100*sum(float(r[2]) for r in nat)/tot, sum(int(r[1]) for r in nat)/steps, sum(int(r[1]) for r in rows)/steps))
for r in sorted(nat, key=lambda r:-float(r[2]))[:28]:
    print("%8.1f us/step %6.1f calls/step  %s" % (float(r[2])/steps, int(r[1])/steps, r[0][:140]))
PY
