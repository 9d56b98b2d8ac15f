#!/bin/bash
# Kernel time of ONE graph-replayed training step: two rocprofv3 --kernel-trace runs of bench.py --config train (S1 and S2 timed steps), per-kernel
# difference / (S2 - S1) -- the trainer's eager warm-up, the capture pass and bench warm-up cancel.  tools/prof_train_diff.sh TAG [S1 S2]
TAG=${1:-t}; S1=${2:-10}; S2=${3:-50}; mkdir -p gpurun_out; export TMPDIR=/tmp
for S in $S1 $S2; do
  rm -rf gpurun_out/${TAG}_prof
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} -- python bench.py --config train --steps $S --warmup 3 --no-pmc > gpurun_out/${TAG}_prof.log 2>&1; echo "rocprof($S) rc=$?"
  python tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*results.db | head -1) gpurun_out/${TAG}_steps_$S.csv > /dev/null
  rm -rf gpurun_out/${TAG}_prof
done
python - <<PY
import csv
def load(f): return {r['kernel']:(int(r['calls']),float(r['total_us'])) for r in csv.DictReader(open(f))}
a,b=load("gpurun_out/${TAG}_steps_$S1.csv"),load("gpurun_out/${TAG}_steps_$S2.csv")
n=$S2-$S1
rows=[]
for k,(c2,t2) in b.items():
    c1,t1=a.get(k,(0,0.0))
    if c2>c1: rows.append((k,(c2-c1)/n,(t2-t1)/n))
rows.sort(key=lambda r:-r[2])
tot=sum(r[2] for r in rows)
lib=[r for r in rows if 'at::native' in r[0] or 'rocclr' in r[0]]
with open("gpurun_out/${TAG}_train_step_per_replay.csv","w") as f:
    f.write("kernel,launches_per_step,us_per_step,percent\n")
    for k,c,t in rows: f.write('"%s",%.2f,%.2f,%.2f\n'%(k,c,t,100*t/tot))
print("per graph-replayed step: GPU %.0f us in %.0f launches; library (at::native / rocclr) %.0f us = %.1f %% in %.0f launches"%(tot,sum(r[1] for r in rows),sum(r[2] for r in lib),100*sum(r[2] for r in lib)/tot,sum(r[1] for r in lib)))
for k,c,t in rows[:40]: print("%8.1f us %6.1f x  %s"%(t,c,k.replace('(anonymous namespace)::','')[:150]))
print("--- library")
for k,c,t in lib[:30]: print("%8.1f us %6.1f x  %s"%(t,c,k[:170]))
PY
