#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy:  tools/gpr.sh LOG TIMEOUT 'command'
LOG=$1; TO=$2; shift 2
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $TO -- "$@" > $LOG 2>&1
  if grep -q "status=transient" $LOG; then sleep 45; continue; fi
  break
done
