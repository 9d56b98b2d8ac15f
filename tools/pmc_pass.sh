#!/bin/bash
# One rocprofv3 --pmc pass over tools/pmc_target.py:  tools/pmc_pass.sh OUT_TXT "COUNTER1 COUNTER2 ..." [pmc_target args...]
# (counters in their own run with --kernel-trace only; never with the hip / hsa trace domains)
OUT=$1; CNT=$2; shift 2
case $OUT in /*) ;; *) OUT=${GRAFT_REPO_ROOT:-/root/repo}/$OUT ;; esac
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
D=$ROOT/gpurun_out/pmc_tmp_$$
rm -rf $D
timeout 300 rocprofv3 --kernel-trace --pmc $CNT -d $D -o pmc -- python $ROOT/tools/pmc_target.py "$@" > $D.log 2>&1
python $ROOT/tools/pmc_summary.py $(ls $D/*results.db $D/*/*results.db 2>/dev/null | head -1) > $OUT 2>&1
tail -5 $D.log; rm -rf $D $D.log
