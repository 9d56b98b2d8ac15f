#!/bin/bash
export TMPDIR=/tmp
PA_KNN_QUAD_NMIN=1024 PA_KNN_QUAD_MMIN=128 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -k "knn or reference or models" 2>&1 | tail -2
bash tools/ab_env.sh "PA_KNN_QUAD_NMIN=1024 PA_KNN_QUAD_MMIN=128" 2>&1 | grep -E "===|^value|sa1.knn" | sed -E 's/sa0.fps.*sa1.knn=([0-9.]+).*/sa1.knn=\1/'
