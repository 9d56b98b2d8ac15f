#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_extract.py -m gpu -q 2>&1 | tail -4
bash tools/ab_env.sh "PA_ENGINE_NO_PRESORT=1" 2>&1 | grep -v "^sa0.fps.*fp0.3nn" 
