#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f16.py tests/test_gpu_attention.py -m gpu -q -x 2>&1 | grep -E "Error|error|assert|passed|failed" | head -20
timeout 300 python bench.py --model pptnet --mlp-dtype f32 --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -5 | cut -c1-400
