#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_models.py tests/test_gpu_f16.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -3
timeout 200 python tools/chain_phases.py sa1 sa2 2>&1 | grep -v offsets
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d[\"value\"]), {k: round(v,3) for k,v in d[\"kernels\"][\"stages_ms\"].items() if \"chain\" in k or \"premul\" in k})"
