#!/bin/bash
# Quick perf visit: chain parity tests, default-protocol bench line, per-phase cycle stamps of the dominant kernel, 1-stream kernel stats.
# usage: gpurun --timeout 900 -- 'bash tools/perf_check.sh TAG'
TAG=${1:-perf}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_models.py tests/test_gpu_f16.py tests/test_gpu_fuzz.py tests/test_gpu_head.py tests/test_gpu_attention.py -m gpu -q 2>&1 | tail -8
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],4))
st=d['kernels']['stages_ms']
print(' '.join(f'{k}={v:.3f}' for k,v in st.items()))
print('roofline',round(d['roofline']['achieved'],2),round(d['roofline']['frac'],4))
PY
timeout 120 python tools/chain_phases.py fp0 fp1 fp2 sa0 sa1 sa2 2>&1 | tail -8
rm -rf gpurun_out/${TAG}_prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --streams 1 > gpurun_out/${TAG}_prof.log 2>&1; echo "rocprof rc=$?"
python tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*results.db | head -1) gpurun_out/${TAG}_kernel_stats_streams_1.csv
rm -rf gpurun_out/${TAG}_prof
head -22 gpurun_out/${TAG}_kernel_stats_streams_1.csv | cut -c1-150
