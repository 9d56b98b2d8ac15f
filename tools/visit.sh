cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2600 python tests/fuzz_gpu.py 40 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r06f_fuzz.log; cat gpurun_out/r06f_fuzz.log
timeout 600 python tools/probes/stream_determinism_soak.py 10 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/r06f_fuzz.log; tail -5 gpurun_out/r06f_fuzz.log
