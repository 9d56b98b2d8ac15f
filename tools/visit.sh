cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
FUZZ_ONLY=sa_tiny,fpx32,chain_sa,chain_fp,fps timeout 1500 python tests/fuzz_gpu.py 60 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/v28_fuzz.txt; cat gpurun_out/v28_fuzz.txt
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "sa_tiny or fpx32" 2>&1 | tail -3
