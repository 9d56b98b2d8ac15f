cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pairwise" 2>&1 | tail -15
