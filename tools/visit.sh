cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fps or furthest or sampling" 2>&1 | tail -2
PA_FPS_NT512=1 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fps or furthest or sampling" 2>&1 | tail -2
bash tools/ab_env.sh "PA_FPS_LDS_RESERVE=0" "PA_FPS_NT512=1" "PA_FPS_NT512=1 PA_FPS_LDS_RESERVE=0" 2>&1 | grep -v "^sa0.fps"
