cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x -k "tgemm_kk" 2>&1 | tail -3
timeout 300 python tools/tgemm_kk_time.py 2>&1 | grep -v amdgpu.ids
for sp in 4 8 16; do echo "splits=$sp"; PA_KK128_SPLITS=$sp timeout 300 python tools/tgemm_kk_time.py 2>&1 | grep -E "fp0|fp1 dW"; done
