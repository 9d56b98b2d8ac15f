cd /root/repo
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x -k "weight_gradient or tgemm_kk" 2>&1 | tail -3
python tools/kkd_time.py 2>&1 | grep -v amdgpu.ids
echo "== no lockstep barrier"; PA_TGEMM_KKD_NOSYNC=1 PA_KKD_ONLY=fp0 python tools/kkd_time.py 2>&1 | grep -v amdgpu.ids
echo "== 1024 waves"; PA_TGEMM_KKD_WAVES=1024 PA_KKD_ONLY=fp0 python tools/kkd_time.py 2>&1 | grep -v amdgpu.ids
echo "== 4096 waves"; PA_TGEMM_KKD_WAVES=4096 PA_KKD_ONLY=fp0 python tools/kkd_time.py 2>&1 | grep -v amdgpu.ids
