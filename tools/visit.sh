cd /root/repo
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do
  echo "== wait-free epilogue"; python tools/tgemm_cm_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-48,95-
  echo "== previous (pu1 build)"; PA_TGEMM_CM_ONLY=fp0 PA_LIB_PATH=patchaugnet_amd/csrc/ab/libpa_pu1.so python tools/tgemm_cm_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-48,95-
done
