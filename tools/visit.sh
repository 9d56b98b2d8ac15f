cd /root/repo
echo "== round-4 library (memset nodes)"; DBG_ROOT=/root/repo/.ab_head python tools/probes/dbg_extract_old.py 2>&1 | grep -v amdgpu.ids | tail -4
echo "== current"; python tools/probes/dbg_extract_old.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1500 python -m pytest tests/test_gpu_extract.py tests/test_gpu_train_ops.py tests/test_gpu_losses.py tests/test_gpu_chain.py -m gpu -q -x 2>&1 | tail -3
