cd /root/repo
timeout 1200 python tools/fuzz_tgemm_cm.py 250 1 2>&1 | grep -v amdgpu.ids | tail -12
PA_TGEMM_CM_CT=4 timeout 900 python tools/fuzz_tgemm_cm.py 120 2 2>&1 | grep -v amdgpu.ids | tail -6
PA_TGEMM_CM_CT=2 timeout 900 python tools/fuzz_tgemm_cm.py 120 3 2>&1 | grep -v amdgpu.ids | tail -6
