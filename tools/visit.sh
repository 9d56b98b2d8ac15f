cd /root/repo
echo "== pre-fix library (commit a0a1c2c)"
PA_LIB_PATH=/root/repo/patchaugnet_amd/csrc/ab/lib_prefix.so timeout 900 python -m pytest tests/test_gpu_f16.py -q -k "beside_fp16" 2>&1 | grep -v amdgpu.ids | tail -4
PA_LIB_PATH=/root/repo/patchaugnet_amd/csrc/ab/lib_prefix.so timeout 900 python -m pytest tests/test_gpu_extract.py -q -k "graphed_extractor_distinct" 2>&1 | grep -v amdgpu.ids | tail -4
echo "== shipped library"
timeout 900 python -m pytest tests/test_gpu_f16.py tests/test_gpu_extract.py -q -k "beside_fp16 or graphed_extractor_distinct" 2>&1 | grep -v amdgpu.ids | tail -2
