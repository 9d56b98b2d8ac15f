cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_properties.py tests/test_gpu_f16.py tests/test_gpu_fuzz.py tests/test_gpu_extract.py tests/test_gpu_boundary.py -x -q -k "fps or furthest or sampling or beside or fuzz or latency or ahead or range" 2>&1 | tail -2
python tools/fps_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-110
