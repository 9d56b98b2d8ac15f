cd /root/repo
timeout 1500 python tools/probes/stream_determinism_soak.py 40 2>&1 | grep -v amdgpu.ids | tail -5
