cd /root/repo
timeout 900 python tools/probes/train_soak.py 300 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-1200
