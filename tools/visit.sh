cd /root/repo
python tools/probes/graph_memset_repro.py 200 2>&1 | grep -v amdgpu.ids | tail -4
