cd /root/repo
for shape in "4 8192" "64 2048" "16 1024"; do
timeout 900 python tools/probes/stream_determinism_soak.py 10 $shape 2>&1 | grep -v amdgpu.ids | tail -4
done
