cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/probes/phase_offset.py 20 2>&1 | grep -v amdgpu.ids
python tools/probes/phase_offset.py 100 2>&1 | grep -v amdgpu.ids | head -8
