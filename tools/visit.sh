cd /root/repo
for f in 4 0; do echo "== victim form $f"; VICTIM_FORM=$f CORUN_MODES=30,8,13,-1 timeout 600 python tools/probes/pk_f32_victim2.py 8 2>&1 | grep -v amdgpu.ids | tail -4; done
