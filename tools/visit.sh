cd /root/repo
timeout 300 tools/probes/hbm_mix.bin
