cd /root/repo
timeout 600 tools/probes/pk_f32_fault_repro.bin 2>&1 | tee gpurun_out/pk_f32_fault_repro.txt | tail -22
