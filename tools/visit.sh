cd /root/repo
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r05f_gpu_tests.log; tail -1 gpurun_out/r05f_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
