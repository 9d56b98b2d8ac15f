cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_extract.py -x -q -m gpu > gpurun_out/v35_extract.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|Fatal\|Error\|error\|test_gpu_extract.py" gpurun_out/v35_extract.log | head -30; head -60 gpurun_out/v35_extract.log | cut -c1-250
