cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/v36_tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/v36_tests.log | cut -c1-200
