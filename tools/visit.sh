cd /root/repo
timeout 900 python -m pytest tests/test_gpu_losses.py tests/test_losses_io.py -m gpu -q 2>&1 | tail -3
python bench.py --config train --steps 30 --warmup 5 --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step ms', d['ms_per_step'])"
timeout 600 python tools/train_libops.py > gpurun_out/libops.txt 2>&1; echo rc=$?
