cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_full.py tests/test_gpu_group_modules.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do python bench.py --config train --steps 40 --warmup 5 --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step ms', round(d['ms_per_step'], 4))"; done
