cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_full.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|^tests.*Error" | head -20
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_glue.py tests/test_gpu_models.py -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | head
for i in 1 2; do
timeout 300 python bench.py --config train --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'])"
done
