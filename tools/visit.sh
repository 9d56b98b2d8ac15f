cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_group_modules.py tests/test_gpu_models.py tests/test_gpu_train_full.py -m gpu -q 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_glue.py tests/test_gpu_extract.py -m gpu -q 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --config train --steps 30 --warmup 5 --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], d['losses_last_step'])"
done
