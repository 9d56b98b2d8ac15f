cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_full.py tests/test_gpu_train_glue.py tests/test_gpu_retrieval_mfma.py -m gpu -q -x 2>&1 | tail -5
for i in 1 2; do
PA_TGEMM_NO_CM=1 timeout 300 python bench.py --config train --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old kernel only', d['ms_per_step'])"
timeout 300 python bench.py --config train --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('with lds-resident', d['ms_per_step'], d['roofline'])"
done
timeout 300 python tools/train_gemm_shapes.py 2>&1 | grep -v amdgpu | head -30
