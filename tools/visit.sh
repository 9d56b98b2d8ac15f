cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f16.py -m gpu -q -x 2>&1 | tail -5
for i in 1 2; do
for v in 0 1; do
PA_ENGINE_FP0_F16=$v timeout 300 python bench.py --model pptnet --mlp-dtype f16 --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pptnet f16 fp0half=$v', round(d['value']), d['ms_per_step'])"
PA_ENGINE_FP0_F16=$v timeout 300 python bench.py --mlp-dtype f16 --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras --no-kernel-pass 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('patchaugnet f16 fp0half=$v', round(d['value']), d['ms_per_step'])"
done; done
