cd /root/repo
for i in 1 2 3; do python tools/probes/cm_shape.py 3 64 40960 32 1 2>&1 | grep -v amdgpu.ids | head -1; done
python tools/probes/cm_shape.py 18 64 20480 32 1 2>&1 | grep -v amdgpu.ids | head -1
python tools/probes/cm_shape.py 3 128 40960 32 1 2>&1 | grep -v amdgpu.ids | head -1
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x 2>&1 | tail -3
timeout 1800 python -m pytest tests/test_gpu_models.py tests/test_gpu_train_full.py tests/test_gpu_train_glue.py tests/test_gpu_extract.py -m gpu -q 2>&1 | tail -3
PA_TGEMM_CM_ONLY=fp0 python tools/tgemm_cm_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-48,95-
python bench.py --config train --steps 30 --warmup 5 --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step ms', d['ms_per_step'], d['roofline']['frac'])"
