cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_group_modules.py tests/test_gpu_models.py tests/test_gpu_train_full.py tests/test_gpu_train_glue.py tests/test_gpu_train_ops.py tests/test_gpu_extract.py -m gpu -q -x 2>&1 | tail -5
