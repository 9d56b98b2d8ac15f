cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|Error" | head -12
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x -k "graphed" 2>&1 | grep -E "^E  |passed|failed|Error" | head -12
PA_FP_NO_FOLD=1 timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|Error" | head -12
