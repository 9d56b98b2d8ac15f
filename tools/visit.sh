cd /root/repo
python bench.py --config train > gpurun_out/r05j_bench_train.json 2>/dev/null; tail -1 gpurun_out/r05j_bench_train.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('train full', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'])"
bash tools/prof_train_diff.sh r05j 10 50 2>&1 | head -3
