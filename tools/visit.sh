cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --config oxford 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('value','extraction_s','extraction_from_pinned_host_s','host_and_resident_descriptors_identical','retrieval_ms','recall_delta_pp','descriptor_probe_max_abs_diff_vs_oracle'): print(k, d[k])"
for M in "--model pptnet --mlp-dtype f16" "--mlp-dtype f16"; do python bench.py $M --no-extras --no-cpu-baseline --no-pmc --no-trace 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'][:40], d['dtype'][:10], round(d['value']), 'plain', d.get('plain_graph_pipeline',{}).get('value'))"; done
python bench.py --model pptnet --mlp-dtype f16 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --no-trace 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pptnet f16 20 steps', round(d['value']), 'plain', d.get('plain_graph_pipeline',{}).get('value'))"
