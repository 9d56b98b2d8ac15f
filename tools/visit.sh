cd /root/repo
for Q in 0 1; do
echo "== PA_ATTN_F16_QV=$Q"
PA_ATTN_F16_QV=$Q python tools/f16_cos.py 2>&1 | grep -v amdgpu.ids | tail -6
for i in 1 2; do PA_ATTN_F16_QV=$Q python bench.py --model pptnet --mlp-dtype f16 --steps 60 --reps 3 --no-cpu-baseline --no-kernel-pass --no-pmc --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pptnet f16', round(d['value']), d['ms_per_step'])"; done
done
