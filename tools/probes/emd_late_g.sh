# workgroups per cloud over the round index (PA_EMD_SCHED): (16, 4096, 3), 64 / 1024 rounds
for cfg in "PA_EMD_SCHED=0:16" "PA_EMD_SCHED=4:16" "PA_EMD_SCHED=8:16" "PA_EMD_SCHED=12:16" "PA_EMD_SCHED=16:16" "PA_EMD_SCHED=8:16" "PA_EMD_SCHED=16:16"; do echo "== $cfg"; env $cfg python tools/emd_time.py 2>&1 | grep "default (chip" ; done
