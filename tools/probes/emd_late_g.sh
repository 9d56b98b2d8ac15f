# (16, 4096, 3), 64 / 1024 rounds: workgroups per cloud over the round index (PA_EMD_SCHED="from:G,...") on the product library
for cfg in "PA_X=0" "PA_EMD_SCHED=64:8" "PA_EMD_SCHED=200:8" "PA_EMD_SCHED=100:12" "PA_EMD_SCHED=300:8,700:4" "PA_X=0"; do echo "== [$cfg]"; env $cfg python tools/emd_time.py 2>&1 | grep "default (chip" ; done
