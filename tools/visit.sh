cd /root/repo
for i in 1 2 3; do python tools/probes/graph_replay_gradients_prefetch.py 12 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -3; done
