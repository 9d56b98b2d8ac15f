python tools/probes/cumask.py 100 check 0 0 2>&1 | grep -v amdgpu.ids
for cfg in "plain 2 6" "e4 2 6" "e4 2 8" "e4 1 6" "e8 2 6" "e8 2 8" "e3 2 6" "e2 2 6" "e4 4 8" "plain 2 8" "e4 2 5"; do
  python tools/probes/cumask.py 100 $cfg 2>&1 | grep -v amdgpu.ids
done
