# scratch script for one gpurun visit (bash tools/gpr.sh gpurun_out/vNNN.log TIMEOUT 'bash tools/visit.sh'); the round's measurement set is tools/final_r05.sh
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-400
