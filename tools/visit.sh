cd /root/repo
python - <<'PY'
import subprocess, time, json
for cmd in (["python", "bench.py"], ["python", "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"], ["python", "bench.py", "--config", "train"]):
    t = time.time(); r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True); dt = time.time() - t
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(" ".join(cmd), "->", round(dt, 1), "s wall;", round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4), "roofline", round(d["roofline"]["frac"], 3))
    if "train" in cmd: open("gpurun_out/r05h_bench_train.json", "w").write(r.stdout.strip().splitlines()[-1] + "\n")
PY
bash tools/prof_train_diff.sh r05h 10 50 2>&1 | head -3
