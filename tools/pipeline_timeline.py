#!/usr/bin/env python
"""Four-stream pipeline picture from a rocprofv3 --kernel-trace database: per HIP queue the idle gaps between its consecutive kernels, per kernel
the duration inside the pipeline, and how many chip-filling kernels overlap.   python tools/pipeline_timeline.py x_results.db [--dump N]"""
import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
view = [n for n, in c.execute("select name from sqlite_master where type='view' and name like 'kernels%'")][-1]
cols = [r[1] for r in c.execute(f"pragma table_info('{view}')")]
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
gy = "grid_y" if "grid_y" in cols else "grid_size_y"
wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
scol = "stream_id" if "stream_id" in cols else None
sel = f"select name, {gx}, {gy}, {wx}, start, end, {qcol or 0}, {scol or 0} from {view} order by start"
rows = [r for r in c.execute(sel) if "at::native" not in r[0]]
def short(n, x, y, w):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
    return (m.group(1) if m else n[:40]) + f"[{x // max(w, 1)},{y}]"
# steady state: the middle half of the fps<256,16> launches
marks = [i for i, r in enumerate(rows) if "fps_reg_kernel<512, 8" in r[0]]
lo, hi = marks[len(marks) // 4], marks[3 * len(marks) // 4]
win = rows[lo:hi]
t0, t1 = win[0][4], win[-1][4]
nsteps = sum(1 for r in win if "fps_reg_kernel<512, 8" in r[0])
print(f"window {(t1 - t0) / 1e6:.3f} ms, {nsteps} steps -> {(t1 - t0) / 1e3 / nsteps:.1f} us per step; columns: {cols}")
byq = collections.defaultdict(list)
for r in win:
    byq[(r[6], r[7])].append(r)
print(f"{len(byq)} queues")
gap_before = collections.defaultdict(list)
for q, rs in byq.items():
    busy = sum(e - s for _, _, _, _, s, e, _, _ in rs)
    span = rs[-1][5] - rs[0][4]
    print(f"  queue {q}: {len(rs)} kernels, busy {busy / span:.1%} of its span")
    for a, b in zip(rs[:-1], rs[1:]):
        gap_before[short(*b[:4])].append((b[4] - a[5]) / 1e3)
dur = collections.defaultdict(list)
for r in win:
    dur[short(*r[:4])].append((r[5] - r[4]) / 1e3)
print("kernel[grid]                                             n   avg_us   avg_gap_before_us (same queue)   sum/step_us  gap/step_us")
tot_d = tot_g = 0.0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    g = gap_before.get(k, [0.0])
    tot_d += sum(v) / nsteps; tot_g += sum(g) / nsteps
    print(f"{k[:56]:56s} {len(v):4d} {sum(v) / len(v):8.1f} {sum(g) / max(len(g), 1):10.1f} {sum(v) / nsteps:22.1f} {sum(g) / nsteps:10.1f}")
print(f"per step: kernel time {tot_d:.1f} us, same-queue gaps {tot_g:.1f} us")
# concurrency of chip-filling kernels (grid >= 128 workgroups)
ev = []
for n, x, y, w, s, e, _, _ in win:
    if (x // max(w, 1)) * max(y, 1) >= 128:
        ev += [(s, 1), (e, -1)]
ev.sort()
hist = collections.Counter(); d = 0; last = ev[0][0]
for t, dd in ev:
    hist[d] += t - last; d += dd; last = t
tot = sum(hist.values())
print("chip-filling kernels (>= 128 workgroups) running at once: " + "  ".join(f"{k}: {v / tot:.1%}" for k, v in sorted(hist.items())))
if "--dump" in sys.argv:
    n = int(sys.argv[sys.argv.index("--dump") + 1])
    for r in win[:n]:
        print(f"{(r[4] - t0) / 1e3:9.1f} {(r[5] - r[4]) / 1e3:8.1f} q={r[6]}/{r[7]} {short(*r[:4])}")
