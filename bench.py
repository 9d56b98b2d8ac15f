#!/usr/bin/env python
"""Benchmark of the descriptor-extraction hot path (BASELINE.json: 4096-pt submaps/sec, PatchAugNet, batch 32).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of 32 synthetic 4096-point submaps already resident in HBM:
(B,1,4096,3) fp32 on device -> (B,256) fp32 descriptors on device.  N > 1: one process per GPU -- either launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment) or, when WORLD_SIZE is not set, by this script itself (one child process per GPU, MASTER_ADDR=127.0.0.1) --
every rank extracts its own shard (weak scaling, no data-path collective), and the timed region ends with ONE RCCL
all-gather of every rank's descriptors (the exchange the retrieval step needs).  Rank 0 prints one JSON line.  Asking for more
GPUs than the box has is an error, never a silent 1-GPU run.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy rate is ~6300
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (same guide)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA peak (same guide; AMD's 5 PF headline includes 2:1 sparsity)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--points", type=int, default=4096)
    p.add_argument("--module-path", action="store_true", help="force the unfused module path instead of the fused engine")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-pass", action="store_true")
    p.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic (it is then null)")
    p.add_argument("--no-trace", action="store_true", help="skip the two rocprofv3 --kernel-trace passes behind stage_rooflines / roofline.frac (the roofline then carries the HIP-event figure only)")
    p.add_argument("--no-lookahead", action="store_true", help="the plain pipeline: one captured graph of the WHOLE step per stream (extract.GraphedExtractor) instead of the "
                                                                "first-level sampling of groups of batches one group ahead (extract.SampledAheadExtractor)")
    p.add_argument("--group", type=int, default=16, help="batches per sampling launch of the look-ahead pipeline (16 x 32 clouds = two sampling workgroups per CU)")
    p.add_argument("--only-steps", action="store_true", help="the timed steps and nothing else (the child run of the kernel-trace passes)")
    p.add_argument("--cpu-batch", type=int, default=8)
    p.add_argument("--model", choices=["patch_aug_net", "pptnet"], default="patch_aug_net", help="pptnet = BASELINE.json configs[4]")
    p.add_argument("--mlp-dtype", choices=["f32", "f16", "f32x3"], default="f32",
                   help="f16: shared-MLP chains on fp16 MFMA (fp32 accumulate; cosine >= 0.999 contract) -- not the headline configuration")
    p.add_argument("--no-graphs", action="store_true", help="issue every step's launches from Python instead of replaying one captured hipGraph per stream")
    p.add_argument("--torch-adam", action="store_true", help="--config train: torch.optim.Adam (fused) instead of patchaugnet_amd.optim.Adam (csrc/adam.hip)")
    p.add_argument("--no-prefetch", action="store_true", help="--config train: no geometry prefetch of the next batch (one graph per step)")
    p.add_argument("--streams", type=int, default=4, help="HIP streams the consecutive steps are issued on (1 = strictly sequential)")
    p.add_argument("--reps", type=int, default=5, help="repetitions of the timed K-step region; the headline value is their median (min / max reported)")
    p.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configurations that ride in the same JSON line at N = 1 "
                                                              "(configs[0] CPU timing, training step, PPT-Net f32 / f16, EMD)")
    p.add_argument("--config", choices=["extract", "train", "oxford"], default="extract",
                   help="extract = BASELINE.json configs[1] (the headline metric); train = configs[3], one quadruplet training step per step; "
                        "oxford = configs[2], the Oxford-sized evaluation set end to end (sharded extraction + descriptor all-gather + retrieval + Recall@N)")
    return p.parse_args()


def ev_time_ms(fn, iters=20, warm=3):
    """Average duration of fn() in ms with HIP events on the current torch stream (the stream the kernels run on)."""
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def grouping_roofline():
    """K5 (the graded gather) at SURVEY.md section 8d's micro-benchmark shape and at the in-model SA1 shape."""
    from patchaugnet_amd import _lib
    out = {}
    for tag, (b, c, n, m, k) in {"micro_c64": (4096, 64, 1024, 128, 20), "micro_c3": (4096, 3, 4096, 1024, 20),
                                 "model_sa1_b32": (32, 64, 1024, 128, 20)}.items():
        pts = torch.randn(b, c, n, device="cuda")
        idx = torch.randint(0, n, (b, m, k), device="cuda", dtype=torch.int32)
        o = torch.empty(b, c, m, k, device="cuda")
        fn = lambda: _lib.call("pa_grouping_forward", b, c, n, m, k, _lib.ptr(pts), _lib.ptr(idx), _lib.ptr(o))
        ms = ev_time_ms(fn, iters=10 if b > 100 else 50)
        alg = 4.0 * (c * n + m * k + c * m * k) * b
        out[tag] = {"shape": [b, c, n, m, k], "ms": ms, "algorithmic_bytes": alg, "GBps": alg / ms / 1e6}
        if alg < 200e6:      # the working set fits the 256 MiB Infinity Cache (and is re-read every iteration): a cache figure, NOT an HBM one
            out[tag]["note"] = "cache-resident working set (< 256 MiB Infinity Cache): not an HBM bandwidth figure"
        del pts, idx, o
    return out


# The sampling round's latency model (latency-bound kernels report achieved time against a stated model).
# fps_reg_kernel<256,16> is one workgroup (4 wavefronts, one per SIMD) per cloud running m - 1 strictly serial rounds; a round of the shipped ISA
# (hipcc --cuda-device-only -S csrc/fps.hip) is 3 v_xor + 3 v_mov (the selected point as negated register pairs: no operand modifiers on packed fp32,
# csrc/pa_common.h) + 128 VALU instructions for the 16 points of a lane (8 pair blocks: 3 v_pk_add + 3 v_pk_mul + 2 v_pk_add,
# 2 v_min, 2 x (v_cmp_gt_u64 + 2 v_cndmask)), the wave maximum (6 DPP steps + v_readlane), and the selection tail (ds_max_u64 -> s_waitcnt ->
# s_barrier -> ds_read_b64 -> ds_read_b96 of the winner's coordinates).  tools/probes/fps_model.hip times each of those instruction blocks ALONE on
# the same occupancy (one wave per SIMD, s_memtime) and the three blocks chained as one dependent stream; profiles/r05_fps_model.txt is its output
# on the MI355X: pair block 88.1 cycles (x 8), wave maximum 140.0, tail 333.0; chained 1030 cycles per round (the tail's LDS round trips overlap the
# head of the next round's VALU block; the blocks' plain sum is 1178).  Model = the chained figure at the clock the launch holds.
FPS_MODEL = {"pair_block_cycles": 88.1, "pair_blocks_per_round": 8, "wave_max_cycles": 140.0, "tail_cycles": 333.0, "chained_round_cycles": 1030.2,
             "clock_ghz": 2.36, "source": "tools/probes/fps_model.hip -> profiles/r05_fps_model.txt (s_memtime, one wave per SIMD, 32 workgroups)"}


def fps_latency_roofline(batch, points, m):
    """roofline_latency of the longest kernel of a step: measured us per sampling round (HIP events on the launch stream around 10 launches)
    against the model above."""
    from patchaugnet_amd import _lib
    from patchaugnet_amd.weights import synthetic_submaps
    xyz = synthetic_submaps(batch, points, seed=5).squeeze(1).cuda().contiguous()
    idx = torch.empty(batch, m, dtype=torch.int32, device="cuda")
    q = torch.empty(batch, m, 3, device="cuda")
    ms = ev_time_ms(lambda: _lib.call("pa_furthestsampling_gather", batch, points, m, _lib.ptr(xyz), _lib.ptr(idx), _lib.ptr(q)), iters=10)
    rounds = m - 1
    us_round = ms * 1e3 / rounds
    model_us = FPS_MODEL["chained_round_cycles"] / (FPS_MODEL["clock_ghz"] * 1e3)
    return {"kernel": "fps_reg_kernel<512,8> (pa_furthestsampling_gather: first level, one workgroup of 8 wavefronts per cloud -- two per SIMD, eight points per lane -- %d CUs of 256 busy; "
                      "the model is the 256-thread round's instruction chain: the same blocks, a lane's 16 points split over the two waves of its SIMD)" % min(batch, 256),
            "bound": "latency (serial rounds; neither HBM nor MFMA)", "rounds": rounds, "ms_per_launch": ms, "us_per_round": us_round,
            "model_us_per_round": model_us, "frac": model_us / us_round,
            "model": "cycles of the round's three instruction blocks measured alone at the kernel's occupancy and chained as one dependent stream "
                     "(3 v_xor + 3 v_mov + 8 x 16-instruction pair block + 6-step DPP wave maximum + ds_max_u64 / barrier / two LDS reads), / the clock the launch holds",
            "model_terms": FPS_MODEL,
            "counters": "profiles/r05_fps_pmc_sq.txt (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_INSTS_VALU / SQ_INSTS_LDS of the same kernel)"}


def stage_pass(model, x, iters=5):
    """Per-stage device time of one step (HIP events between stages) for the roofline attribution."""
    from patchaugnet_amd import profiling
    with torch.no_grad():
        return profiling.stage_times(model, x, iters)


def profiling_fp0_launch_ms(model, x):
    from patchaugnet_amd import profiling
    try:
        return profiling.fp0_launch_time_ms(model, x)
    except Exception:
        return None


def dominant_kernel_clock(model, x):
    """Shader clock under the dominant kernel: the kernel's debug stamps (csrc/fpx_f32.hip FX_STAMP: s_memtime at a tile's start and end, the constant
    100 MHz s_memrealtime beside them; written only when pa_chain_debug_buffer hands the launch a buffer) over the first 512 wave tiles of one launch
    inside a real forward.  The nominal fp32 MFMA peak is 2.4 GHz; what the chip holds under this kernel is what the kernel can be measured against."""
    import ctypes
    import numpy as np
    from patchaugnet_amd import _lib
    lib = _lib.lib()
    lib.pa_chain_debug_buffer.argtypes, lib.pa_chain_debug_buffer.restype = [ctypes.c_void_p], None
    chain = model._engine.fp[0]
    buf = torch.zeros(512 * 8, dtype=torch.int64, device=x.device)
    origs = {}
    for nm in ("fp_premul", "fp"):
        def wrapped(*a, _o=getattr(chain, nm), **k):
            lib.pa_chain_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
            try:
                return _o(*a, **k)
            finally:
                lib.pa_chain_debug_buffer(None)
        origs[nm] = getattr(chain, nm)
        setattr(chain, nm, wrapped)
    try:
        with torch.no_grad():
            model(x, return_feat=False)
        torch.cuda.synchronize()
    finally:
        for nm, o in origs.items():
            setattr(chain, nm, o)
    t = buf.view(512, 8).cpu().numpy()
    t = t[(t[:, 0] > 0) & (t[:, 6] > t[:, 5])]
    if len(t) < 64:
        return None
    mhz = (t[:, 4] - t[:, 0]) / ((t[:, 6] - t[:, 5]) / 100.0)
    return {"mhz": float(np.median(mhz)), "p10": float(np.percentile(mhz, 10)), "p90": float(np.percentile(mhz, 90)), "tiles": int(len(t)),
            "tile_cycles": float(np.median(t[:, 4] - t[:, 0])), "tile_us": float(np.median((t[:, 6] - t[:, 5]) / 100.0))}


def dominant_kernel_roofline(st, cfg, batch, points, grouping, pmc=None, pmc_note=None):
    """Roofline of the kernel that owns the most CU-time in a step.

    Stage times come from HIP events on the launch stream (patchaugnet_amd/profiling.py).  CU-time = duration x share of the
    256 CUs the launch can occupy: the FPS launches are 1 workgroup per cloud (batch/256 of the chip) and are latency-bound,
    every other launch fills the chip.  In round 1 the owner is the finest feature-propagation chain (fp0: 259->256->256->256
    on batch*points rows), an MFMA-bound kernel; its algorithmic FLOPs are 2*rows*sum(K_l*N_l) with the TRUE K (no padding)."""
    if "fp0.chain" not in st:
        mc = grouping["micro_c64"]
        return {"roofline": {"kernel": "group_lds_kernel (pa_grouping_forward)", "bound": "hbm", "achieved": mc["GBps"], "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": mc["GBps"] / HBM_PEAK_GBS, "traffic": None}}
    fs = cfg["FEATURE_SIZE"]
    dims = [fs[1] + (3 if cfg["USE_ORIGIN_PC_IN_FP"] else 0), 256, 256, fs[0]]
    rows = batch * points
    premul = "fp0.premul" in st
    if premul:   # first layer folded into the prologue (pa_fp_chain_premul): the launch runs layers 2.. on MFMA + 2*3*256 VALU FLOPs per row
        flops = 2.0 * rows * (sum(k * n for k, n in zip(dims[1:-1], dims[2:])) + (dims[0] - fs[1]) * dims[1])
        kname = "fpx32_kernel<1,4,3,2> (pa_fp_chain_premul, csrc/fpx_f32.hip, fp0: 3-NN interpolation of the pre-multiplied coarse features + xyz term, then 256->256->256 on MFMA in half-K passes)"
    else:
        flops = 2.0 * rows * sum(k * n for k, n in zip(dims[:-1], dims[1:]))
        kname = "chain_kernel<2,16,FP,0,1> (pa_mlp_chain, fp0: 3-NN interpolate + 259->256->256->256 shared MLP)"
    ms = st["fp0.chain"]
    tf = flops / (ms * 1e-3) / 1e12
    cu_time = {k: v * (min(batch, 256) / 256.0 if k.endswith(".fps") else 1.0) for k, v in st.items() if k != "total"}
    owner = max(cu_time, key=cu_time.get)
    mc = grouping["micro_c64"]
    traffic = pmc["dominant"]["bytes_per_launch"] if pmc and "dominant" in pmc else None     # measured in this run, or null
    g_traffic = pmc["grouping"]["bytes_per_launch"] if pmc and "grouping" in pmc else None
    return {
        "roofline": {"kernel": kname,
                     "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                     "traffic": traffic, "traffic_source": pmc_note, "algorithmic_flops_per_launch": flops, "ms_per_launch": ms, "cu_time_owner": owner},
        "roofline_grouping": {"kernel": "group_lds_kernel (pa_grouping_forward, K5)", "bound": "hbm", "achieved": mc["GBps"], "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": mc["GBps"] / HBM_PEAK_GBS,
                              "traffic": g_traffic, "shape": mc["shape"],
                              "algorithmic_bytes_per_launch": mc["algorithmic_bytes"], "ms_per_launch": mc["ms"]},
    }


TRAIN_DOMINANT_KERNEL_RE = r"tgemm_cm_kernel<8, 1, 2, 1, 2, 12>"     # pa_tgemm_nn at 18 x (256 x 4096 x 256), forward form (csrc/train_gemm_cm.hip)
DOMINANT_KERNEL_RE = r"fpx32_kernel<|chain_kernel<[12], 16, 3, false, 1[,>]"     # fp0 feature-propagation chain (pa_fp_chain_premul): csrc/fpx_f32.hip, or the 16-row tile kernel (PA_CHAIN_NO_FPX32)
GROUPING_KERNEL_RE = r"group_lds_kernel<4>"


def measure_traffic(batch, points, target=None, patterns=None):
    """HBM bytes per launch of the dominant kernel and of the graded gather, measured NOW: two rocprofv3 passes (FETCH_SIZE and
    WRITE_SIZE cannot share a pass on gfx950: MI355X_MICROARCH.md, PMC slots) over tools/pmc_target.py, which runs the same engine
    steps at the same shape in a child process.  Correction per the guide's HBM section: the counters are KiB; FETCH_SIZE reports
    half the bytes of a wide (16 B/lane) coalesced read, which is how both kernels read, so it is doubled; WRITE_SIZE as reported.
    Returns ({...}, note); any failure gives (None, reason) -- the bench line then carries traffic = null, never a stale constant."""
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "bench.py itself runs under rocprofv3: nested counter passes skipped"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = {}
    tmp = tempfile.mkdtemp(prefix="pa_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable] + (
                target or [os.path.join(ROOT, "tools", "pmc_target.py"), "--batch", str(batch), "--points", str(points)])
            res = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("results.db")]
            if res.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {counter} failed (rc {res.returncode}): {res.stdout[-300:]}"
            c = sqlite3.connect(dbs[0])
            for name, avg, n in c.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
                for tag, rx in (patterns or (("dominant", DOMINANT_KERNEL_RE), ("grouping", GROUPING_KERNEL_RE))):
                    if re.search(rx, name):
                        out.setdefault(tag, {})[counter] = avg * 1024.0
                        out[tag]["launches_sampled"] = n
    except Exception as ex:
        return None, f"PMC pass failed: {ex!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for tag, v in out.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            res[tag] = {"bytes_per_launch": 2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"], "FETCH_SIZE_bytes_raw": v["FETCH_SIZE"], "WRITE_SIZE_bytes": v["WRITE_SIZE"],
                        "launches_sampled": v["launches_sampled"]}
    return (res or None), ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/%s in this run; FETCH_SIZE x2 (gfx950, 16-B reads), "
                           "KiB -> bytes" % (os.path.basename(target[0]) if target else "pmc_target.py"))


def stage_algorithmic_flops(model, batch, points):
    """Algorithmic FLOPs of ONE extraction step PER STAGE, as the engine runs it (post-fold: the first layer of a feature-propagation level applied
    to the known points before interpolation, pa_fp_chain_premul), true K (no padding): 2 * rows * K * N per dense layer of every set-abstraction /
    feature-propagation chain (+ the grouped self-attention of PPT-Net: q/k/v and output projections and the two N x N contractions), the NetVLAD
    assignment and aggregation GEMMs of every scale, the APFA attention logits and the final FC / gating.  Sampling, neighbour search, interpolation
    weights and soft-max are not counted (VALU work).  Keys: the engine's stage names (sa<i>.chain, sa<i>.attn, fp<j>.premul, fp<j>.chain, afa) and
    one entry per NetVLAD scale (vlad.k<K>).  SURVEY.md section 8(d) per-unit figures x the units one launch processes."""
    from patchaugnet_amd.engine import engine_for
    eng = engine_for(model, "cuda")
    L, nfp = len(eng.sa), len(eng.fp)
    npts = [points] + list(eng.sampling[:L])
    out = {}
    for i, ch in enumerate(eng.sa):
        rows = batch * npts[i + 1] * eng.knn[i]
        out[f"sa{i}.chain"] = 2.0 * rows * sum(l[2] * l[4] for l in ch.layers)
        if eng.attn[i] is not None:
            c, n = eng.attn[i].c, npts[i + 1]
            out[f"sa{i}.attn"] = 2.0 * batch * n * (c * 2 * c + c * c) + 2.0 * batch * n * n * c * 3      # projections; energy twice (statistics pass + apply pass) + V p
    off = L - nfp
    for j, ch in enumerate(eng.fp):
        n_u, m_k = npts[j + off], npts[j + off + 1]
        c2 = eng.fp[j + 1].n_last if j + 1 < nfp else eng.sa[-1].n_last
        c1 = (3 if eng.use_origin else 0) if j == 0 else eng.sa[j - 1].n_last
        folded = eng._fold_static[j] and (c1 <= 4 or n_u >= 512) and n_u >= 2 * m_k
        l0 = ch.layers[0]
        rest = sum(l[2] * l[4] for l in ch.layers[1:])
        if folded:
            out[f"fp{j}.premul"] = 2.0 * batch * m_k * c2 * l0[4]
            out[f"fp{j}.chain"] = 2.0 * batch * n_u * (c1 * l0[4] + rest)
        else:
            out[f"fp{j}.chain"] = 2.0 * batch * n_u * (l0[2] * l0[4] + rest)
    ktot = 0
    for v in eng.vlads:
        out[f"vlad.k{v.k}"] = out.get(f"vlad.k{v.k}", 0.0) + 2.0 * batch * v.n * v.c * v.k * 2
        ktot += v.k
    head = 0.0
    if eng.head_kind == "afa":
        head += 2.0 * batch * ktot * 256 * 256 + 2.0 * batch * ktot * 256 * eng.afa.nout
    elif eng.head_kind == "fc":
        head += 2.0 * batch * ktot * 256 * eng.head.nout
    if eng.gate is not None:
        head += 2.0 * batch * eng.gate.dim * eng.gate.dim
    out["afa"] = head
    return out


def step_algorithmic_flops(model, batch, points):
    """Sum of stage_algorithmic_flops: the dense-layer FLOPs of one extraction step."""
    return sum(stage_algorithmic_flops(model, batch, points).values())


# Kernel -> stage table of the HEADLINE configuration (PatchAugNet, batch 32, 4096 points, f32): (stage, regex on the kernel name, workgroups in x or None).
# The trace pass below reads every kernel's average duration out of rocprofv3's kernel trace of this very script; a row whose regex matches nothing
# (a kernel was renamed / re-tiled) is reported as unmatched, never silently dropped.
HEADLINE_KERNELS = [
    ("fp0.chain", r"fpx32_kernel<1, 4, 3, 2>", None, "mfma"),
    ("fp0.premul", r"linear_lds_kernel<64, 8>", None, "mfma"),
    ("vlad.k64", r"vlad_accum_kernel<4>", None, "mfma"),
    ("fp1.chain", r"chain_kernel<2, 8, 2, false, 4, false>", None, "mfma"),
    ("sa2.chain", r"chain_kernel<1, 8, 1, false, 4, true>", None, "mfma"),
    ("sa0.chain", r"sa_tiny_reg_kernel<5, 8, false>", None, "mfma"),
    ("sa1.chain", r"chain_kernel<5, 2, 1, true, 4, false>", None, "mfma"),
    ("fp2.chain", r"chain_kernel<1, 8, 2, false, 4, false>", None, "mfma"),
    ("fp1.premul", r"chain_kernel<1, 8, 0, false, 4, false>", None, "mfma"),
    ("vlad.k16", r"vlad_accum_kernel<1>", 16, "mfma"),
    ("vlad.k4", r"vlad_accum_kernel<1>", 2, "mfma"),
    ("afa", r"afa_cluster_kernel|afa_combine_kernel", None, "mfma"),
    ("vlad.finalize", r"vlad_finalize_multi_kernel", None, "latency"),
    ("sa0.fps", r"fps_reg_kernel<(512, 8|256, 16)", None, "latency"),
    ("sa1.fps", r"fps_reg_kernel<256, 4", None, "latency"),
    ("sa2.fps", r"fps_reg_kernel<64, 2", None, "latency"),
    ("sa0.knn", r"knn_quad_kernel", 8, "valu"),
    ("sa1.knn", r"knn_quad_kernel", 1, "valu"),
    ("sa2.knn", r"knn_wave_kernel", None, "valu"),
    ("fp0.3nn", r"three_nn_grid_kernel", None, "valu"),
    ("fp1.3nn", r"three_nn_kernel", 4, "valu"),
    ("fp2.3nn", r"three_nn_kernel", 1, "valu"),
]


def trace_kernel_times(a, streams):
    """Average duration of every kernel of the headline step as rocprofv3's --kernel-trace sees it, measured NOW: this script re-run as a child
    (same batch / points / model, --streams `streams`, the driver's 20 + 5 steps, no extras) under `rocprofv3 --kernel-trace`, the database read
    back.  Only launches of the steady state (the middle half of the steps, by the first-level sampling launches) count.  Returns
    ({(kernel name, workgroups_x, grid_y): (calls per step, avg_us)}, steps, note) or (None, 0, reason)."""
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, 0, "bench.py itself runs under rocprofv3: nested trace passes skipped"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, 0, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="pa_trace_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "-d", tmp, "-o", "trace", "--", sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "40", "--warmup", "8",
               "--batch", str(a.batch), "--points", str(a.points), "--streams", str(streams), "--reps", "2", "--no-cpu-baseline", "--no-kernel-pass", "--no-pmc",
               "--no-extras", "--no-trace", "--only-steps"]
        res = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        dbs = [os.path.join(r, f) for r, _, fs in os.walk(tmp) for f in fs if f.endswith("results.db")]
        if res.returncode != 0 or not dbs:
            return None, 0, f"rocprofv3 --kernel-trace failed (rc {res.returncode}): {res.stdout[-300:]}"
        c = sqlite3.connect(dbs[0])
        view = [n for n, in c.execute("select name from sqlite_master where type='view' and name like 'kernels%'")][-1]
        cols = [r[1] for r in c.execute(f"pragma table_info('{view}')")]
        gx = "grid_x" if "grid_x" in cols else "grid_size_x"
        gy = "grid_y" if "grid_y" in cols else "grid_size_y"
        wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
        rows = list(c.execute(f"select name, {gx}, {gy}, {wx}, start, end from {view} order by start"))
        marks = [i for i, r in enumerate(rows) if re.search(DOMINANT_KERNEL_RE, r[0])]      # one launch of the finest FP level per step (the child runs the headline shape only)
        if len(marks) < 8:
            return None, 0, "trace holds fewer than 8 steps"
        lo, hi = marks[len(marks) // 4], marks[3 * len(marks) // 4]
        steps = sum(1 for i in marks if lo <= i < hi)
        agg = {}
        for name, x, y, w, s0, e0 in rows[lo:hi]:
            k = (name, x // max(w, 1), y)
            v = agg.setdefault(k, [0, 0.0])
            v[0] += 1
            v[1] += (e0 - s0) / 1e3
        wall_us = (rows[hi][4] - rows[lo][4]) / 1e3 / steps
        return {k: (n / steps, t / n) for k, (n, t) in agg.items()}, steps, f"rocprofv3 --kernel-trace of `bench.py --streams {streams} --steps 40` in this run, {steps} steady-state steps, {wall_us:.1f} us per step under the tracer"
    except Exception as ex:
        return None, 0, f"trace pass failed: {ex!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def stage_rooflines(a, model, ms_per_step):
    """One roofline row per kernel of the headline step, from rocprofv3's kernel trace of this run: the kernel ALONE on the chip (one stream) and
    INSIDE the four-stream pipeline the headline value is measured in.  MFMA rows: algorithmic FLOPs (stage_algorithmic_flops) / average launch
    duration / 157.3 TFLOP/s; the sampling / search kernels are VALU- or latency-bound and carry their durations only (roofline_latency holds the
    sampling round's model).  share_of_step = calls x one-stream duration / sum over all kernels."""
    import re
    t1, n1, note1 = trace_kernel_times(a, 1)
    t4, n4, note4 = trace_kernel_times(a, a.streams) if a.streams > 1 else (None, 0, "one stream requested")
    if t1 is None:
        return {"error": note1}
    fl = stage_algorithmic_flops(model, a.batch, a.points)
    total1 = sum(c * us for c, us in t1.values())
    total4 = sum(c * us for c, us in t4.values()) if t4 else None

    def pick(table, rx, wgx):
        hits = [(k, v) for k, v in table.items() if re.search(rx, k[0]) and (wgx is None or k[1] == wgx)]
        if not hits:
            return None
        calls = sum(v[0] for _, v in hits)
        return calls, sum(v[0] * v[1] for _, v in hits) / calls          # launches per step, average us per launch
    rows, seen = [], 0.0
    for stage, rx, wgx, bound in HEADLINE_KERNELS:
        h1 = pick(t1, rx, wgx)
        if h1 is None:
            rows.append({"stage": stage, "kernel": rx, "error": "no kernel of the trace matches (renamed / re-tiled?)"})
            continue
        calls, us1 = h1
        h4 = pick(t4, rx, wgx) if t4 else None
        row = {"stage": stage, "kernel": rx, "bound": bound, "launches_per_step": round(calls, 2), "us_per_launch": us1,
               "us_per_launch_in_pipeline": h4[1] if h4 else None, "share_of_step_kernel_time": calls * us1 / total1}
        seen += calls * us1
        if bound == "mfma" and stage in fl:
            f = fl[stage] / max(round(calls), 1)
            row.update({"algorithmic_flops_per_launch": f, "achieved": f / us1 / 1e6, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": f / us1 / 1e6 / MFMA_F32_PEAK_TFLOPS, "frac_in_pipeline": (f / h4[1] / 1e6 / MFMA_F32_PEAK_TFLOPS) if h4 else None})
        rows.append(row)
    return {"rows": rows, "kernel_time_per_step_us": total1, "kernel_time_per_step_in_pipeline_us": total4,
            "cross_stream_inflation": (total4 / total1) if total4 else None, "covered_share_of_kernel_time": seen / total1,
            "source": note1, "source_in_pipeline": note4,
            "note": "durations = rocprofv3 kernel-trace averages over the steady state (not HIP events); frac = the kernel alone on the chip, frac_in_pipeline = "
                    "the same kernel while the other streams' kernels share the CUs (how the headline value is measured); a stage below 0.65 has its "
                    "counter-backed bound in DESIGN.md section 5 (profiles/r06_pmc_*)"}


def reference_protocol(model, points, batch=100, nbatches=12):
    """The reference's own (and only) self-reported timing: SceneDataSet.make_descs with stat_time (datasets/scene_dataset.py:666-686, 710-711;
    place_recognition/evaluate.py:170 runs it at batch 100): per batch, synchronize, start the clock, `model(feed)` on a batch already on the
    device, descriptors `.detach().cpu().numpy()`, synchronize, stop; the batch time divided by the batch size is every submap's run time; it
    prints mean +- std in ms.  One stream, one batch at a time, the model API as evaluate.py calls it (model(x) -> (desc, fp_features, centre idx))."""
    import numpy as np
    from patchaugnet_amd.weights import synthetic_submaps
    feeds = [synthetic_submaps(batch, points, seed=900 + i).cuda() for i in range(3)]
    per = []
    with torch.no_grad():
        for i in range(2):
            out = model(feeds[i % 3])
            (out[0] if isinstance(out, tuple) else out).detach().cpu().numpy()
        for i in range(nbatches):
            feed = feeds[i % 3]
            torch.cuda.synchronize()
            t0 = time.time()
            g = model(feed)
            if isinstance(g, tuple):
                g = g[0]
            g = g.detach().cpu().numpy()
            g = np.squeeze(g).reshape([-1, g.shape[-1]])
            torch.cuda.synchronize()
            per += [(time.time() - t0) * 1000 / batch] * batch
    return {"run_time_ms_per_submap_mean": float(np.mean(per)), "run_time_ms_per_submap_std": float(np.std(per)), "batch": batch, "batches": nbatches,
            "submaps_per_s": 1000.0 / float(np.mean(per)),
            "protocol": "datasets/scene_dataset.py:666-686 (stat_time): synchronize, model(feed) on a device-resident batch + descriptors .cpu().numpy(), "
                        "synchronize; (batch time / batch size) per submap; mean +- std as the reference prints it; evaluate.py:170 batch size"}


def cpu_baseline(cfg, sd, batch, points):
    """The CPU oracle (a port) on this host's cores: same model, same weights, bounded sample."""
    from oracle import models_cpu
    from patchaugnet_amd.weights import synthetic_submaps
    from patchaugnet_amd.hostcpu import limit_host_threads
    cores = limit_host_threads()             # the container's CPU grant (cgroup quota), not the machine's core count
    x = synthetic_submaps(batch, points, seed=1234)
    done, dt = 0, 0.0
    with torch.no_grad():
        models_cpu.patch_aug_net_forward(sd, cfg, x[:1])            # warm-up (page-in, OpenMP pool)
        t0 = time.perf_counter()
        while dt < 10.0 and done < 64 * batch:                      # bounded sample: about 10 s of CPU work
            models_cpu.patch_aug_net_forward(sd, cfg, x)
            done += batch
            dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "submaps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/models_cpu.patch_aug_net_forward, {done // batch} batches of {batch} x {points}-pt synthetic submaps, {dt:.2f} s"}


def pcie_inclusive(model, a, pipe):
    """The same steps with the batch handed over in (pinned) HOST memory and the descriptors returned to the host, like
    SceneDataSet.make_descs does (datasets/scene_dataset.py:667-683): H2D of the (B,1,N,3) fp32 batch and D2H of the (B,256)
    descriptors are inside the timed region, issued on the step's own stream so they overlap other streams' kernels."""
    from patchaugnet_amd.weights import synthetic_submaps
    nbuf = max(a.streams, 1) * 2
    host_x = [synthetic_submaps(a.batch, a.points, seed=77 + i).pin_memory() for i in range(nbuf)]
    host_d = torch.empty(a.steps, a.batch, 256).pin_memory()

    graphed = hasattr(pipe, "run")          # GraphedExtractor: the pinned batch is copied straight into the slot's static input buffer
    if graphed:                             # slots with their OWN staging buffers (the headline's extractor reads the resident batch in place)
        from patchaugnet_amd.extract import GraphedExtractor
        with torch.no_grad():
            pipe = GraphedExtractor(model, (a.batch, 1, a.points, 3), a.streams)

    def one(i):
        xd = host_x[(i if i >= 0 else -1 - i) % nbuf].to("cuda", non_blocking=True)
        d = model(xd, return_feat=False)
        if i >= 0:
            host_d[i].copy_(d, non_blocking=True)

    def submit(i):
        if graphed:
            pipe.run(host_x[(i if i >= 0 else -1 - i) % nbuf], out=host_d[i] if i >= 0 else None)
        else:
            pipe.submit(one, i)

    rates = []
    with torch.no_grad():
        pipe.begin()
        for i in range(nbuf):
            submit(-1 - i)
        pipe.end()
        for rep in range(4):    # repetition 0 is warm-up (first D2H into the fresh pinned result pages); then the median of 3
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.begin()
            for i in range(a.steps):
                submit(i)
            pipe.end()
            torch.cuda.synchronize()
            if rep > 0:
                rates.append(a.steps * a.batch / (time.perf_counter() - t0))
    med = sorted(rates)[1]
    return {"value": med, "unit": "submaps/s", "ms_per_step": a.batch / med * 1e3, "repetitions": [round(r, 1) for r in rates],
            "note": "host pinned fp32 batch -> H2D -> extraction -> D2H descriptors, all inside the timed region; median of 3 repetitions of "
                    "the K steps after one warm-up repetition (never the headline value)"}


def train_bench(a, emit=True, pmc=None):
    """BASELINE.json configs[3]: one training step per bench step -- the reference's native tuple of 18 clouds (1 query + 2 positives +
    14 negatives + 1 other negative, configs/patch_aug_net.yaml:60-62; BASELINE.json says batch=16, the reference's loader only makes
    18), nn_dict with 2 (query, positive) pairs => 3 related clouds through the decoder and the patch Chamfer loss, quadruplet loss,
    backward, Adam step (train_place_recognition.py:142-169, :255-392).  The dense path runs on csrc/train_gemm.hip."""
    from patchaugnet_amd import configs, patch_aug_net, train_ops
    from patchaugnet_amd.train import training_step
    from patchaugnet_amd.weights import seeded_state_dict
    cfg = configs.patch_aug_net_config()
    model = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    model.load_state_dict(seeded_state_dict(model.state_dict()))
    model = model.cuda()
    g = torch.Generator().manual_seed(5)
    n = a.points
    q, pos, neg, oth = (torch.rand(1, k, n, 3, generator=g) * 2 - 1 for k in (1, 2, 14, 1))
    nn_dict = {(0, 1): torch.randint(0, n, (1024, 1), generator=g).numpy(), (0, 2): torch.randint(0, n, (1024, 1), generator=g).numpy()}
    graphed = not a.no_graphs
    # the training loop's optimizer (train_place_recognition.py:386-392: torch.optim.Adam) on the HIP kernel of csrc/adam.hip: same update rule and
    # state_dict layout, the tensor list in the kernel arguments (capturable); --torch-adam keeps torch's fused multi-tensor kernels for A/B
    from patchaugnet_amd.optim import Adam as HipAdam
    make_opt = (lambda: torch.optim.Adam(model.parameters(), lr=1e-5, capturable=graphed, fused=graphed)) if a.torch_adam else \
               (lambda: HipAdam(model.parameters(), lr=1e-5))
    opt = make_opt()
    q, pos, neg, oth = (t.cuda() for t in (q, pos, neg, oth))           # inputs resident in HBM before the clock starts
    if graphed:     # forward + losses + backward + Adam captured once (train.GraphedTrainer), one replay per step
        from patchaugnet_amd.train import GraphedTrainer
        try:
            # prefetch: the coordinate-only launches (sampling, neighbour search, 3-NN) of the NEXT batch run on a side stream under the
            # current step (a data loader knows the next batch); here the next batch is the same synthetic tuple, recomputed every step
            trainer = GraphedTrainer(model, opt, q, pos, neg, oth, nn_dict, num_points=n, prefetch=not a.no_prefetch)
            nb = None if a.no_prefetch else (q, pos, neg, oth)
            step = lambda: trainer.step(q, pos, neg, oth, next_batch=nb)
        except Exception as ex:
            print(f"bench.py: hipGraph capture of the training step failed ({ex!r}); eager launches instead", file=sys.stderr)
            graphed = False
            opt = make_opt()
    if not graphed:
        step = lambda: training_step(model, opt, q, pos, neg, oth, nn_dict=nn_dict, num_points=n)
    for _ in range(max(a.warmup, 2)):
        losses = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = {k: float(v) for k, v in losses.items()}
    if graphed:
        trainer.close()
    clouds = 18
    # dominant dense kernel: the 256 -> 256 layers of the finest feature-propagation level, forward form (BatchNorm + ReLU of the previous
    # layer in the operand loader, statistics in the epilogue), in isolation on the launch stream
    B, M, N, K = clouds, 256, n, 256
    W = torch.randn(M, K, device="cuda")
    X = torch.randn(B, K, N, device="cuda")
    Y = torch.empty(B, M, N, device="cuda")
    pblk = torch.rand(7, K, device="cuda")
    stats = torch.zeros(train_ops.STAT_SLOTS, 2, M, dtype=torch.float64, device="cuda")
    ms = ev_time_ms(lambda: train_ops.tgemm_nn(B, M, N, K, W, 0, K, True, X, K * N, N, Y, M * N, N, bmode=1, bp=pblk, stats=stats), iters=20)
    flops = 2.0 * B * M * N * K
    tf = flops / (ms * 1e-3) / 1e12
    traffic = tnote = None
    if (emit if pmc is None else pmc) and not a.no_pmc:      # HBM bytes of that launch: FETCH_SIZE / WRITE_SIZE passes over the same call in a child process (tools/tgemm_target.py)
        pm, tnote = measure_traffic(0, 0, target=[os.path.join(ROOT, "tools", "tgemm_target.py"), "6", str(clouds)],
                                    patterns=(("dominant", TRAIN_DOMINANT_KERNEL_RE),))
        traffic = pm["dominant"]["bytes_per_launch"] if pm and "dominant" in pm else None
    # the whole step against the fp32 MFMA peak: forward dense FLOPs of the 18 clouds (the same per-stage count as the extraction
    # line, stage_algorithmic_flops on an eval() copy: set-abstraction / feature-propagation chains, NetVLAD, head) + the decoder over the related
    # clouds' patch features (256 -> 1024 -> 1024 -> 3 k per patch, pointnet_autoencoder.py:85-111), times 3 (input-gradient and weight-gradient
    # contractions of the backward pass each repeat the forward's FLOPs; sampling, searches, losses, BatchNorm and Adam are VALU / memory work)
    step_frac = None
    try:
        import copy
        ev = copy.deepcopy(model).eval()
        fwd = sum(stage_algorithmic_flops(ev, clouds, n).values())
        dec = model.decoder
        related = len({i for pair in nn_dict for i in pair})
        patches = cfg["SAMPLING"][0]
        dec_fl = 2.0 * related * patches * (dec.fc1.in_features * dec.fc1.out_features + dec.fc2.in_features * dec.fc2.out_features + dec.fc3.in_features * dec.fc3.out_features)
        del ev
        tot = 3.0 * (fwd + dec_fl)
        step_frac = {"algorithmic_flops_per_step": tot, "forward_flops": fwd + dec_fl, "decoder_forward_flops": dec_fl, "achieved_tflops": tot / (dt / a.steps) / 1e12,
                     "peak_tflops": MFMA_F32_PEAK_TFLOPS, "frac": tot / (dt / a.steps) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                     "note": "3 x (forward dense FLOPs of the 18 clouds + decoder over the related clouds) / ms_per_step / fp32 MFMA peak; the step also holds ~1.2 ms of "
                             "sampling / search / loss / BatchNorm / optimizer kernels that are not matrix work"}
    except Exception as ex:
        step_frac = {"error": repr(ex)}
    line = {
        "metric": "training steps/sec (PatchAugNet quadruplet step, patch Chamfer reconstruction loss)", "value": a.steps / dt, "unit": "steps/s",
        "clouds_per_s": clouds * a.steps / dt, "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PatchAugNet training step, quadruplet tuple of {clouds} x {n}-pt synthetic submaps (1+2+14+1), 3 related clouds "
                               "through the decoder, patch Chamfer + quadruplet loss, backward, Adam (BASELINE.json configs[3]), 1xMI355X",
                   "clouds_per_step": clouds, "points": n, "path": "HIP point ops + HIP training GEMMs (csrc/train_gemm_cm.hip, train_gemm.hip, fp_fold_train.hip), autograd graph in torch",
                   "weights": "key-seeded random init", "parallelism": "dp1",
                   "launch": ("one hipGraph replay per step (forward + losses + backward + Adam" + (" (torch fused)" if a.torch_adam else " (csrc/adam.hip)") + ")" + ("" if a.no_prefetch else
                              "; sampling / neighbour search / 3-NN of the next batch replayed on a side stream under it")) if graphed else "python launches"},
        "step_mfma_frac": step_frac,
        "losses_last_step": losses,
        "losses_note": "place_recognition = 0.0 means the hinge of the quadruplet loss is inactive on this synthetic tuple (random-init descriptors of "
                       "unrelated clouds); the captured graph replays every kernel of the step regardless, so the timing is representative",
        "roofline": {"kernel": "tgemm_cm_kernel<8,1,2,1,2,12> (pa_tgemm_nn on LDS-resident weights, csrc/train_gemm_cm.hip: 256 -> 256 layer of the finest FP "
                               "level, forward: BatchNorm + ReLU of the previous layer on the loaded operand registers, statistics in the epilogue)", "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": tnote,
                     "algorithmic_bytes_per_launch": 4.0 * B * N * (M + K) + 4.0 * M * K, "algorithmic_flops_per_launch": flops, "ms_per_launch": ms},
    }
    if emit:
        print(json.dumps(line))
    return line


def oxford_eval(a, world=1, rank=0):
    """BASELINE.json configs[2] / SURVEY.md section 8(d) config 3: the Oxford-sized evaluation set END TO END -- 2 999 synthetic 4096-pt submaps
    in 23 trips (the reference's evaluation loop: datasets/scene_dataset.py:666-711 make_descs -> :1016-1099 get_recall_precision, driven by
    place_recognition/evaluate.py:167-237) through
        distributed.extract_dataset(graphs=True)   contiguous ceil(n / world) shard per rank, one hipGraph per stream, pinned host batches
        -> ONE all_gather_into_tensor of the (n_r, 256) blocks (RCCL; a no-op at N = 1)
        -> retrieval.get_recall_precision          query-sharded HIP brute-force kNN + host bookkeeping
        -> retrieval.average                        Recall@N / one-percent recall as evaluate.py prints them.
    The SAME function runs at any world size (`bench.py --config oxford --gpus N`, or under torchrun); at N = 1 it also rides in the headline
    line's other_configs.  The clouds and the reference numbers are the committed fixture's (tests/golden/e2e_recall_oxford.npz: the reference's
    SceneDataSet.get_recall_precision on the oracle's descriptors of the same clouds, generated by oracle/gen_e2e_golden.py -- used here as the
    checker of the recall figures and as the seeded input generator, never as the thing timed)."""
    import numpy as np
    from oracle import gen_e2e_golden as g           # seeded cloud generator + trip layout of the committed fixture (checker side)
    from patchaugnet_amd import configs, distributed, patch_aug_net, retrieval
    from patchaugnet_amd.weights import seeded_state_dict
    z = np.load(os.path.join(ROOT, "tests", "golden", "e2e_recall_oxford.npz"))
    sizes = [int(v) for v in z["sizes"]]
    n = sum(sizes)
    model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
    model.load_state_dict(seeded_state_dict(model.state_dict()))
    model = model.cuda().eval()
    lo, hi = distributed.shard_bounds(n, rank, world)
    t0 = time.perf_counter()
    host = g.clouds(lo, hi, g.OX_SEED, g.OX_SIZES, g.NUM_POINTS, g.OX_PLACES).pin_memory() if hi > lo else torch.empty(0, 1, g.NUM_POINTS, 3)
    t_gen = time.perf_counter() - t0
    resident = host.cuda()
    dist, _, _ = distributed.dist_info()

    def timed_extract(load):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = distributed.extract_dataset(model, load, n, batch_size=a.batch, n_streams=a.streams, graphs=not a.no_graphs)
        torch.cuda.synchronize()
        return d, time.perf_counter() - t0

    with torch.no_grad():
        timed_extract(lambda b0, b1: resident[b0 - lo:b1 - lo])                     # warm-up: engine build, graph capture, allocator pools
        desc, t_res = timed_extract(lambda b0, b1: resident[b0 - lo:b1 - lo])      # inputs resident in HBM (the metric's convention)
        desc_h, t_host = timed_extract(lambda b0, b1: host[b0 - lo:b1 - lo])       # pinned host batches, H2D inside (what make_descs does)
    same = bool(torch.equal(desc, desc_h))
    # the exchange alone (already inside the two extraction times above): one collective over the local blocks
    local = desc[lo:hi].contiguous()
    distributed.all_gather_descriptors(local, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    distributed.all_gather_descriptors(local, n)
    torch.cuda.synchronize()
    ag_ms = (time.perf_counter() - t0) * 1e3 if dist is not None else 0.0
    xy = g.trip_positions(g.OX_SEED, g.OX_SIZES, g.OX_PLACES, g.OX_ROUTE)
    tuples = g.positives(xy, g.OX_SIZES, g.OX_POS_RADIUS)
    top_k = int(z["top_k"])
    knn_s = [0.0, 0]

    def timed_knn(db, q, k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = retrieval.hip_knn(db, q, k)
        torch.cuda.synchronize()
        knn_s[0] += time.perf_counter() - t
        knn_s[1] += 1
        return r
    # the positives as arrays, once per dataset (retrieval.PositiveTable: what loading the reference's pickles is to its evaluation loop); the
    # per-pair dict path (the reference's own data structure walked per query trip / reference trip pair) is timed beside it
    t0 = time.perf_counter()
    table = retrieval.PositiveTable(tuples, sizes)
    t_table = time.perf_counter() - t0
    retrieval.get_recall_precision(desc, sizes, table, top_k=top_k, skip_trip_itself=True)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = retrieval.get_recall_precision(desc, sizes, table, top_k=top_k, skip_trip_itself=True)
    torch.cuda.synchronize()
    t_ret = time.perf_counter() - t0
    retrieval.get_recall_precision(desc, sizes, table, top_k=top_k, skip_trip_itself=True, knn=timed_knn)      # same call, the kNN launches bracketed
    t0 = time.perf_counter()
    res_pp = retrieval.get_recall_precision(desc, sizes, tuples, top_k=top_k, skip_trip_itself=True)
    torch.cuda.synchronize()
    t_ret_pp = time.perf_counter() - t0
    same_res = sorted(res) == sorted(res_pp) and all(all(np.array_equal(u, v) if isinstance(u, np.ndarray) else u == v for u, v in zip(res[k], res_pp[k])) for k in res)
    if dist is not None:
        tt = torch.tensor([t_res, t_host, t_ret, knn_s[0], t_gen, t_ret_pp, t_table], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)                    # a stage takes as long as its slowest rank
        t_res, t_host, t_ret, knn_s[0], t_gen, t_ret_pp, t_table = tt.tolist()
    if rank != 0:
        return None
    ave = retrieval.average(res, top_k)
    ref_rec, ref_opr = z["recall"].astype(np.float64).mean(0), float(z["opr"].mean())
    probe = g.desc_probe(desc.cpu().numpy())
    e2e = t_res + t_ret
    return {
        "metric": "4096-pt submaps/sec, Oxford-sized evaluation set end to end (extraction + descriptor all-gather + retrieval kNN + Recall@N)",
        "value": n / e2e, "unit": "submaps/s", "n_gpus": world, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PatchAugNet Oxford-sized eval set: {n} synthetic 4096-pt submaps in {len(sizes)} trips, {len(res)} trip pairs, data-parallel shard "
                               f"(ceil(n/{world}) contiguous records per rank) + RCCL descriptor all-gather + brute-force kNN retrieval (BASELINE.json configs[2])",
                   "batch_per_gpu": a.batch, "streams": a.streams, "parallelism": f"dp{world}", "submaps": n, "trips": len(sizes), "top_k": top_k},
        "extraction_s": t_res, "extraction_submaps_per_s": n / t_res,
        "extraction_from_pinned_host_s": t_host, "extraction_from_pinned_host_submaps_per_s": n / t_host, "host_and_resident_descriptors_identical": same,
        "all_gather_ms": ag_ms, "retrieval_ms": t_ret * 1e3, "retrieval_knn_ms": knn_s[0] * 1e3, "retrieval_knn_launches": knn_s[1],
        "retrieval_host_bookkeeping_ms": max(t_ret - knn_s[0], 0.0) * 1e3, "retrieval_per_pair_dict_path_ms": t_ret_pp * 1e3,
        "positive_table_build_ms_once_per_dataset": t_table * 1e3, "table_and_dict_paths_identical": bool(same_res),
        "end_to_end_s": e2e, "end_to_end_with_table_build_s": e2e + t_table, "cloud_generation_s_untimed": t_gen,
        "recall_at_1": float(ave[0][0]), "recall_at_5": float(ave[0][4]), "one_percent_recall": float(ave[2]),
        "reference_recall_at_1": float(ref_rec[0]), "reference_recall_at_5": float(ref_rec[4]), "reference_one_percent_recall": ref_opr,
        "recall_delta_pp": {"at_1": float(ave[0][0] - ref_rec[0]), "at_5": float(ave[0][4] - ref_rec[4]), "max_over_N": float(np.abs(ave[0] - ref_rec).max()),
                            "one_percent": float(ave[2] - ref_opr)},
        "descriptor_probe_max_abs_diff_vs_oracle": float(np.abs(probe - z["desc_probe"]).max()),
        "reference": "tests/golden/e2e_recall_oxford.npz: the reference's SceneDataSet.get_recall_precision (datasets/scene_dataset.py:1016-1099) run on the CPU "
                     "oracle's descriptors of the same clouds (oracle/gen_e2e_golden.py); north_star: Recall@1 within 0.1 percentage points",
    }


def extras(a):
    """The other BASELINE.json configurations, measured in the SAME run and carried in the same JSON line (N = 1 only): configs[0] PointNetVLAD
    B = 1 on the host cores, configs[3] the training step (+ its dense kernel's roofline), configs[4] PPT-Net with
    the fp32 and the fp16 MLP path, and the EMD call of the reference's reconstruction loss (16, 4096, 3) at 64 / 1024 iterations.  Each
    entry is independent: a failure is reported in place, never raised."""
    import copy
    import gc
    out = {}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as ex:
            out[name] = {"error": repr(ex)}
        gc.collect()
        torch.cuda.empty_cache()

    def pointnetvlad_cpu():
        from patchaugnet_amd import pointnet_vlad
        from patchaugnet_amd.hostcpu import limit_host_threads
        from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
        cores = limit_host_threads()
        m = pointnet_vlad.PointNetVlad(global_feat=True, feature_transform=True, max_pool=False, output_dim=256, num_points=4096)
        m.load_state_dict(seeded_state_dict(m.state_dict()))
        m.eval()
        x = synthetic_submaps(1, 4096, seed=3)
        with torch.no_grad():
            for _ in range(3):
                m(x)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                m(x)
                ts.append(time.perf_counter() - t0)
        ts.sort()
        return {"ms_per_submap": ts[len(ts) // 2] * 1e3, "min_ms": ts[0] * 1e3, "max_ms": ts[-1] * 1e3, "batch": 1, "cores": cores, "device": "cpu",
                "workload": "PointNetVLAD, 4096 pts, batch = 1, CPU-only PyTorch path (BASELINE.json configs[0]); protocol of scene_dataset.py:672-686, 20 timed forwards, median"}

    def train():
        b = copy.copy(a)
        b.steps, b.warmup, b.no_graphs, b.no_prefetch = 20, 3, False, False
        line = train_bench(b, emit=False, pmc=True)
        return {k: line[k] for k in ("ms_per_step", "value", "unit", "clouds_per_s", "roofline", "step_mfma_frac", "losses_last_step", "losses_note")} | {"workload": line["config"]["workload"]}

    def extract_rate(model_name, mlp_dtype):
        from patchaugnet_amd import configs, patch_aug_net, pptnet
        from patchaugnet_amd.extract import GraphedExtractor
        from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
        model = pptnet.Network(param=configs.pptnet_config(), use_normalize=True) if model_name == "pptnet" else \
            patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
        model.load_state_dict(seeded_state_dict(model.state_dict()))
        model = model.cuda().eval()
        model.mlp_dtype = mlp_dtype
        x = synthetic_submaps(a.batch, a.points, seed=1234).cuda()
        steps = 40
        with torch.no_grad():
            # the headline's pipeline (extract.SampledAheadExtractor: sampling of 16 batches a group ahead) on 40 distinct resident batches
            from patchaugnet_amd.extract import SampledAheadExtractor
            xs = torch.stack([synthetic_submaps(a.batch, a.points, seed=4321 + i) for i in range(steps)]).cuda()
            out = torch.empty(steps, a.batch, 256, device="cuda")
            gx = SampledAheadExtractor(model, tuple(x.shape), a.streams, group=a.group)
            rates = []
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gx.extract(xs, out)
                torch.cuda.synchronize()
                if rep:
                    rates.append(steps * a.batch / (time.perf_counter() - t0))
            del xs, out
        rates.sort()
        res = {"value": rates[1], "unit": "submaps/s", "min": rates[0], "max": rates[-1], "batch": a.batch, "steps": steps,
               "dtype": {"f32": "f32", "f16": "f16 MFMA operands in the shared-MLP chains and the self-attention contractions, fp32 accumulate / soft-max / everything else fp32",
                         "f32x3": "f32 everywhere except the two 256 -> 256 layers of the finest FP level: every product from (hi, lo) fp16 operand pairs (three fp16 MFMAs, "
                                  "~2^-21 relative), fp32 accumulate; OPT-IN (model.mlp_dtype = 'f32x3'), not the headline"}[mlp_dtype]}
        if mlp_dtype == "f32x3":      # the evidence the mode is offered on: its descriptors against the exact-fp32 path's on the same batch
            with torch.no_grad():
                d3 = model(x, return_feat=False)
                d3 = (d3[0] if isinstance(d3, (tuple, list)) else d3).double()
                model.mlp_dtype = "f32"
                d1 = model(x, return_feat=False)
                d1 = (d1[0] if isinstance(d1, (tuple, list)) else d1).double()
                model.mlp_dtype = mlp_dtype
            res["parity_vs_exact_fp32_path"] = {"max_abs_diff_of_l2_normalised_descriptors": (d3 - d1).abs().max().item(),
                                                "min_cosine": torch.nn.functional.cosine_similarity(d3, d1, dim=1).min().item(),
                                                "note": "tests/test_gpu_chain.py::test_finest_fp_level_from_split_fp16_operands_meets_the_fp32_tolerance: max error against "
                                                        "float64 6.5e-7 of the scale (the fp32 MFMA kernel on the same data: 6.1e-7)"}
        try:      # rooflines of the finest feature-propagation chain (the largest dense launch) against BOTH bounds, and the attention's share of the step
            del gx
            with torch.no_grad():
                st = stage_pass(model, x)
            rows = a.batch * a.points
            fl = 2.0 * rows * (256 * 256 * 2 + 3 * 256)
            ms = st.get("fp0.chain")
            if ms:
                peak = MFMA_F32_PEAK_TFLOPS if mlp_dtype == "f32" else MFMA_F16_PEAK_TFLOPS
                if mlp_dtype == "f32x3":
                    fl = 3.0 * 2.0 * rows * (256 * 256 * 2) + 2.0 * rows * 3 * 256      # ISSUED on the fp16 pipe: three MFMAs per product of the two dense layers
                byts = rows * 256 * 4.0 + rows * (3 * 4 + 3 * 4 + 3 * 4) + (rows // 4) * 256 * (2.0 if mlp_dtype == "f16" else 4.0)      # output + (idx3, w3, xyz) + the pre-multiplied known rows once
                res["roofline"] = {"kernel": {"f16": "fpx16_kernel<4,2,2> (fpx_f16.hip: weights shared through LDS, activations in registers, fp16 pre-multiplied table)", "f32": "fpx32_kernel<1,4,3,2> (fpx_f32.hip)",
                                              "f32x3": "fpx3_kernel<4> (fpx_f32x3.hip: three fp16 MFMAs per product from (hi, lo) operand pairs)"}[mlp_dtype] + " (fp0: finest feature-propagation chain)",
                                   "bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / peak,
                                   "traffic": None, "algorithmic_flops_per_launch": fl, "ms_per_launch": ms, "timing": "one launch bracketed by HIP events on the launch stream (stage pass)"}
                res["roofline_hbm"] = {"bound": "hbm", "achieved": byts / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "algorithmic_bytes_per_launch": byts}
            res["attention_ms_per_step"] = sum(v for k, v in st.items() if k.endswith(".attn"))
            res["stages_ms"] = st
            fls = step_algorithmic_flops(model, a.batch, a.points)
            res["step_flops"] = fls
        except Exception as ex:
            res["roofline"] = {"error": repr(ex)}
        return res

    def emd():
        from patchaugnet_amd import emd_module
        g = torch.Generator().manual_seed(11)
        p1, p2 = (torch.rand(16, 4096, 3, generator=g).cuda() for _ in range(2))
        f = emd_module.emdModule()
        res = {"shape": [16, 4096, 3], "eps": 0.02, "workload": "emdModule forward as patch_emd_loss calls it (pointnetvlad_loss.py:205-221)"}
        for iters in (64, 1024):
            ms = ev_time_ms(lambda: f(p1, p2, 0.02, iters), iters=3, warm=1)
            res[f"ms_{iters}_iters"] = ms
        return res

    def config2_sweep():
        """SURVEY.md section 8(d), config 2 beside the headline point: the street-like input (3-5 planes + 5 % exact duplicates: ties in sampling and
        neighbour search) at batch 32, and uniform clouds at batch 1 / 8 / 256 -- same engine, one hipGraph per stream, inputs resident in HBM."""
        from patchaugnet_amd import configs, patch_aug_net
        from patchaugnet_amd.extract import GraphedExtractor
        from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
        model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
        model.load_state_dict(seeded_state_dict(model.state_dict()))
        model = model.cuda().eval()
        res = []
        for batch, kind, steps in ((32, "street", 40), (1, "uniform", 200), (8, "uniform", 100), (256, "uniform", 12)):
            x = synthetic_submaps(batch, a.points, seed=1234, kind=kind).cuda()
            with torch.no_grad():
                gx = GraphedExtractor(model, tuple(x.shape), a.streams, resident_inputs=[x])
                rates = []
                for rep in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    gx.begin()
                    for _ in range(steps):
                        gx.run(x)
                    gx.end()
                    torch.cuda.synchronize()
                    if rep:
                        rates.append(steps * batch / (time.perf_counter() - t0))
            rates.sort()
            res.append({"batch": batch, "input": kind, "steps": steps, "streams": a.streams, "submaps_per_s": rates[1], "min": rates[0], "max": rates[-1],
                        "ms_per_step": batch / rates[1] * 1e3})
            del gx, x
            gc.collect()
            torch.cuda.empty_cache()
        return {"workload": "PatchAugNet inference, 4096-pt synthetic submaps, 1xMI355X (BASELINE.json configs[1] at other batch sizes / input distribution)", "points": res}

    guarded("configs0_pointnetvlad_cpu", pointnetvlad_cpu)
    guarded("configs1_sweep", config2_sweep)
    guarded("configs2_oxford_eval", lambda: oxford_eval(a))
    guarded("configs3_training_step", train)
    guarded("configs4_pptnet_f32", lambda: extract_rate("pptnet", "f32"))
    guarded("configs4_pptnet_f16", lambda: extract_rate("pptnet", "f16"))
    guarded("configs1_f32x3_opt_in", lambda: extract_rate("patch_aug_net", "f32x3"))
    guarded("emd_16x4096", emd)
    return out


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it: start one child per GPU (rank i on GPU i) with the torchrun environment
    contract and wait for them.  Rank 0's stdout (the JSON line) is this process's stdout."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} requested but this box has {have} visible GPU(s); refusing to print a {have}-GPU number "
                         f"as an {a.gpus}-GPU one")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    if rc:
        raise SystemExit(f"bench.py: a rank exited with status {rc}")


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return self_launch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} does not match WORLD_SIZE={world} of the launcher")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} are visible")
    from patchaugnet_amd.hostcpu import limit_host_threads
    # host-side torch ops sized by os.cpu_count() overrun the cgroup CPU quota and get the process throttled; N ranks share the grant
    from patchaugnet_amd.hostcpu import cpu_budget
    limit_host_threads(max(1, cpu_budget() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))))
    torch.cuda.set_device(local)
    if a.config == "train":
        if world != 1:
            raise SystemExit("bench.py --config train is a single-GPU configuration (BASELINE.json configs[3])")
        return train_bench(a)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    if a.config == "oxford":      # BASELINE.json configs[2]: same code at every N (strong scaling: the set is fixed, the shards shrink)
        line = oxford_eval(a, world, rank)
        if rank == 0:
            print(json.dumps(line))
        if dist is not None:
            dist.destroy_process_group()
        return

    from patchaugnet_amd import configs, patch_aug_net
    from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps

    cfg = configs.patch_aug_net_config()
    if a.points != 4096:
        cfg = configs.scaled_config(cfg, a.points)
    if a.model == "pptnet":
        from patchaugnet_amd import pptnet
        cfg = configs.pptnet_config() if a.points == 4096 else configs.scaled_config(configs.pptnet_config(), a.points)
        model = pptnet.Network(param=cfg, use_normalize=True)
    else:
        model = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    sd = seeded_state_dict(model.state_dict())
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    model.mlp_dtype = a.mlp_dtype
    if a.module_path:
        model.fused_eval = False
    x = synthetic_submaps(a.batch, a.points, seed=1234 + rank).cuda()
    descs = torch.empty(a.steps, a.batch, 256, device="cuda")

    from patchaugnet_amd.extract import GraphedExtractor, StreamPipeline
    use_graphs = not a.no_graphs and not a.module_path
    if use_graphs:
        # one captured hipGraph of the step per stream (same kernels, same arguments, bit-identical descriptors: tests/test_gpu_extract.py);
        # a replay costs the host 0.03 ms instead of 0.30 ms of Python launches
        class _Graphed:
            def __init__(self):
                self.gx = GraphedExtractor(model, tuple(x.shape), a.streams, resident_inputs=[x])      # the batch is resident in HBM: read in place
            def begin(self):
                self.gx.begin()
            def end(self):
                self.gx.end()
            def submit(self, fn, i):
                self.gx.run(x, out=descs[i] if i >= 0 else None)
        try:
            pipe = _Graphed()
        except Exception as ex:      # capture refused (driver / allocator state): same kernels issued from Python instead
            print(f"bench.py: hipGraph capture failed ({ex!r}); falling back to python launches", file=sys.stderr)
            use_graphs = False
            torch.cuda.synchronize()
    if not use_graphs:
        pipe = StreamPipeline(a.streams)
    # Look-ahead pipeline (default): the K steps' batches are K DISTINCT resident batches (one (K, B, 1, N, 3) tensor: the dataset in HBM); the first-level
    # sampling of `--group` consecutive batches is one launch a group ahead of the graphs that consume it (extract.SampledAheadExtractor).  Every
    # step's descriptors come from its own batch through the same kernels; nothing is cached between steps or repetitions.
    ahead = None
    x_steps = None
    if use_graphs and not a.no_lookahead and a.streams >= 2:
        try:
            from patchaugnet_amd.extract import SampledAheadExtractor
            uniq = min(a.steps, 64)                                      # distinct batches (a 64-batch dataset is 100 MB; steps beyond reuse them in order)
            x_steps = torch.stack([synthetic_submaps(a.batch, a.points, seed=1234 + rank * 1000 + i) for i in range(uniq)]).cuda()
            if uniq < a.steps:
                x_steps = torch.cat([x_steps] * ((a.steps + uniq - 1) // uniq))[:a.steps].contiguous()
            with torch.no_grad():
                ahead = SampledAheadExtractor(model, tuple(x.shape), a.streams, group=a.group)
        except Exception as ex:
            print(f"bench.py: look-ahead pipeline unavailable ({ex!r}); plain per-stream graphs", file=sys.stderr)
            ahead = None

    def one(i):
        d = model(x, return_feat=False)
        if i >= 0:
            descs[i].copy_(d)

    def run_steps(n, warm=False):
        """n steps of the hot path; descriptors of step i -> descs[i] (warm-up steps write nowhere that is read)"""
        if ahead is not None:
            ahead.extract(x_steps[:n] if n <= x_steps.shape[0] else x_steps, descs[:n] if n <= descs.shape[0] else descs)
            return
        pipe.begin()
        for i in range(n):
            pipe.submit(one, -1 if warm else i)
        pipe.end()

    with torch.no_grad():
        torch.manual_seed(0)
        run_steps(min(max(a.warmup, a.streams), a.steps), warm=True)   # every stream (and its allocator pool) gets warmed
        if dist is not None:   # communicator and buffers for the one collective of the timed region are set up before the clock starts
            gathered = torch.empty(world * a.steps * a.batch, 256, device="cuda")
            dist.all_gather_into_tensor(gathered, descs.view(-1, 256))
            dist.barrier()
        # The timed region -- EXACTLY K steps between barrier + synchronize on both sides, ending (N > 1) with the one RCCL all-gather -- is
        # repeated a.reps times; the headline is the MEDIAN repetition (max over ranks per repetition), min / max ride along.  One 19 ms
        # region is at the mercy of a single scheduling hiccup; the spread says how much.
        rep_dt, rep_local, rep_ag = [], [], []
        for rep in range(max(a.reps, 1)):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(a.steps)
            ag_ms = None
            if dist is not None:   # the one exchange step: every rank's descriptors to every rank (its own duration is reported beside the total)
                torch.cuda.synchronize()
                t_local = time.perf_counter() - t0          # this rank's own extraction, before the exchange
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_gather_into_tensor(gathered, descs.view(-1, 256))
                e1.record()
            torch.cuda.synchronize()
            if dist is not None:
                ag_ms = e0.elapsed_time(e1)
                dist.barrier()
            dt = time.perf_counter() - t0
            if dist is None:
                t_local = dt
            rep_dt.append(dt)
            rep_local.append(t_local)
            rep_ag.append(ag_ms)
    rccl_ranks = per_rank = None
    if dist is not None:     # slowest rank per repetition, distinct rank ids seen through the all-gather, every rank's own rate
        from patchaugnet_amd.distributed import run_stats
        rep_dt, rccl_ranks, per_rank = run_stats(rep_dt, rep_local, a.steps * a.batch, "cuda")
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    order = sorted(range(len(rep_dt)), key=lambda i: rep_dt[i])
    mid = order[len(order) // 2]
    dt, ag_ms = rep_dt[mid], rep_ag[mid]

    submaps = world * a.steps * a.batch
    line = {
        "metric": f"4096-pt submaps/sec descriptor extraction ({'PatchAugNet' if a.model == 'patch_aug_net' else 'PPT-Net'}, inputs resident in HBM)",
        "value": submaps / dt, "unit": "submaps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "f16": "f16 MFMA operands in the shared-MLP chains, fp32 accumulate / everything else fp32",
                  "f32x3": "f32 everywhere except the two 256 -> 256 layers of the finest FP level: products from (hi, lo) fp16 operand pairs (three fp16 MFMAs, ~2^-21 relative), fp32 accumulate -- OPT-IN, not the headline"}[a.mlp_dtype], "data": "synthetic",
        "config": {"workload": (f"PatchAugNet inference, {a.points}-pt synthetic submaps, batch={a.batch}, 1xMI355X per rank (BASELINE.json configs[1])"
                                if a.model == "patch_aug_net" else
                                f"PPT-Net inference, {a.points}-pt synthetic submaps, batch={a.batch} per MI355X (BASELINE.json configs[4])"),
                   "batch_per_gpu": a.batch, "points": a.points,
                   "path": "fused HIP engine" if model.fused_eval else "HIP point ops + torch dense ops (module path)",
                   "weights": "key-seeded random init", "parallelism": f"dp{world}", "streams": a.streams,
                   "launch": ((f"look-ahead pipeline: the farthest-point sampling (every level) of {a.group} consecutive batches ({a.group * a.batch} clouds) is one launch per level on a sampling stream, a group ahead; "
                               f"the rest of every step is one hipGraph replay on {a.streams} feature streams reading the group's coordinates and samples in place "
                               "(extract.SampledAheadExtractor); K distinct HBM-resident batches, descriptors copied to the result buffer") if ahead is not None else
                              "hipGraph replay of the whole step per stream (the HBM-resident batch is read in place; descriptors copied to the result buffer)") if use_graphs else "python launches"},
    }
    if ahead is not None and world == 1 and use_graphs and not a.only_steps:
        # the plain pipeline beside it, same run: one captured graph of the whole step per stream, the resident batch read in place (the headline of rounds 2-5)
        try:
            with torch.no_grad():
                rates = []
                for rep in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    pipe.begin()
                    for i in range(a.steps):
                        pipe.submit(one, i)
                    pipe.end()
                    torch.cuda.synchronize()
                    if rep:
                        rates.append(a.steps * a.batch / (time.perf_counter() - t0))
            rates.sort()
            line["plain_graph_pipeline"] = {"value": rates[1], "unit": "submaps/s", "min": rates[0], "max": rates[-1], "steps": a.steps,
                                            "note": "extract.GraphedExtractor: the first-level sampling inside every step's graph; after a synchronisation its four graphs run in lock-step for ~16 steps"}
        except Exception as ex:
            line["plain_graph_pipeline"] = {"error": repr(ex)}
    line["repetitions"] = {"count": len(rep_dt), "statistic": "median", "submaps_per_s": [round(submaps / t, 1) for t in rep_dt],
                           "min": submaps / max(rep_dt), "max": submaps / min(rep_dt)}
    if ag_ms is not None:
        line["all_gather_ms"] = ag_ms      # rank 0's view of the one RCCL collective that ends the timed region (median repetition)
        line["rccl_ranks"] = rccl_ranks    # distinct rank ids received through the all-gather: must equal n_gpus
        line["per_rank_submaps_per_s"] = [round(v, 1) for v in per_rank]
    if world == 1 and (a.model != "patch_aug_net" or a.mlp_dtype != "f32"):   # non-headline configurations: stage times only
        try:
            line["kernels"] = {"stages_ms": stage_pass(model, x)}
        except Exception as ex:
            line["kernels"] = {"stages_ms": {"error": repr(ex)}}
    elif world == 1:
        if not a.no_kernel_pass:
            g = grouping_roofline()
            line["kernels"] = {"grouping": g}
            try:
                st = stage_pass(model, x)
                line["kernels"]["stages_ms"] = st
                with torch.no_grad():
                    b2b = profiling_fp0_launch_ms(model, x)
                if b2b:       # the roofline's launch duration: the average over nine back-to-back launches between two HIP events on the launch stream.
                    #           The stage time above brackets ONE launch and carries ~15 us of event / dispatch overhead (0.308 vs 0.293 ms in
                    #           rocprofv3's trace of the same command); the back-to-back average agrees with the trace to ~2 % (profiles/).
                    line["kernels"]["fp0_chain_back_to_back_ms"] = b2b
                    line["kernels"]["fp0_chain_single_bracketed_ms"] = st.get("fp0.chain")
                    st = dict(st, **{"fp0.chain": b2b})
                pmc, note = (None, "--no-pmc") if a.no_pmc else measure_traffic(a.batch, a.points)
                line.update(dominant_kernel_roofline(st, cfg, a.batch, a.points, g, pmc, note))
                if "roofline" in line and b2b:
                    line["roofline"]["timing"] = "HIP events on the launch stream around 9 back-to-back launches of the kernel (average); one bracketed launch: kernels.fp0_chain_single_bracketed_ms"
                try:      # the clock the chip holds under the dominant kernel (the nominal peak is 2.4 GHz)
                    ck = dominant_kernel_clock(model, x) if "roofline" in line and line["roofline"].get("bound") == "mfma" else None
                    if ck:
                        line["roofline"]["shader_clock_mhz"] = ck
                        line["roofline"]["shader_clock_note"] = ("s_memtime ticks per s_memrealtime microsecond over the first 512 wave tiles of one launch inside a forward "
                                                                 "(the kernel's own debug stamps); peak = 256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz assumes the nominal clock")
                except Exception as ex:
                    line["roofline"]["shader_clock_mhz"] = {"error": repr(ex)}
                try:
                    line["roofline_latency"] = fps_latency_roofline(a.batch, a.points, cfg["SAMPLING"][0])
                except Exception as ex:
                    line["roofline_latency"] = {"error": repr(ex)}
                if not a.no_trace and a.batch == 32 and a.points == 4096:
                    # every kernel of the step from rocprofv3's kernel trace of this run, alone (one stream) and inside the pipeline; the dominant
                    # kernel's roofline.frac becomes the TRACE figure (the back-to-back HIP-event average is the flattering one)
                    try:
                        sr = stage_rooflines(a, model, line["ms_per_step"])
                        line["stage_rooflines"] = sr
                        dom = next((r for r in sr.get("rows", []) if r.get("stage") == "fp0.chain" and "frac" in r), None)
                        if dom and "roofline" in line:
                            r = line["roofline"]
                            r["frac_hip_events_back_to_back"], r["ms_per_launch_hip_events_back_to_back"] = r["frac"], r["ms_per_launch"]
                            r["ms_per_launch"], r["achieved"], r["frac"] = dom["us_per_launch"] / 1e3, dom["achieved"], dom["frac"]
                            r["frac_in_pipeline"], r["ms_per_launch_in_pipeline"] = dom["frac_in_pipeline"], (dom["us_per_launch_in_pipeline"] or 0) / 1e3
                            if isinstance(r.get("shader_clock_mhz"), dict) and r["shader_clock_mhz"].get("mhz"):
                                r["frac_of_peak_at_measured_clock"] = r["frac"] * 2400.0 / r["shader_clock_mhz"]["mhz"]
                            r["timing"] = ("rocprofv3 --kernel-trace average of the kernel inside the real step, one stream (frac) and the headline's "
                                           f"{a.streams}-stream pipeline (frac_in_pipeline), measured in this run; HIP events around 9 back-to-back launches: frac_hip_events_back_to_back")
                    except Exception as ex:
                        line["stage_rooflines"] = {"error": repr(ex)}
            except Exception as ex:  # attribution is diagnostics; never lose the bench line over it
                line["kernels"]["stages_ms"] = {"error": repr(ex)}
        try:      # the number that describes the whole step (the per-kernel roofline above is its best kernel): algorithmic FLOPs / step time / peak
            fl = step_algorithmic_flops(model, a.batch, a.points)
            line["step_mfma_frac"] = {"algorithmic_flops_per_step": fl, "achieved_tflops": fl / (line["ms_per_step"] * 1e-3) / 1e12, "peak_tflops": MFMA_F32_PEAK_TFLOPS,
                                      "frac": fl / (line["ms_per_step"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                      "note": "post-fold dense-layer FLOPs of one step (step_algorithmic_flops) / ms_per_step / fp32 MFMA peak; sampling and neighbour search (VALU) not counted"}
        except Exception as ex:
            line["step_mfma_frac"] = {"error": repr(ex)}
        if a.only_steps:
            print(json.dumps(line))
            return
        try:
            line["reference_protocol"] = reference_protocol(model, a.points)
        except Exception as ex:
            line["reference_protocol"] = {"error": repr(ex)}
        try:
            line["pcie_inclusive"] = pcie_inclusive(model, a, pipe.gx if use_graphs else pipe)
        except Exception as ex:
            line["pcie_inclusive"] = {"error": repr(ex)}
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, {k: v.cpu() for k, v in sd.items()}, a.cpu_batch, a.points)
        if not a.no_extras:
            del pipe
            line["other_configs"] = extras(a)
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
