#!/usr/bin/env python
"""SURVEY.md section 8(d) measurements beside the headline bench line: the batch-size sweep and the "street-like" input
distribution of config 2, and config 4 (one training step: 18-cloud tuple fwd+bwd with the Chamfer patch-reconstruction loss;
Chamfer on (3072,20,3); EMD on (16,4096,3), eps 0.02, iters 64 / 1024).  Prints one JSON object.
    python tools/config_sweep.py > gpurun_out/config_sweep.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from patchaugnet_amd import configs, patch_aug_net, chamfer_dist, emd_module
from patchaugnet_amd.extract import StreamPipeline
from patchaugnet_amd.train import training_step
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def extraction_rate(model, batch, kind, streams=4, steps=None):
    x = synthetic_submaps(batch, 4096, seed=1234, kind=kind).cuda()
    steps = steps or max(12, min(100, 3200 // batch))
    out = torch.empty(steps, batch, 256, device="cuda")
    pipe = StreamPipeline(streams)

    def one(i):
        d = model(x, return_feat=False)
        if i >= 0:
            out[i].copy_(d)

    with torch.no_grad():
        pipe.begin()
        for _ in range(max(6, streams)):
            pipe.submit(one, -1)
        pipe.end()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.begin()
        for i in range(steps):
            pipe.submit(one, i)
        pipe.end()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"batch": batch, "input": kind, "streams": streams, "steps": steps, "submaps_per_s": round(steps * batch / dt, 1),
            "ms_per_step": round(dt / steps * 1e3, 4)}


def main():
    from patchaugnet_amd.hostcpu import limit_host_threads
    limit_host_threads()
    torch.cuda.set_device(0)
    res = {"device": torch.cuda.get_device_name(0)}
    model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
    model.load_state_dict(seeded_state_dict(model.state_dict()))
    model = model.cuda().eval()

    res["config2_batch_sweep"] = [extraction_rate(model, b, "uniform") for b in (1, 8, 32, 100, 256)]
    res["config2_street_like"] = [extraction_rate(model, b, "street") for b in (32, 100)]
    res["config2_single_stream_latency"] = [extraction_rate(model, b, "uniform", streams=1) for b in (1, 32)]
    model.geo_overlap = True      # latency mode: coordinate-only kernels of the coarser levels on a side stream (engine.py)
    res["config2_single_stream_latency_mode"] = [extraction_rate(model, b, "uniform", streams=1) for b in (1, 32)]
    model.geo_overlap = False

    # config 4: the reference's native tuple = 1 query + 2 positives + 14 negatives + 1 other negative = 18 clouds
    # (configs/patch_aug_net.yaml:60-62); nn_dict with 2 (query, positive) pairs => 3 related clouds => Chamfer on (3072,20,3)
    g = torch.Generator().manual_seed(5)
    q = torch.rand(1, 1, 4096, 3, generator=g) * 2 - 1
    pos = torch.rand(1, 2, 4096, 3, generator=g) * 2 - 1
    neg = torch.rand(1, 14, 4096, 3, generator=g) * 2 - 1
    oth = torch.rand(1, 1, 4096, 3, generator=g) * 2 - 1
    nn_dict = {(0, 1): torch.randint(0, 4096, (1024, 1), generator=g).numpy(), (0, 2): torch.randint(0, 4096, (1024, 1), generator=g).numpy()}
    opt = torch.optim.Adam(model.parameters(), lr=1e-5)
    try:
        step = lambda: training_step(model, opt, q, pos, neg, oth, nn_dict=nn_dict)
        losses = step()
        t = timed(step, 5, warm=2)
        res["config4_training_step"] = {"clouds": 18, "related_clouds": 3, "ms_fwd_bwd_opt": round(t * 1e3, 2), "losses": losses,
                                        "note": "module path: autograd over the HIP point-op backward kernels + the HIP training GEMMs (csrc/train_gemm.hip); quadruplet + patch Chamfer loss, Adam step"}
    except Exception as ex:   # keep the other numbers
        res["config4_training_step"] = {"error": repr(ex)}
    model.eval()

    a = (torch.rand(3072, 20, 3, device="cuda") * 2 - 1).requires_grad_(True)
    b = torch.rand(3072, 20, 3, device="cuda") * 2 - 1
    cd = chamfer_dist.ChamferDistanceL1()
    def cham():
        a.grad = None
        cd(a, b).backward()
    res["config4_chamfer_3072x20"] = {"ms_fwd_bwd": round(timed(cham, 50) * 1e3, 4)}
    with torch.no_grad():
        res["config4_chamfer_3072x20"]["ms_fwd"] = round(timed(lambda: chamfer_dist.forward(a.detach(), b), 50) * 1e3, 4)

    e1 = torch.rand(16, 4096, 3, device="cuda")
    e2 = torch.rand(16, 4096, 3, device="cuda")
    emd = emd_module.emdModule()
    res["config4_emd_16x4096"] = {}
    for iters in (64, 1024):
        with torch.no_grad():
            t = timed(lambda: emd(e1, e2, 0.02, iters), 3, warm=1)
            d, _ = emd(e1, e2, 0.02, iters)
        res["config4_emd_16x4096"][f"iters{iters}"] = {"ms_fwd": round(t * 1e3, 2), "mean_sqrt_dist": round(float(d.sqrt().mean()), 6)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
