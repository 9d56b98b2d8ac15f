"""Soak of the cross-stream guarantee: GraphedExtractor on 4 streams, batch 32, a different batch per replay, REPS rounds over 12 batches, both models in both
mlp dtypes -- every replay must equal the serial forward of its batch bit for bit (tests/test_gpu_extract.py runs one short round of this).
python tools/probes/stream_determinism_soak.py [reps [batch points]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net, pptnet
from patchaugnet_amd.extract import GraphedExtractor
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
B, N = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 4096)          # other shapes pick other tilings
xs = [synthetic_submaps(B, N, 170 + i, "street" if i % 3 == 0 else "uniform").cuda() for i in range(12)]
for name in ("patch_aug_net", "pptnet"):
    for dtype in ("f32", "f16"):
        cfg = configs.patch_aug_net_config() if name == "patch_aug_net" else configs.pptnet_config()
        if N < 4096: cfg = configs.scaled_config(cfg, N)
        elif N > 4096: cfg["NUM_POINTS"], cfg["MAX_SAMPLES"] = N, list(reversed(cfg["SAMPLING"][:-1])) + [N]
        m = (patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True) if name == "patch_aug_net" else pptnet.Network(param=cfg, use_normalize=True))
        m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval(); m.mlp_dtype = dtype
        with torch.no_grad():
            ref = [m(x, return_feat=False).clone() for x in xs]
        gx = GraphedExtractor(m, (B, 1, N, 3), n_streams=4)
        out = torch.empty(len(xs), B, 256, device="cuda")
        bad, worst, t0 = 0, 0.0, time.time()
        for r in range(reps):
            gx.begin()
            for i, x in enumerate(xs):
                gx.run(x, out=out[i])
            gx.end()
            torch.cuda.synchronize()
            for i in range(len(xs)):
                if not torch.equal(out[i], ref[i]):
                    bad += 1; worst = max(worst, float((out[i] - ref[i]).abs().max()))
            out.zero_()
        print(f"{name:14s} {dtype} B={B} n={N}: {reps * len(xs)} replays on 4 streams, {bad} differ from the serial forward (max |diff| {worst:.2e}), {time.time() - t0:.1f} s", flush=True)
        del gx
