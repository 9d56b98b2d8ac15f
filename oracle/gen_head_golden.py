"""Generate tests/golden/heads.npz from the REFERENCE's SpatialPyramidNetVLAD (build container only): aggregation types 0 (FC),
2 (APFA) and 3 (max-pool), each with and without context gating (place_recognition/patch_aug_net/models/loupe.py:225-329).  Pins
oracle/models_cpu.spvlad_aggregate, which in turn checks the HIP head kernels (tests/test_gpu_head.py).

Usage: python -m oracle.gen_head_golden
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CASES = [(0, False), (0, True), (2, False), (2, True), (3, False), (3, True)]
NS, KS = [16, 64, 256], [4, 16, 64]


def features(seed=5, b=3):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(b, 256, n, 1, generator=g) * 0.7 for n in NS]


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, ROOT)
    sys.path.insert(1, os.path.join(REF, "place_recognition", "patch_aug_net", "models"))
    import loupe as ref                                     # the reference's own file, unmodified
    from patchaugnet_amd.weights import seeded_state_dict
    out = {}
    feats = features()
    for t, gating in CASES:
        agg = ref.SpatialPyramidNetVLAD(feature_size=[256] * 3, max_samples=NS, cluster_size=KS, output_dim=[256] * 3, gating=gating,
                                        aggregation_type=t, add_batch_norm=True)
        sd = seeded_state_dict(agg.state_dict(), seed=100 + t)
        agg.load_state_dict(sd, strict=True)
        agg.eval()
        with torch.no_grad():
            d = agg(feats)
        out[f"type{t}_gating{int(gating)}"] = d.numpy()
        out[f"type{t}_gating{int(gating)}_keys"] = np.array(sorted(sd.keys()))
        print(t, gating, tuple(d.shape), float(d.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "heads.npz"), **out)
    print("wrote heads.npz", os.path.getsize(os.path.join(GOLD, "heads.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
