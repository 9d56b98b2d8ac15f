#!/usr/bin/env python
"""Per-phase cycle stamps of the chain kernels (profiling aid): python tools/chain_phases.py [sa0|sa1|sa2|fp0|fp1|fp2 ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from patchaugnet_amd import _lib, configs, patch_aug_net
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps

model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda().eval()
x = synthetic_submaps(32, 4096, seed=1234).cuda()
lib = _lib.lib()
lib.pa_chain_debug_buffer.argtypes = [ctypes.c_void_p]
lib.pa_chain_debug_buffer.restype = None
with torch.no_grad():
    for _ in range(2):
        model(x, return_feat=False)
    eng = model._engine
    for which in (sys.argv[1:] or ["fp0"]):
        kind, lvl = which[:2], int(which[2])
        chain = (eng.sa if kind == "sa" else eng.fp)[lvl]
        names = ["sa"] if kind == "sa" else ["fp_premul", "fp"]
        buf = torch.zeros(512 * 8, dtype=torch.int64, device="cuda")
        used = []

        def wrap(nm):
            orig = getattr(chain, nm)

            def wrapped(*a, _o=orig, **k):
                used.append(nm)
                lib.pa_chain_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
                r = _o(*a, **k)
                lib.pa_chain_debug_buffer(None)
                return r
            setattr(chain, nm, wrapped)
            return orig
        origs = {nm: wrap(nm) for nm in names}
        model(x, return_feat=False)
        torch.cuda.synchronize()
        for nm, o_ in origs.items():
            setattr(chain, nm, o_)
        name = used[0] if used else names[0]
        t = buf.view(512, 8).cpu().numpy()
        t = t[t[:, 0] > 0]
        if len(t) == 0:
            print(which, "no tiles stamped (the level runs a kernel without phase stamps)")
            continue
        nl = chain.n - (1 if (name == "fp_premul" and lvl == 0) else 0)
        d = t[:, 1:nl + 2] - t[:, 0:nl + 1]
        parts = ["prologue"] + [f"layer{i}" for i in range(nl)]
        print(which, f"({len(t)} tiles stamped)", "  ".join(f"{p} {np.median(d[:, i]):.0f}" for i, p in enumerate(parts)),
              " total", np.median(t[:, nl + 1] - t[:, 0]))
        st, en = t[:, 0] - t[:, 0].min(), t[:, nl + 1] - t[:, 0].min()
        q = lambda a: " ".join(f"{np.percentile(a, p):.0f}" for p in (0, 10, 25, 50, 75, 90, 100))
        print("   start offsets (cycles, percentiles 0/10/25/50/75/90/100):", q(st), "| end offsets:", q(en))
