"""Descriptors of the fused engine in mlp_dtype="f16" on concurrent streams (captured graphs, replayed concurrently and one at a time) against the serial forward:
identical since the library issues no packed fp32 instruction with operand modifiers (before: up to 5e-3 apart).  python tools/probes/f16_determinism.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, pptnet, patch_aug_net
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
from patchaugnet_amd.extract import GraphedExtractor, StreamPipeline
name = "patch_aug_net"
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval(); m.mlp_dtype = os.environ.get("DT", "f16")
xs = [synthetic_submaps(32, 4096, 70 + i, "street" if i % 3 == 0 else "uniform").cuda() for i in range(8)]
with torch.no_grad():
    a = [m(x, return_feat=False).clone() for x in xs]
def report(tag, out):
    print(tag, [bool(torch.equal(out[i], a[i])) for i in range(len(a))], "max diff", f"{max(float((out[i] - a[i]).abs().max()) for i in range(len(a))):.2e}")
for ns in (2,):
    gx = GraphedExtractor(m, (32, 1, 4096, 3), n_streams=ns)
    out = torch.empty(8, 32, 256, device="cuda")
    gx.begin()
    for i, x in enumerate(xs): gx.run(x, out=out[i])
    gx.end(); torch.cuda.synchronize()
    report(f"graphs, {ns} stream(s), concurrent:", out)
    out = torch.empty(8, 32, 256, device="cuda")
    for i, x in enumerate(xs):
        gx.begin(); gx.run(x, out=out[i]); gx.end(); torch.cuda.synchronize()
    report(f"graphs, {ns} stream(s), one replay at a time:", out)
