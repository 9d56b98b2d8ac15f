"""Which Python lines of one fused-engine forward (eval, no autograd) launch library work (aten ops with device tensors)?  PatchAugNet and PPT-Net, fp32 and
fp16 paths.  python tools/probes/engine_libops.py"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from patchaugnet_amd import configs, patch_aug_net, pptnet
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SKIP = ("aten::view", "aten::_unsafe_view", "aten::empty", "aten::as_strided", "aten::detach", "aten::t", "aten::transpose", "aten::permute", "aten::slice",
        "aten::select", "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::alias", "aten::reshape", "aten::_reshape_alias", "aten::empty_like",
        "aten::empty_strided", "aten::new_empty", "aten::unbind", "aten::split", "aten::lift_fresh", "aten::is_", "aten::sym_", "aten::result_type",
        "aten::_local_scalar_dense", "aten::item", "aten::new_empty_strided", "aten::stride", "aten::size", "aten::storage_offset", "aten::numel")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.log = {}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        out = func(*args, **(kwargs or {}))
        if not name.startswith(SKIP):
            ts = [a for a in args if isinstance(a, torch.Tensor)]
            if any(t.is_cuda for t in ts) or (isinstance(out, torch.Tensor) and out.is_cuda):
                fr = [f for f in traceback.extract_stack() if root in f.filename and "engine_libops" not in f.filename][-3:]
                where = " <- ".join(f"{os.path.relpath(f.filename, root)}:{f.lineno}" for f in reversed(fr))
                shp = ",".join("x".join(map(str, t.shape)) for t in ts[:3])
                key = (name, where, shp)
                self.log[key] = self.log.get(key, 0) + 1
        return out


x = synthetic_submaps(32, 4096, seed=3).cuda()
for tag, build, dt in (("patch_aug_net f32", lambda: patch_aug_net.Network(param=configs.patch_aug_net_config()), "f32"),
                       ("patch_aug_net f16", lambda: patch_aug_net.Network(param=configs.patch_aug_net_config()), "f16"),
                       ("pptnet f16", lambda: pptnet.Network(param=configs.pptnet_config()) if hasattr(configs, "pptnet_config") else None, "f16")):
    m = build()
    if m is None:
        continue
    m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval()
    m.mlp_dtype = dt
    with torch.no_grad():
        for _ in range(2):
            m(x, return_feat=False)
        torch.cuda.synchronize()
        with Log() as lg:
            m(x, return_feat=False)
    print(f"---- {tag}: {sum(lg.log.values())} device aten ops in one descriptor-only forward")
    for (name, where, shp), c in sorted(lg.log.items(), key=lambda kv: (-kv[1], kv[0])):
        print(f"{c:4d}x {name:24s} [{shp}]  {where}")
