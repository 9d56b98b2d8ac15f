"""LDS sentinel workgroups (tools/probes/lds_sentinel.hip) resident on one stream while a fused-engine forward runs on another: does any kernel of the
forward write into their LDS?  python tools/probes/lds_sentinel.py [f16|f32] [bytes]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
nbytes = int(sys.argv[2]) if len(sys.argv) > 2 else 49152
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lds_sentinel.so"))
lib.lds_sentinel_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval(); m.mlp_dtype = dt
x = synthetic_submaps(32, 4096, 70).cuda()
with torch.no_grad():
    m(x, return_feat=False)
torch.cuda.synchronize()
blocks = 256
bad = torch.zeros(blocks, dtype=torch.int32, device="cuda"); fw = torch.zeros_like(bad); fv = torch.zeros_like(bad)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
with torch.no_grad():
    for rep in range(6):
        lib.lds_sentinel_launch(blocks, nbytes, 3000, bad.data_ptr(), fw.data_ptr(), fv.data_ptr(), sb.cuda_stream)      # 3 ms resident on stream B
        with torch.cuda.stream(sa):
            for _ in range(2): m(x, return_feat=False)
        torch.cuda.synchronize()
nb = int((bad > 0).sum())
print(f"{dt}: sentinel workgroups of {nbytes} B LDS with changed words: {nb} of {blocks}; total changed words {int(bad.sum())}")
if nb:
    i = int((bad > 0).nonzero()[0])
    print("  e.g. workgroup", i, "changed", int(bad[i]), "first word index", int(fw[i]), f"value 0x{int(fv[i]) & 0xffffffff:08x}")
