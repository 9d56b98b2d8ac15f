"""fps_reg_kernel (pa_furthestsampling_gather, 32 clouds x 4096 points -> 1024 samples) on one stream while a synthetic co-runner occupies the chip on another:
which kind of neighbour changes its result?  python tools/probes/corun_stress.py [trials]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import pointops
from patchaugnet_amd.weights import synthetic_submaps
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "corun_stress.so"))
lib.corun_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
xs = [synthetic_submaps(32, 4096, 70 + i, "street" if i % 3 == 0 else "uniform").cuda().squeeze(1).contiguous() for i in range(4)]
ref = [pointops.furthestsampling_gather(x, 1024)[0].clone() for x in xs]
sink = torch.zeros(4, device="cuda"); buf = torch.zeros(64 << 20, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for mode, name in ((0, "fp16 MFMA"), (1, "fp32 MFMA"), (2, "LDS hammer"), (3, "packed fp32 VALU"), (4, "global memory stream"), (-1, "nothing")):
    for blocks in (512, 2048):
        bad = 0
        for t in range(trials):
            if mode >= 0:
                lib.corun_launch(mode, blocks, 1500, sink.data_ptr(), buf.data_ptr(), buf.numel(), sa.cuda_stream)
            with torch.cuda.stream(sb):
                g = pointops.furthestsampling_gather(xs[t % 4], 1024)[0]
            torch.cuda.synchronize()
            bad += int(not torch.equal(g, ref[t % 4]))
        print(f"co-runner {name:22s} {blocks:5d} workgroups: sampling differs from the serial result in {bad} of {trials} runs")
