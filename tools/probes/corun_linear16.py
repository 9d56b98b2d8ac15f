"""fps_reg_kernel on one stream while pa_linear_f16 / pa_linear (one-layer chain kernels, fp16 / fp32 operands) loop on another: does the sampling result change?
python tools/probes/corun_linear16.py [trials]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import pointops, engine
from patchaugnet_amd._lib import call, ptr
from patchaugnet_amd.weights import synthetic_submaps
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
xs = [synthetic_submaps(32, 4096, 70 + i, "street" if i % 3 == 0 else "uniform").cuda().squeeze(1).contiguous() for i in range(4)]
ref = [pointops.furthestsampling_gather(x, 1024)[0].clone() for x in xs]
g = torch.Generator().manual_seed(0)
shapes = ((131072, 256, 256), (32768, 256, 256), (131072, 64, 64), (4096, 512, 256))
for rows, k, n in shapes[:int(os.environ.get("CORUN_SHAPES", "4"))]:
    x = torch.randn(rows, k, generator=g).cuda()
    wt = (torch.randn(k, n, generator=g) / k ** 0.5).cuda().contiguous()
    bias = torch.zeros(n, device="cuda")
    out = torch.empty(rows, n, device="cuda")
    wp = engine.pack_weights(wt); wp16 = engine.pack_weights_f16(wt)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for name in ("pa_linear_f16", "pa_linear"):
        bad = 0
        for t in range(trials):
            with torch.cuda.stream(sa):
                for _ in range(8):
                    call(name, rows, k, n, ptr(x), k, ptr(wt), ptr(wp16 if name.endswith("f16") else wp), ptr(bias), 1, None, 0, ptr(out), n)
            with torch.cuda.stream(sb):
                gg = pointops.furthestsampling_gather(xs[t % 4], 1024)[0]
            torch.cuda.synchronize()
            bad += int(not torch.equal(gg, ref[t % 4]))
        print(f"{name:14s} rows {rows} k {k} n {n}: sampling differs in {bad} of {trials} runs")
