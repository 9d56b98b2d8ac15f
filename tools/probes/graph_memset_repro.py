"""Do memset / memcpy NODES of a captured hipGraph stay ordered in front of the kernels that follow them?  Minimal form of what train.GraphedTrainer
hit in round 5: capture [hipMemsetAsync(x, 0) -> x += 1] (and [hipMemcpyAsync(y <- src) -> y += 1]) a few times per graph with
two buffers each, replay with eager work between replays, and check x == 1 (y == src + 1) after every replay.  python tools/probes/graph_memset_repro.py"""
import ctypes, sys
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
replays = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for n in (61440 * 3, 1 << 20, 1000):
    xs = [torch.full((n,), 7.0, device="cuda") for _ in range(2)]
    src = torch.arange(n, device="cuda", dtype=torch.float32)
    ys = [torch.full((n,), -1.0, device="cuda") for _ in range(2)]
    big = torch.randn(4096, 4096, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for x in xs: x.add_(1.0)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        t = big * 2.0                                   # something in front, so the nodes are not the first of the graph
        for x in xs:
            hip.hipMemsetAsync(x.data_ptr(), 0, n * 4, st)
        for x in xs:
            x.add_(1.0)                                 # accumulates: 1 only if the memset ran first
        for y in ys:
            hip.hipMemcpyAsync(y.data_ptr(), src.data_ptr(), n * 4, 3, st)
        for y in ys:
            y.add_(1.0)
    scratch = torch.empty(1 << 20, device="cuda")
    bad_set = bad_cpy = 0
    seen = []
    for r in range(replays):
        g.replay()
        scratch.fill_(float(r)); junk = torch.full((1 << 20,), float("nan"), device="cuda"); float(scratch[0]); del junk
        torch.cuda.synchronize()
        bad_set += any(not bool((x == 1.0).all()) for x in xs)
        if r < 6: seen.append([sorted(set(x[:: max(n // 7, 1)].tolist()))[:3] for x in xs])
        bad_cpy += any(not bool((y == src + 1.0).all()) for y in ys)
    print(f"n = {n}: {replays} replays; memset-then-accumulate wrong in {bad_set}, memcpy-then-accumulate wrong in {bad_cpy}; values of the two memset buffers after replays 1..6: {seen}")
