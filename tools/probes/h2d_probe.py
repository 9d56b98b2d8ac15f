import torch, time
torch.cuda.init(); torch.zeros(1).cuda()
def T():
    torch.cuda.synchronize(); return time.perf_counter()
big = torch.rand(18,1,4096,3)
small = torch.randperm(40)
bigp = big.pin_memory()
w = torch.rand(4096,4096,device="cuda")
for name, fn in [("small pageable", lambda: small.cuda()), ("big pageable", lambda: big.cuda()), ("big pinned", lambda: bigp.to("cuda", non_blocking=True)),
                 ("small pageable after gpu work", lambda: (w@w, small.cuda())), ("big pageable after gpu work", lambda: (w@w, big.cuda())),
                 ("d2h item", lambda: float(w[0,0])), ("d2h item after work", lambda: float((w@w)[0,0]))]:
    ts = []
    for i in range(60):
        t0 = T(); fn(); ts.append((T()-t0)*1e3)
    ts2 = sorted(ts)
    print(f"{name:32s} median {ts2[30]:.3f} ms  max {ts2[-1]:.3f}  >5ms: {sum(t>5 for t in ts)}")
