"""Does confining the first level's sampling to a few CUs pay?  The sampling launch of a batch (32 workgroups, ~0.77 ms of serial rounds) takes a
CU each and the eight-wave chain workgroups of the other streams cannot share those CUs (registers).  Here the launch goes to streams created with
a CU mask (hipExtStreamCreateWithCUMask) and the rest of the step is a graph on the slot's own stream behind an event.
usage: python tools/probes/cumask.py [steps]"""
import ctypes
import sys
import time
import torch
sys.path.insert(0, ".")
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import GraphedExtractor, _prime_stream_queues
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps

hip = ctypes.CDLL("libamdhip64.so")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B, N = 32, 4096


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    r = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert r == 0, f"hipExtStreamCreateWithCUMask -> {r}"
    return torch.cuda.ExternalStream(s.value)


def every(k, total=256, phase=0):
    w = [0] * (total // 32)
    for i in range(phase, total, k):
        w[i // 32] |= 1 << (i % 32)
    return w


def first(n, total=256):
    w = [0] * (total // 32)
    for i in range(n):
        w[i // 32] |= 1 << (i % 32)
    return w


model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True) if hasattr(configs, "patch_aug_net_config") else None
if model is None:
    raise SystemExit("configs.patch_aug_net_config missing")
model.load_state_dict(seeded_state_dict(model.state_dict()), strict=True)
model = model.cuda().eval()
x = synthetic_submaps(B, N, seed=1234).cuda()
descs = torch.empty(STEPS, B, 256, device="cuda")
dev = torch.device("cuda", 0)


def rate(run, begin, end, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        begin()
        for i in range(STEPS):
            run(i)
        end()
        torch.cuda.synchronize()
        out.append(STEPS * B / (time.perf_counter() - t0))
    out.sort()
    return out[len(out) // 2], out[0], out[-1]


with torch.no_grad():
    gx = GraphedExtractor(model, tuple(x.shape), 4, resident_inputs=[x])
    r = rate(lambda i: gx.run(x, out=descs[i]), gx.begin, gx.end)
    ref = descs[0].clone()
    print("baseline, 4 graph streams:                     %.0f  (%.0f .. %.0f)" % r, flush=True)

    eng = model._engine
    m0 = eng.sampling[0]
    xyz = x.squeeze(1).contiguous()

    class Split:
        def __init__(self, fstreams, nslots):
            self.f = fstreams
            self.slots = []
            cur = torch.cuda.current_stream()
            for i in range(nslots):
                st = torch.cuda.Stream()
                cidx = torch.empty(B, m0, dtype=torch.int32, device="cuda")
                nxyz = torch.empty(B, m0, 3, device="cuda")
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    for _ in range(2):
                        eng.sample_first_level(xyz, cidx, nxyz)
                        eng.forward(x, views=False, s0=(cidx, nxyz))
                cur.wait_stream(st)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    y = eng.forward(x, views=False, s0=(cidx, nxyz))[0]
                self.slots.append((g, y, st, cidx, nxyz, torch.cuda.Event(), torch.cuda.Event()))
            torch.cuda.synchronize()
            self.i = 0

        def begin(self):
            cur = torch.cuda.current_stream()
            for s in self.f:
                s.wait_stream(cur)
            for sl in self.slots:
                sl[2].wait_stream(cur)
                sl[6].record(sl[2])

        def run(self, i):
            g, y, st, cidx, nxyz, ev, free = self.slots[self.i % len(self.slots)]
            fs = self.f[self.i % len(self.f)]
            self.i += 1
            fs.wait_event(free)
            with torch.cuda.stream(fs):
                eng.sample_first_level(xyz, cidx, nxyz)
                ev.record(fs)
            st.wait_event(ev)
            with torch.cuda.stream(st):
                g.replay()
                descs[i].copy_(y, non_blocking=True)
                free.record(st)

        def end(self):
            cur = torch.cuda.current_stream()
            for sl in self.slots:
                cur.wait_stream(sl[2])

    def trial(name, fstreams, nslots):
        sp = Split(fstreams, nslots)
        r = rate(sp.run, sp.begin, sp.end)
        same = bool((descs[0] == ref).all())
        print("%-46s %.0f  (%.0f .. %.0f)  identical=%s" % (name, *r, same), flush=True)

    MASKS = {"e4": every(4), "e8": every(8), "e2": every(2), "e16": every(16), "f64": first(64), "f32": first(32), "e4p2": every(4, phase=2), "e3": every(3)}
    kind, nf, ns = (sys.argv[2], int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("plain", 2, 6)
    if kind == "check":          # is the mask applied at all?  a chip-filling launch on a masked stream must slow down by the mask's ratio
        a = torch.randn(8192, 8192, device="cuda")
        for label in ("none", "e2", "e4", "e16", "f32"):
            fs = torch.cuda.Stream() if label == "none" else masked_stream(MASKS[label])
            with torch.cuda.stream(fs):
                fs.wait_stream(torch.cuda.current_stream())
                torch.sin(a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(fs)
                for _ in range(5):
                    b = a @ a
                e1.record(fs)
            torch.cuda.synchronize()
            print("8192^3 library product on stream [%s]: %.2f ms" % (label, e0.elapsed_time(e1) / 5), flush=True)
    else:
        fstreams = [torch.cuda.Stream() if kind == "plain" else masked_stream(MASKS[kind]) for _ in range(nf)]
        trial(f"split, {nf} sampling streams [{kind}], {ns} slots:", fstreams, ns)
    sys.exit(0)

    # the sampling launch alone on a masked stream: does packing slow it?
    for label, words in (("unmasked", None), ("every 4th", every(4)), ("every 8th", every(8)), ("every 16th", every(16)), ("first 32", first(32))):
        fs = torch.cuda.Stream() if words is None else masked_stream(words)
        cidx = torch.empty(B, m0, dtype=torch.int32, device="cuda")
        nxyz = torch.empty(B, m0, 3, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(fs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            eng.sample_first_level(xyz, cidx, nxyz)
            e0.record(fs)
            for _ in range(5):
                eng.sample_first_level(xyz, cidx, nxyz)
            e1.record(fs)
        torch.cuda.synchronize()
        print("sampling launch alone, %-12s %.1f us" % (label, e0.elapsed_time(e1) / 5 * 1e3), flush=True)
