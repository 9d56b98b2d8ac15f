"""Fused inference engine for PatchAugNet on MI355X (evaluation mode).

Same parameters as the nn.Module tree (patch_aug_net.Network), different execution plan:

  * activations are point-major (B*N, C) so neighbour gathers read whole contiguous rows;
  * BatchNorm (eval) is folded into the 1x1-conv weights once, weights are stored K-major for the MFMA B operand;
  * each set-abstraction level is ONE kernel (gather + centre-subtract + concat + 3-layer shared MLP + max over the
    neighbourhood, csrc/mlp_chain.hip) and each feature-propagation level is ONE kernel (3-NN interpolation + concat
    + shared MLP); the (B, C, m, k) grouped tensors of the reference (pointops.py:559-570, patch_aug_net.py:234-237)
    are never materialised;
  * the kNN dilation of the reference keeps the nsample NEAREST of dilation*nsample candidates in a random order
    (pointops.py:553-555); the max over the neighbourhood is order-invariant, so the engine
    asks for nsample neighbours directly and skips the permutation.

What the reference computes per op is cited in csrc/*.hip; parity is checked in tests/test_gpu_models.py against the
reference-generated golden vectors and the CPU oracle.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import call, ptr

_F32P = ctypes.POINTER(ctypes.c_float)


def fold_shared_mlp(mlp, device):
    """[(Wt (kpad, n) K-major, bias (n,), k, kpad, n)] for each Conv1x1+BN+ReLU layer, BatchNorm folded (fp64 math)."""
    out = []
    for layer in mlp.children():
        w = layer.conv.weight.detach().double().flatten(1)            # (n, k)
        bn = layer.bn.bn
        scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
        n, k = w.shape
        if n % 16:
            raise ValueError(f"fused engine needs output widths that are multiples of 16, got {n}")
        kpad = (k + 3) // 4 * 4
        wt = torch.zeros(kpad, n, dtype=torch.float64)
        wt[:k] = (w * scale[:, None]).t()
        out.append((wt.float().contiguous().to(device), shift.float().contiguous().to(device), k, kpad, n))
    return out


def pack_weights(wt):
    """K-major (kpad, n) device tensor -> fragment-major packed copy (pa_pack_weights), or None when n % 64 != 0."""
    kpad, n = wt.shape
    if n % 64:
        return None
    wp = torch.empty(kpad * n, dtype=torch.float32, device=wt.device)
    call("pa_pack_weights", kpad, n, ptr(wt), ptr(wp))
    return wp


def pack_weights_f16(wt):
    """K-major (kpad, n) fp32 device tensor -> fp16 fragment packing for the fp16 chain kernels (pa_pack_weights_f16)."""
    kpad, n = wt.shape
    wp = torch.empty(_lib.lib().pa_pack_weights_f16_halfs(kpad, n), dtype=torch.float16, device=wt.device)
    call("pa_pack_weights_f16", kpad, n, ptr(wt), ptr(wp))
    return wp


class _Chain:
    """Host-side descriptor of one pa_mlp_chain call (pointer arrays are built once).  f16=True selects the fp16-operand
    kernels (fp32 accumulation and outputs; BASELINE configs[4])."""

    def __init__(self, layers, f16=False):
        self.f16 = f16
        self.layers = layers
        n = len(layers)
        self.n = n
        self.wt = (ctypes.c_void_p * n)(*[l[0].data_ptr() for l in layers])
        self.packed = [pack_weights(l[0]) for l in layers]
        self.wpk = (ctypes.c_void_p * n)(*[(p.data_ptr() if p is not None else None) for p in self.packed])
        if f16:
            if any(l[4] % 32 for l in layers[:-1]):
                raise ValueError("fp16 chain kernels need hidden widths that are multiples of 32")
            self.packed16 = [pack_weights_f16(l[0]) for l in layers]
            self.wpk = (ctypes.c_void_p * n)(*[p.data_ptr() for p in self.packed16])       # same argument slot, fp16 entry points
        self._fn = "pa_mlp_chain_f16" if f16 else "pa_mlp_chain_packed"
        self.bias = (ctypes.c_void_p * n)(*[l[1].data_ptr() for l in layers])
        self.kpad = (ctypes.c_int * n)(*[l[3] for l in layers])
        self.nout = (ctypes.c_int * n)(*[l[4] for l in layers])
        self.k0 = layers[0][2]
        self.n_last = layers[-1][4]
        self.hidden_ok_pooled = all(l[4] <= 64 for l in layers[:-1])
        # the shared-tile pooled kernel (four waves split the columns of one 4-group tile) also takes hidden widths 64 / 128 / 256
        self._split_widths_ok = all(l[4] in (64, 128, 256) for l in layers[:-1]) and layers[-1][4] % 64 == 0

    def pooled_ok(self, groups, ns):
        """Can the pooled (max over the ns neighbours inside the kernel) variant run this level?  Narrow hidden layers: always; wide ones
        only through the shared-tile variant, which mlp_chain.hip selects for 17 <= ns <= 20 and fewer than 2048 four-group tiles."""
        tiles = (groups + 3) // 4
        # fp32: below ~256 tiles the 16-row unpooled tiling (5x the workgroups) wins; both give the same bits (tests/test_gpu_chain.py)
        return self.hidden_ok_pooled or (self._split_widths_ok and 17 <= ns <= 20 and (0 if self.f16 else 256) <= tiles < 2048)

    def atomic_pool_ok(self, groups, ns):
        """The unpooled shared-tile tiling with the atomic-max epilogue applies (mlp_chain.hip: every layer splits four ways, few enough row
        tiles for the shared-tile variant, fp32)."""
        return (not self.f16 and ns >= 16 and all(l[4] % 64 == 0 for l in self.layers) and all(l[4] in (64, 128, 256, 512) for l in self.layers[:-1])
                and (groups * ns + 31) // 32 < 2048)

    def _common(self):
        return (self.n, ctypes.cast(self.wt, ctypes.c_void_p), ctypes.cast(self.wpk, ctypes.c_void_p), ctypes.cast(self.bias, ctypes.c_void_p),
                ctypes.cast(self.kpad, ctypes.c_void_p), ctypes.cast(self.nout, ctypes.c_void_p))

    def sa(self, xyz, feat, center_idx, nbr_idx, c_feat, pooled, window=None, out=None):
        """pooled: False / 0 = every (group, neighbour) row; True / 1 = max over the neighbourhood in the pooled tiling; 2 = the same result
        from the UNPOOLED shared-tile tiling with an atomic-max epilogue (few groups: the 16-row tiling fills the chip, the pooled one
        does not; fp32 chains, nsample >= 16).  window = (length, offset): only those centres of every cloud, written into `out` (the full
        (B * m, C) buffer) -- pooled first-level kernel only (pa_sa_group_window)."""
        B, n_src, _ = xyz.shape
        m, ns = nbr_idx.shape[1], nbr_idx.shape[2]
        groups = B * m
        pooled = int(pooled)
        if out is None:
            out = torch.empty((groups if pooled else groups * ns, self.n_last), dtype=torch.float32, device=xyz.device)
        if window is not None:
            assert pooled == 1 and not self.f16
            groups = B * window[0]
            rc = _lib.lib().pa_sa_group_window(int(window[0]), int(window[1]))
            assert rc == 0
        call(self._fn, 1, pooled, *self._common(), groups, self.k0, None, 0,
             ptr(xyz), ptr(feat), ptr(center_idx), ptr(nbr_idx), n_src, m, ns, c_feat,
             None, None, None, None, 0, 0, 0, 0, ptr(out), self.n_last)
        return out

    def fp(self, known_feat, idx3, w3, skip, B, n_unknown, m_known, c2, c1):
        rows = B * n_unknown
        out = torch.empty((rows, self.n_last), dtype=torch.float32, device=known_feat.device)
        call(self._fn, 2, 0, *self._common(), rows, self.k0, None, 0,
             None, None, None, None, 0, 0, 0, 0,
             ptr(known_feat), ptr(idx3), ptr(w3), ptr(skip), n_unknown, m_known, c2, c1, ptr(out), self.n_last)
        return out

    def build_premul(self, c2, c1, x3=False):
        """Split and pack the first layer for pa_fp_chain_premul ONCE, at engine construction (on the constructing stream): nothing is
        packed lazily in the hot path, so pipeline streams never race a pack kernel issued on another stream."""
        wt0, b0, _, _, n0 = self.layers[0]
        dev = wt0.device
        w1a = wt0[:c2].contiguous()
        rest = self.layers[1:]
        m = len(rest)
        pk = lambda w: pack_weights_f16(w) if self.f16 else pack_weights(w)
        wskip = wt0[c2:c2 + c1].contiguous()
        self._premul = {
            "c2": c2, "c1": c1,
            "w1a": w1a, "w1a_p": pk(w1a), "zero": torch.zeros(n0, dtype=torch.float32, device=dev),
            "wskip": wskip, "bias0": b0, "n0": n0, "m": m,
            # coarser levels (c1 > 4): the skip part of the first layer is an MFMA layer of its own
            "wskip_p": pk(wskip) if (c1 > 4 and c1 % 4 == 0 and (self.f16 or n0 % 64 == 0)) else None,
            "wt": (ctypes.c_void_p * m)(*[l[0].data_ptr() for l in rest]),
            "wpk": (ctypes.c_void_p * m)(*[(p.data_ptr() if p is not None else None) for p in (self.packed16 if self.f16 else self.packed)[1:]]),
            "bias": (ctypes.c_void_p * m)(*[l[1].data_ptr() for l in rest]),
            "kpad": (ctypes.c_int * m)(*[l[3] for l in rest]), "nout": (ctypes.c_int * m)(*[l[4] for l in rest]),
        }
        # fp16 path, finest level's shape: fp16 table + LDS-shared weights (csrc/fpx_f16.hip); a function of the layer shapes only
        self._premul["g16"] = bool(self.f16 and c2 == 256 and n0 == 256 and 1 <= c1 <= 4 and m == 2 and all(l[3] == 256 and l[4] == 256 for l in rest)
                                   and os.environ.get("PA_ENGINE_FPX16", "1") != "0")
        # opt-in "f32x3" (model.mlp_dtype): the two 256 -> 256 layers of the finest level from (hi, lo) fp16 operand pairs (csrc/fpx_f32x3.hip)
        self._premul["x3"] = None
        if x3 and not self.f16 and c2 == 256 and n0 == 256 and 1 <= c1 <= 4 and m == 2 and all(l[3] == 256 and l[4] == 256 for l in rest):
            def hilo(w):
                s = int(torch.floor(torch.log2(1024.0 / w.abs().max().clamp_min(1e-30))).item())
                s = max(min(s, 24), -24)
                ws = w * (2.0 ** s)                                   # exact
                hi = ws.half().float()
                return torch.cat([pack_weights_f16(hi.contiguous()), pack_weights_f16((ws - hi).contiguous())]), 2.0 ** -s
            packed = [hilo(l[0]) for l in rest]
            bufs, inv = [b for b, _ in packed], [i for _, i in packed]
            w1a3, inv1a = hilo(w1a)
            self._premul["x3"] = {"bufs": bufs, "wq": (ctypes.c_void_p * 2)(*[b.data_ptr() for b in bufs]), "inv": (ctypes.c_float * 2)(*inv),
                                  "w1a": w1a3, "inv1a": inv1a}

    def fp_premul(self, known_feat, idx3, w3, skip, B, n_unknown, m_known, c2, c1, mark=None, out16=False):
        """Finest feature-propagation level: the first layer is applied to the m_known coarse points BEFORE interpolation
        (pa_fp_chain_premul; interpolation is linear), the skip (xyz) term is added in the kernel's prologue."""
        dev = known_feat.device
        if getattr(self, "_premul", None) is None or (self._premul["c2"], self._premul["c1"]) != (c2, c1):
            raise RuntimeError("fp_premul: build_premul(c2, c1) was not run for this level at engine construction")
        pm = self._premul
        if pm["g16"]:
            g16 = torch.empty((B * m_known, 256), dtype=torch.float16, device=dev)
            call("pa_fp_premul_g16", B * m_known, ptr(known_feat), c2, ptr(pm["w1a_p"]), ptr(g16))
            if mark is not None:
                mark()
            rows = B * n_unknown
            cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
            if out16:       # descriptor-only extraction: the map goes to the fp16 NetVLAD kernel and nowhere else -> fp16 rows, half the bytes both ways
                out = torch.empty((rows, self.n_last), dtype=torch.float16, device=dev)
                for _ in range(getattr(self, "bench_repeat", 1)):
                    call("pa_fp_chain_premul_g16h", pm["m"], cast(pm["wpk"]), cast(pm["bias"]), cast(pm["kpad"]), cast(pm["nout"]), rows, ptr(g16), ptr(idx3),
                         ptr(w3), ptr(skip), n_unknown, m_known, pm["n0"], c1, ptr(pm["wskip"]), ptr(pm["bias0"]), ptr(out))
                return out
            out = torch.empty((rows, self.n_last), dtype=torch.float32, device=dev)
            for _ in range(getattr(self, "bench_repeat", 1)):
                call("pa_fp_chain_premul_g16", pm["m"], cast(pm["wpk"]), cast(pm["bias"]), cast(pm["kpad"]), cast(pm["nout"]), rows, ptr(g16), ptr(idx3),
                     ptr(w3), ptr(skip), n_unknown, m_known, pm["n0"], c1, ptr(pm["wskip"]), ptr(pm["bias0"]), ptr(out), self.n_last)
            return out
        if pm.get("x3") is not None:      # opt-in f32x3: the pre-multiply in the same split-operand arithmetic
            g = torch.empty((B * m_known, pm["n0"]), dtype=torch.float32, device=dev)
            call("pa_linear_x3", B * m_known, ptr(known_feat), c2, ptr(pm["x3"]["w1a"]), ctypes.c_float(pm["x3"]["inv1a"]), ptr(g), pm["n0"])
        else:
            g = torch.empty((B * m_known, pm["n0"]), dtype=torch.float32, device=dev)
            call("pa_linear_f16" if self.f16 else "pa_linear", B * m_known, c2, pm["n0"], ptr(known_feat), c2, ptr(pm["w1a"]), ptr(pm["w1a_p"]),
                 ptr(pm["zero"]), 0, None, 0, ptr(g), pm["n0"])
        if mark is not None:
            mark()
        rows = B * n_unknown
        out = torch.empty((rows, self.n_last), dtype=torch.float32, device=dev)
        rest = self.layers[1:]
        if pm.get("x3") is not None:
            x3 = pm["x3"]
            cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
            for _ in range(getattr(self, "bench_repeat", 1)):
                call("pa_fp_chain_premul_x3", pm["m"], cast(x3["wq"]), cast(x3["inv"]), cast(pm["bias"]), rows, ptr(g), ptr(idx3), ptr(w3), ptr(skip),
                     n_unknown, m_known, pm["n0"], c1, ptr(pm["wskip"]), ptr(pm["bias0"]), ptr(out), self.n_last)
            return out
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
        for _ in range(getattr(self, "bench_repeat", 1)):     # > 1 only under profiling.launch_time_ms: the same launch back to back (idempotent)
            call("pa_fp_chain_premul_f16" if self.f16 else "pa_fp_chain_premul", pm["m"], cast(pm["wt"]), cast(pm["wpk"]), cast(pm["bias"]), cast(pm["kpad"]), cast(pm["nout"]), rows,
                 ptr(g), ptr(idx3), ptr(w3), ptr(skip), n_unknown, m_known, pm["n0"], c1, ptr(pm["wskip"]), ptr(pm["wskip_p"]), ptr(pm["bias0"]),
                 ptr(out), self.n_last)
        return out

    def plain(self, x):
        rows, k = x.shape
        out = torch.empty((rows, self.n_last), dtype=torch.float32, device=x.device)
        call(self._fn, 0, 0, *self._common(), rows, self.k0, ptr(x), k,
             None, None, None, None, 0, 0, 0, 0, None, None, None, None, 0, 0, 0, 0, ptr(out), self.n_last)
        return out


def fold_bn1d(bn):
    """BatchNorm1d (eval) as y*scale + shift, fp64 math."""
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    return scale, shift


class _Vlad:
    """One pyramid scale for pa_netvlad: BatchNorm folded into the K-major assignment weights."""

    def __init__(self, v, device):
        c, k = v.cluster_weights.shape
        self.c, self.k, self.n = c, k, v.max_samples
        kp = (k + 15) // 16 * 16
        scale, shift = fold_bn1d(v.bn1)
        wc = torch.zeros(c, kp, dtype=torch.float64)
        wc[:, :k] = v.cluster_weights.detach().double().cpu() * scale.cpu()[None, :]
        bias = torch.zeros(kp, dtype=torch.float64)
        bias[:k] = shift.cpu()
        self.wc_t = wc.float().contiguous().to(device)
        self.bias = bias.float().contiguous().to(device)
        self.w2 = v.cluster_weights2.detach()[0].float().contiguous().to(device)      # (C, K)
        self.wc_p = None
        if kp == 64:                                   # fragment order of the K = 64 assignment GEMM (vlad.hip)
            self.wc_p = torch.empty(c * kp, dtype=torch.float32, device=device)
            call("pa_netvlad_pack_weights", c, kp, ptr(self.wc_t), ptr(self.wc_p))

    def run(self, x, out, ldo, koff, rows=False):
        """rows=False: out (B, C, ldo) as the reference lays it out; rows=True: out (B, ldo, C), one contiguous row per cluster."""
        b = x.shape[0]
        nfl = _lib.lib().pa_netvlad_scratch_floats(b, self.n, self.k)
        scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
        if rows:
            call("pa_netvlad_rows", b, self.n, self.c, self.k, ptr(x), ptr(self.wc_t), ptr(self.wc_p), ptr(self.bias), ptr(self.w2),
                 ptr(scratch), ptr(out), ldo, koff)
        else:
            call("pa_netvlad", b, self.n, self.c, self.k, ptr(x), ptr(self.wc_t), ptr(self.bias), ptr(self.w2), ptr(scratch), ptr(out), ldo, koff)


class _Pyramid:
    """All scales of the NetVLAD pyramid through pa_netvlad_pyramid: the coarse scales share one accumulate launch and every scale
    shares one finalize launch (per-scale _Vlad.run: two launches each).  Pointer arrays of the weights are built once."""

    def __init__(self, vlads, f16=False):
        self.vlads = vlads
        ns = len(vlads)
        # fp16 path (model.mlp_dtype = "f16"): the 64-cluster scale on the fp16 MFMA (vlad_accum16_kernel); PA_ENGINE_VLAD_F16=0 = A/B knob
        self.f16 = bool(f16) and os.environ.get("PA_ENGINE_VLAD_F16", "1") != "0" and any(v.k > 48 and v.c == 256 for v in vlads)
        if self.f16:
            # (hi, lo) fp16 pairs of the assignment weights: the logits keep fp32 accuracy (vlad.hip)
            def hilo(w):
                hi = w.half().float()
                return torch.cat([pack_weights_f16(w), pack_weights_f16((w - hi).contiguous())])
            self._wc16 = [hilo(v.wc_t) if (v.k > 48 and v.c == 256) else None for v in vlads]
            self.wc16 = (ctypes.c_void_p * ns)(*[(t.data_ptr() if t is not None else None) for t in self._wc16])
        self.ns = ns
        self.ktot = sum(v.k for v in vlads)
        arr = lambda ts: (ctypes.c_void_p * ns)(*[(t.data_ptr() if t is not None else None) for t in ts])
        self.n = (ctypes.c_int * ns)(*[v.n for v in vlads])
        self.k = (ctypes.c_int * ns)(*[v.k for v in vlads])
        self.wc_t, self.wc_p = arr([v.wc_t for v in vlads]), arr([v.wc_p for v in vlads])
        self.bias, self.w2 = arr([v.bias for v in vlads]), arr([v.w2 for v in vlads])

        self.small = [v.k <= 16 for v in vlads]       # scales that share the early accumulate launch

    def begin(self, b, dev):
        """Per-forward state: the partial-sum scratch of every scale."""
        lib = _lib.lib()
        scr = [torch.empty(lib.pa_netvlad_scratch_floats(b, v.n, v.k), dtype=torch.float32, device=dev) for v in self.vlads]
        return {"b": b, "scr": scr, "sc": (ctypes.c_void_p * self.ns)(*[t.data_ptr() for t in scr])}

    def launch(self, state, feats, out, phases, x16_mask=0):
        """feats: per scale (B, n_s, 256) point-major, coarse -> fine (None where `phases` does not read the scale); phases as pa_netvlad_pyramid."""
        for v, f in zip(self.vlads, feats):
            if f is not None and (f.shape[1] != v.n or f.shape[2] != v.c):
                raise ValueError(f"NetVLAD scale built for ({v.n}, {v.c}) features, got {tuple(f.shape[1:])}")
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
        xs = (ctypes.c_void_p * self.ns)(*[(f.data_ptr() if f is not None else None) for f in feats])
        state["keep"] = feats                         # the launches read these buffers asynchronously
        if self.f16 and x16_mask:        # the flagged scales' maps are fp16 rows (fp_premul(out16=True))
            call("pa_netvlad_pyramid_f16h", state["b"], self.ns, cast(self.n), cast(self.k), cast(xs), cast(self.wc_t), cast(self.wc_p), cast(self.wc16),
                 cast(self.bias), cast(self.w2), cast(state["sc"]), ptr(out), phases, x16_mask)
            return
        if self.f16:
            call("pa_netvlad_pyramid_f16", state["b"], self.ns, cast(self.n), cast(self.k), cast(xs), cast(self.wc_t), cast(self.wc_p), cast(self.wc16),
                 cast(self.bias), cast(self.w2), cast(state["sc"]), ptr(out), phases)
            return
        call("pa_netvlad_pyramid", state["b"], self.ns, cast(self.n), cast(self.k), cast(xs), cast(self.wc_t), cast(self.wc_p), cast(self.bias),
             cast(self.w2), cast(state["sc"]), ptr(out), phases)

    def run(self, feats, out):
        """All phases at once: feats per scale (B, n_s, 256) point-major, coarse -> fine; out (B, ktot, 256)."""
        self.launch(self.begin(out.shape[0], out.device), feats, out, 7)


class _Afa:
    """AdaptiveFeatureAggregator for pa_afa: K-major FC weight, BatchNorm folded to scale/shift."""

    def __init__(self, afa, device):
        if len(afa.mlpa.mlps) != 1:
            raise ValueError("fused APFA supports the single-conv attention layer of the shipped configuration")
        self.watt = afa.mlpa.mlps[0].weight.detach().squeeze(-1).float().contiguous().to(device)   # (out, in)
        self.fc_wt = afa.fc.weight.detach().t().float().contiguous().to(device)                    # (C*K, nout) K-major
        self.fc_bias = afa.fc.bias.detach().float().contiguous().to(device)
        scale, shift = fold_bn1d(afa.bn)
        self.scale, self.shift = scale.float().contiguous().to(device), shift.float().contiguous().to(device)
        self.l2 = 1 if afa.l2_norm else 0
        self.nout = afa.fc.out_features
        # cluster-major variant (pa_afa_rows): attention conv as a K-major (in, out) matrix, FC rows re-ordered from c*K + k to k*C + c
        c = self.watt.shape[0]
        k = afa.fc.in_features // c
        self.watt_t = self.watt.t().contiguous()
        self.watt_p = pack_weights(self.watt_t)
        self.zero = torch.zeros(c, dtype=torch.float32, device=device)
        self.fc_wt_rows = afa.fc.weight.detach().float().view(self.nout, c, k).permute(2, 1, 0).reshape(k * c, self.nout).contiguous().to(device)

    def run_fused(self, vt):
        """vt (B, ktot, C) cluster-major rows -> desc: two launches (pa_afa_fused: per-cluster attention logits + FC partials, then the
        soft-max over clusters scales the partials).  Needs nout % 64 == 0."""
        b, ktot, c = vt.shape
        scratch = torch.empty(_lib.lib().pa_afa_fused_scratch_floats(b, ktot, self.nout), dtype=torch.float32, device=vt.device)
        desc = torch.empty((b, self.nout), dtype=torch.float32, device=vt.device)
        call("pa_afa_fused", b, c, ktot, self.nout, ptr(vt), ptr(self.watt_t), ptr(self.fc_wt_rows), ptr(self.fc_bias), ptr(self.scale),
             ptr(self.shift), self.l2, ptr(scratch), ptr(desc))
        return desc

    def run_rows(self, vt):
        """vt (B, ktot, C) from _Vlad.run(rows=True): the five-launch form (dense logits, re-weight, split-K FC, finalize)."""
        b, ktot, c = vt.shape
        nfl = _lib.lib().pa_afa_rows_scratch_floats(b, c, ktot, self.nout)
        scratch = torch.empty(nfl, dtype=torch.float32, device=vt.device)
        desc = torch.empty((b, self.nout), dtype=torch.float32, device=vt.device)
        call("pa_afa_rows", b, c, ktot, self.nout, ptr(vt), ptr(self.watt_t), ptr(self.watt_p), ptr(self.zero), ptr(self.fc_wt_rows),
             ptr(self.fc_bias), ptr(self.scale), ptr(self.shift), self.l2, ptr(scratch), ptr(desc))
        return desc

    def run(self, v):
        b, c, ktot = v.shape
        nfl = _lib.lib().pa_afa_scratch_floats(b, c, ktot, self.nout)
        scratch = torch.empty(nfl, dtype=torch.float32, device=v.device)
        desc = torch.empty((b, self.nout), dtype=torch.float32, device=v.device)
        call("pa_afa", b, c, ktot, self.nout, ptr(v), ptr(self.watt), ptr(self.fc_wt), ptr(self.fc_bias), ptr(self.scale),
             ptr(self.shift), self.l2, ptr(scratch), ptr(desc))
        return desc


class _Attn:
    """Grouped self-attention level of PPT-Net (pptnet.py:246-282) for pa_linear + pa_sa_attention + pa_linear.

    The tied grouped q/k conv is expanded to a dense block-diagonal (C, C) matrix and concatenated with v_conv into one
    K-major (C, 2C) weight; trans_conv + after_norm (eval) fold into one K-major (C, C) weight + bias."""

    def __init__(self, sa, device, f16=False):
        c, gp = sa.v_conv.weight.shape[0], sa.gp
        if c not in (64, 128, 256, 512):
            raise ValueError(f"fused attention is built for 64/128/256/512 channels, got {c}")
        cg = c // gp
        wq = sa.k_conv.weight.detach().double().squeeze(-1)                   # (C, C/gp): row o uses inputs of group o // cg
        dense = torch.zeros(c, c, dtype=torch.float64)
        for g in range(gp):
            dense[g * cg:(g + 1) * cg, g * cg:(g + 1) * cg] = wq[g * cg:(g + 1) * cg].cpu()
        wv = sa.v_conv.weight.detach().double().squeeze(-1).cpu()              # (C_out, C_in)
        self.wqv_t = torch.cat([dense.t(), wv.t()], dim=1).float().contiguous().to(device)                 # (C, 2C) K-major
        self.bqv = torch.cat([torch.zeros(c, dtype=torch.float64), sa.v_conv.bias.detach().double().cpu()]).float().contiguous().to(device)
        scale, shift = fold_bn1d(sa.after_norm)
        wt = sa.trans_conv.weight.detach().double().squeeze(-1).cpu() * scale.cpu()[:, None]
        self.wt_t = wt.t().float().contiguous().to(device)                                                   # (C, C) K-major
        self.bt = (sa.trans_conv.bias.detach().double().cpu() * scale.cpu() + shift.cpu()).float().contiguous().to(device)
        self.c = c
        self.wqv_p, self.wt_p = pack_weights(self.wqv_t), pack_weights(self.wt_t)
        # fp16 path (model.mlp_dtype = "f16", BASELINE configs[4]): the two dense layers on the fp16 chain kernel and both contractions of the
        # attention on fp16 MFMA (csrc/attention_f16.hip; widths it is built for); PA_ATTN_F16_SPLIT=0 drops the (hi, lo) energy operands (A/B knob)
        self.f16 = bool(f16) and c in (64, 128, 256)
        self.split = 0 if os.environ.get("PA_ATTN_F16_SPLIT", "1") == "0" else 1
        # the q/k/v projection feeds the soft-max logits (a relative 2^-11 there is an ABSOLUTE |e| 2^-11 in the exponent): it stays on
        # the exact fp32 kernel; PA_ATTN_F16_QV=1 is the A/B knob for the fp16 form
        self.qv16 = self.f16 and os.environ.get("PA_ATTN_F16_QV", "0") == "1"
        self.t16 = self.f16 and os.environ.get("PA_ATTN_F16_TRANS", "1") == "1"
        if self.f16:
            self.wqv_16, self.wt_16 = pack_weights_f16(self.wqv_t), pack_weights_f16(self.wt_t)
            # trans_conv + BatchNorm + ReLU + residual inside the attention's second pass (pa_sa_attention_trans_f16); PA_ATTN_F16_FUSE=0 = A/B knob
            self.fuse_t = self.t16 and os.environ.get("PA_ATTN_F16_FUSE", "1") == "1"
            if self.fuse_t:
                self.wt_fused = torch.empty(c * c, dtype=torch.float16, device=device)
                call("pa_sa_attention_f16_pack_trans", c, ptr(self.wt_t), ptr(self.wt_fused))

    def _fuse_here(self, n):
        """The layer behind the attention rides in the second pass's epilogue at the levels with many points and narrow features (PPT-Net's first
        two: 1024 x 64, 256 x 128; measured 171 -> 163 us and 99 -> 89 us (fp16) at the first).  At the coarse levels (64 x 256, 16 x 512) a wave would
        stream the whole (C x C) weight for its 16 points with a handful of workgroups on the chip (60 -> 210 us measured at 16 x 512): those keep the
        column-sliced pa_linear launch.  A rule on the level's shape, never on the batch."""
        return n >= 256 and self.c <= 128

    def run(self, x, B, n):
        """x (B*n, C) point-major -> x + relu(BN(trans_conv(x - x_r)))."""
        c, dev = self.c, x.device
        rows = B * n
        yv = torch.empty((rows, 2 * c), dtype=torch.float32, device=dev)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=dev)
        d = torch.empty((rows, c), dtype=torch.float32, device=dev)
        out = torch.empty((rows, c), dtype=torch.float32, device=dev)
        if self.f16:
            if self.qv16:
                call("pa_linear_f16", rows, c, 2 * c, ptr(x), c, ptr(self.wqv_t), ptr(self.wqv_16), ptr(self.bqv), 0, None, 0, ptr(yv), 2 * c)
            else:
                call("pa_linear", rows, c, 2 * c, ptr(x), c, ptr(self.wqv_t), ptr(self.wqv_p), ptr(self.bqv), 0, None, 0, ptr(yv), 2 * c)
            scratch = torch.empty(_lib.lib().pa_sa_attention_f16_scratch_halfs(B, n, c, self.split), dtype=torch.float16, device=dev)
            if self.fuse_t and self._fuse_here(n):
                call("pa_sa_attention_trans_f16", B, n, c, self.split, ptr(yv), ptr(x), ptr(scratch), ptr(stats), ptr(self.wt_fused), ptr(self.bt), ptr(out))
                return out
            call("pa_sa_attention_f16", B, n, c, self.split, ptr(yv), ptr(x), ptr(scratch), ptr(stats), ptr(d))
            if self.t16:
                call("pa_linear_f16", rows, c, c, ptr(d), c, ptr(self.wt_t), ptr(self.wt_16), ptr(self.bt), 1, ptr(x), c, ptr(out), c)
            else:
                call("pa_linear", rows, c, c, ptr(d), c, ptr(self.wt_t), ptr(self.wt_p), ptr(self.bt), 1, ptr(x), c, ptr(out), c)
            return out
        call("pa_linear", rows, c, 2 * c, ptr(x), c, ptr(self.wqv_t), ptr(self.wqv_p), ptr(self.bqv), 0, None, 0, ptr(yv), 2 * c)
        if os.environ.get("PA_ATTN_FUSE", "1") == "1" and self._fuse_here(n):      # trans_conv + BatchNorm + ReLU + residual in the second pass's epilogue (A/B knob: 0)
            call("pa_sa_attention_trans", B, n, c, ptr(yv), ptr(x), ptr(stats), ptr(self.wt_t), ptr(self.bt), ptr(out))
            return out
        call("pa_sa_attention", B, n, c, ptr(yv), ptr(x), ptr(stats), ptr(d))
        call("pa_linear", rows, c, c, ptr(d), c, ptr(self.wt_t), ptr(self.wt_p), ptr(self.bt), 1, ptr(x), c, ptr(out), c)
        return out


class _FcHead:
    """Fully connected aggregation head on the cluster-major VLAD rows (B, sum K, 256): FC -> BatchNorm1d (eval) [-> L2 normalise].

    per_scale=True : PPT-Net (pptnet_origin/models/loupe.py:94-105): every scale is flattened C-major on its own and the blocks are
                     concatenated, so the reference's FC row of (scale i, channel c, cluster k) is off_i + c*K_i + k;
    per_scale=False: PatchAugNet aggregation_type 0 (patch_aug_net/models/loupe.py:298-300): the concatenated (B, C, sum K) tensor is
                     flattened C-major, row c*sum K + koff_i + k.
    Either way the rows are permuted ONCE to the kernel's order (koff_i + k)*256 + c."""

    def __init__(self, hidden_weights, bn, ks, per_scale, l2, device):
        c, ktot = 256, sum(ks)
        hw = hidden_weights.detach()
        perm = torch.empty(c * ktot, dtype=torch.long)
        off, koff = 0, 0
        for k in ks:
            cc, kk = torch.meshgrid(torch.arange(c), torch.arange(k), indexing="ij")
            src = (off + cc * k + kk) if per_scale else (cc * ktot + koff + kk)
            perm[((koff + kk) * c + cc).flatten()] = src.flatten()
            off += c * k
            koff += k
        self.fc_wt = hw[perm.to(hw.device)].float().contiguous().to(device)                 # (256*ktot, nout) K-major
        scale, shift = fold_bn1d(bn)
        self.scale, self.shift = scale.float().contiguous().to(device), shift.float().contiguous().to(device)
        self.nout = hw.shape[1]
        if self.nout % 16:
            raise ValueError("fused FC head needs an output width that is a multiple of 16")
        self.l2 = l2

    def run(self, v):
        b, kdim, dev = v.shape[0], v.shape[1] * v.shape[2], v.device
        scratch = torch.empty(_lib.lib().pa_fc_scratch_floats(b, kdim, self.nout), dtype=torch.float32, device=dev)
        h = torch.empty((b, self.nout), dtype=torch.float32, device=dev)
        call("pa_fc", b, kdim, self.nout, ptr(v), ptr(self.fc_wt), None, ptr(self.scale), ptr(self.shift), self.l2, None, ptr(scratch), ptr(h))
        return h


class _Gate:
    """GatingContext (loupe.py:332-361): x * sigmoid(BN(x @ W)) [-> L2 normalise], one pa_fc launch pair."""

    def __init__(self, g, l2, device):
        self.g_wt = g.gating_weights.detach().float().contiguous().to(device)               # (dim, dim): x @ W, already K-major
        gs, gh = fold_bn1d(g.bn1)
        self.g_scale, self.g_shift = gs.float().contiguous().to(device), gh.float().contiguous().to(device)
        self.dim = self.g_wt.shape[0]
        if self.dim % 16:
            raise ValueError("fused context gating needs a width that is a multiple of 16")
        self.l2 = l2

    def run(self, h):
        b, dev = h.shape[0], h.device
        scratch = torch.empty(_lib.lib().pa_fc_scratch_floats(b, self.dim, self.dim), dtype=torch.float32, device=dev)
        out = torch.empty_like(h)
        call("pa_fc", b, self.dim, self.dim, ptr(h), ptr(self.g_wt), None, ptr(self.g_scale), ptr(self.g_shift), self.l2, ptr(h), ptr(scratch), ptr(out))
        return out


class PatchAugNetEngine:
    """Fused evaluation plan for both model families (PatchAugNet: 3 levels + APFA head; PPT-Net: 4 levels with grouped
    self-attention + FC/gating head).  Reads the parameters of the nn.Module tree; holds folded copies."""

    def __init__(self, model, device):
        self.device = torch.device(device)
        cfg = model.param
        self.sampling, self.knn = list(cfg["SAMPLING"]), list(cfg["KNN"])
        self.use_origin = cfg.get("USE_ORIGIN_PC_IN_FP", True)
        bb = model.backbone
        # "f32" (default): exact fp32 MFMA.  "f16": the shared-MLP chains run on fp16 MFMA with fp32 accumulation (model.mlp_dtype or
        # PA_ENGINE_MLP_DTYPE); sampling, grouping indices, attention, NetVLAD and the heads stay fp32.
        self.mlp_dtype = getattr(model, "mlp_dtype", None) or os.environ.get("PA_ENGINE_MLP_DTYPE", "f32")
        # "f32x3" (opt-in, never a default): everything as "f32" except the two 256 -> 256 layers of the finest FP level, whose products are
        # evaluated from (hi, lo) fp16 operand pairs on the fp16 MFMA (~2^-21 relative per product; csrc/fpx_f32x3.hip).
        if self.mlp_dtype not in ("f32", "f16", "f32x3"):
            raise ValueError("mlp_dtype must be 'f32', 'f16' or 'f32x3'")
        f16 = self.mlp_dtype == "f16"
        if any(len(m.mlps) != 1 for m in bb.SA_modules):
            raise ValueError("fused engine: a multi-scale-grouping level (backbone.SAModuleMSG with several scales) runs on the module path; "
                             "set model.fused_eval = False")
        with torch.no_grad():
            self.sa = [_Chain(fold_shared_mlp(m.mlps[0], self.device), f16) for m in bb.SA_modules]
            self.fp = [_Chain(fold_shared_mlp(m.mlp, self.device), f16) for m in bb.FP_modules]
            self.attn = [_Attn(m.sas[0], self.device, f16) if hasattr(m, "sas") else None for m in bb.SA_modules]
        self.agg = model.aggregation
        agg = self.agg
        self.ppt = hasattr(agg, "vlad0")
        vl = [getattr(agg, f"vlad{i}") for i in range(4)] if self.ppt else list(agg.vlads)
        if not (all(v.feature_size == 256 and v.cluster_size <= 64 for v in vl) and sum(v.cluster_size for v in vl) <= 256):
            # no PyTorch fallback in the product path: shapes the HIP head kernels are not built for are refused, loudly
            raise ValueError("fused engine: the NetVLAD / aggregation kernels are built for 256-wide features, <= 64 clusters per scale and "
                             "<= 256 clusters in total; set model.fused_eval = False to run the autograd module path instead")
        ks = [v.cluster_size for v in vl]
        with torch.no_grad():
            self.vlads = [_Vlad(v, self.device) for v in vl]
            self.pyramid = _Pyramid(self.vlads, f16=f16) if os.environ.get("PA_ENGINE_VLAD_PER_SCALE") is None else None     # A/B knob
            self._vlad_early = os.environ.get("PA_ENGINE_VLAD_LATE") is None                                       # A/B knob
            self._atomic_pool = os.environ.get("PA_ENGINE_NO_ATOMIC_POOL") is None                                  # A/B knob
            self.afa = self.head = self.gate = None
            if self.ppt:
                self.head_kind = "fc"
                self.head = _FcHead(agg.hidden_weights, agg.bn2, ks, per_scale=True, l2=0 if agg.gating else (1 if model.use_normalize else 0), device=self.device)
                if agg.gating:
                    self.gate = _Gate(agg.context_gating, l2=1 if model.use_normalize else 0, device=self.device)
            else:
                if agg.aggregation_type == 2:
                    if len(agg.afa.mlpa.mlps) != 1 or agg.afa.fc.out_features % 16:
                        raise ValueError("fused APFA head supports the single-conv attention layer and an output width that is a multiple of 16")
                    self.head_kind = "afa"
                    self.afa = _Afa(agg.afa, self.device)
                    # pa_afa_fused is built for <= 256 clusters in total and <= 1024 outputs (vlad.hip); anything else takes the five-launch head
                    self._afa_fused = (self.afa.nout % 64 == 0 and self.afa.nout <= 1024 and sum(ks) <= 256
                                       and os.environ.get("PA_ENGINE_AFA_ROWS") is None)      # A/B knob: the five-launch head
                elif agg.aggregation_type == 0:       # loupe.py:298-300: FC over the C-major flattening of (B, C, sum K), BN, L2 normalise
                    self.head_kind = "fc"
                    self.head = _FcHead(agg.hidden_weights, agg.bn, ks, per_scale=False, l2=1, device=self.device)
                else:                                 # loupe.py:304-306: max over the clusters, L2 normalise
                    self.head_kind = "max"
                if agg.gating:                        # loupe.py:308-309, applied after the normalisation
                    self.gate = _Gate(agg.context_gating, l2=0, device=self.device)
        self.premul = os.environ.get("PA_ENGINE_NO_PREMUL") is None     # fold the FP levels' first layer through the interpolation
        # Channel counts of every FP level are fixed by the architecture: known = the level above (coarsest: the last SA level), skip = the
        # encoder features of the same level (finest: xyz).  The first-layer split of pa_fp_chain_premul is packed HERE, never in a forward.
        nfp = len(self.fp)
        self._fold_static = []
        with torch.no_grad():
            for j, chain in enumerate(self.fp):
                c2 = self.fp[j + 1].n_last if j + 1 < nfp else self.sa[-1].n_last
                c1 = (3 if self.use_origin else 0) if j == 0 else self.sa[j - 1].n_last
                ok = (1 <= c1 <= 4 or (c1 > 4 and c1 % 4 == 0 and chain.n <= 3 and (not chain.f16 or chain.layers[0][4] % 32 == 0)))
                ok = bool(self.premul and ok and chain.n >= 2 and c2 % 4 == 0 and chain.layers[0][4] % 16 == 0 and chain.layers[0][2] == c2 + c1)
                self._fold_static.append(ok)
                if ok:
                    chain.build_premul(c2, c1, x3=self.mlp_dtype == "f32x3")
        self._tensors = list(model.parameters()) + list(model.buffers())
        self._key = self._params_key(model)
        self.timer = None     # optional profiling.StageTimer: per-stage HIP-event marks (bench.py kernel attribution)
        # latency mode (model.geo_overlap = True or PA_ENGINE_GEO_OVERLAP=1): coordinate-only kernels of the coarser levels on a side stream.
        # Single-stream step at B = 32: 1.78 -> 1.68 ms, at B = 1: 1.25 -> 1.16 ms.  NOT for the throughput pipeline: a captured graph with
        # a forked branch serialises against the other streams' graphs at replay (34.5 k -> 18.7 k submaps/s measured), so it is never
        # applied while a stream is capturing.
        self.geo_overlap = os.environ.get("PA_ENGINE_GEO_OVERLAP") is not None
        # keep_geometry = True: backbone() leaves its level-local centre / neighbour indices and the set-abstraction features in
        # self.last_geometry (parity tests read the reference's sample_idx_origin / sa_features from it, patch_aug_net.py:169-177)
        self.keep_geometry = False
        self.last_geometry = None
        # fp16 path: forward(views=False) keeps the finest feature map in fp16 between its producer and the NetVLAD kernel (PA_ENGINE_FP0_F16=0: A/B knob)
        self._fp0_half_ok = os.environ.get("PA_ENGINE_FP0_F16", "1") != "0"
        self._fp0_half = False
        self._geo_streams = {}

    @staticmethod
    def _presort_ok(n, m, ns):
        """The first level's neighbour search takes a pre-sorted cloud (pa_knnquery_presorted: the cell-grid kernel's shapes).  OPT-IN
        (PA_ENGINE_PRESORT=1): measured on MI355X at batch 32, four streams, it is a wash (36.9 k vs 36.7 k submaps/s, inside the run-to-run
        spread; profiles/r04_ab_log.txt) -- the sort the eight workgroups of a cloud repeat is hidden under the other streams' work, while the
        extra launch sits in front of the sampling chain on the step's own stream."""
        return (2048 <= n <= 4096 and m >= 256 and ns in (16, 20, 32) and os.environ.get("PA_ENGINE_PRESORT") == "1"
                and os.environ.get("PA_KNN_NO_QUAD") is None)

    def _first_level_chunks(self, npts, ns):
        """Sample ranges [j0, j1, ..., m] of the first level's sampling in latency mode (None = one launch): the level must run the kernels that take windows --
        the cell-grid kNN (1024..4096 source points, >= 128 centres, 16 / 20 / 32 neighbours; the ranges / windows here only at the first level's 2048..4096 / >= 256) and the persistent first-level chain
        (<= 8 -> 32 -> 32 -> 64, fp32, no attention in between) -- rules on the architecture's shapes, never on the batch."""
        ch = self.sa[0]
        ok = (len(self.sa) > 1 and 2048 <= npts[0] <= 4096 and npts[1] >= 1024 and npts[1] % 16 == 0 and ns in (16, 20, 32) and 13 <= ns <= 20 and not ch.f16
              and self.attn[0] is None and [l[4] for l in ch.layers] == [32, 32, 64] and ch.layers[0][3] == 8
              and os.environ.get("PA_ENGINE_NO_FPS_CHUNKS") is None)
        # OPT-IN (PA_ENGINE_FPS_CHUNKS = cut points, e.g. "896" or "256,512,768"): measured on MI355X (profiles/r04_latency_mode.txt) every variant
        # LOSES to the one-launch form of the same latency mode -- batch 1 / 8 / 32: 1.13 / 1.24 / 1.68 ms in one launch, 1.18 / 1.26 / 1.70 ms with
        # one cut at 896, 1.21 / 1.29 / 1.72 ms with three cuts -- each extra sampling launch and cross-stream hand-over costs ~20 us, about what the
        # tail it takes off the critical path is worth (the first level's neighbour search + chain are ~35 us at batch 1, ~115 us at batch 32, and
        # a window's search still pays the whole cloud sort).  The sampling chain itself is what the latency is made of.
        cuts = os.environ.get("PA_ENGINE_FPS_CHUNKS")
        if not ok or not cuts:
            return None
        inner = [int(v) for v in cuts.split(",")]
        return [0] + [c for c in inner if 0 < c < npts[1]] + [npts[1]]

    def _mark(self, name):
        if self.timer is not None:
            self.timer.mark(name)

    def _params_key(model_or_self, model=None):
        """Identity + version of EVERY parameter and buffer (partial load_state_dict, p.data.copy_(), edited BatchNorm statistics all
        bump a tensor's _version or move its storage); ~30 us per forward."""
        if model is None:                              # called as a static helper on a model without an engine
            model = model_or_self
            ts = list(model.parameters()) + list(model.buffers())
        else:
            ts = model_or_self._tensors
        ver = ptrs = 0
        for t in ts:
            ver += t._version
            ptrs ^= t.data_ptr()
        return (ts[0].device, len(ts), ver, ptrs, getattr(model, "mlp_dtype", None))

    def matches(self, model, x):
        return x.device == self.device and self._key == self._params_key(model)

    def stale(self, model):
        return self._key != self._params_key(model)

    def sample_first_level(self, xyz, cidx0, nxyz0):
        """The first level's farthest-point sampling alone, into caller-owned buffers ((B, m) int32 indices, (B, m, 3) centres): the
        launch extract.GraphedExtractor puts on its sampling streams; hand the buffers to forward(..., s0=(cidx0, nxyz0))."""
        call("pa_furthestsampling_gather", xyz.shape[0], xyz.shape[1], self.sampling[0], ptr(xyz), ptr(cidx0), ptr(nxyz0))

    def geometry_buffers(self, clouds, points, device=None):
        """Caller-owned buffers for everything of `clouds` clouds that depends on the coordinates only: per set-abstraction level the centre
        indices, centre coordinates and neighbour lists, per feature-propagation level the 3-NN indices and weights.  compute_geometry fills them,
        forward(..., geo=a per-batch slice) consumes them in place (extract.SampledAheadExtractor computes a whole group of batches ahead)."""
        dev = self.device if device is None else device
        L, nfp = len(self.sa), len(self.fp)
        npts = [points] + list(self.sampling[:L])
        off = L - nfp
        return {"cidx": [torch.empty((clouds, npts[i + 1]), dtype=torch.int32, device=dev) for i in range(L)],
                "nxyz": [torch.empty((clouds, npts[i + 1], 3), dtype=torch.float32, device=dev) for i in range(L)],
                "nbr": [torch.empty((clouds, npts[i + 1], self.knn[i]), dtype=torch.int32, device=dev) for i in range(L)],
                "d2": [torch.empty((clouds, npts[i + 1], self.knn[i]), dtype=torch.float32, device=dev) for i in range(L)],
                "w3": [torch.empty((clouds, npts[j + off], 3), dtype=torch.float32, device=dev) for j in range(nfp)],
                "idx3": [torch.empty((clouds, npts[j + off], 3), dtype=torch.int32, device=dev) for j in range(nfp)]}

    @staticmethod
    def geometry_slice(geo, lo, hi):
        """The buffers of clouds lo .. hi - 1 of a geometry_buffers() set (views: the batch's graphs read them in place)."""
        return {k: [t[lo:hi] for t in v] for k, v in geo.items()}

    def compute_geometry(self, xyz, geo, first_level_only=False, samplings_only=False):
        """Sampling, centre gather, neighbour search of every level and the 3-NN weights of every decoder level for xyz (clouds, N, 3) into `geo`
        (geometry_buffers(clouds)), on the current stream: exactly the launches backbone() would issue between its chains, for any number of clouds at once."""
        B = xyz.shape[0]
        L, nfp = len(self.sa), len(self.fp)
        npts = [xyz.shape[1]] + list(self.sampling[:L])
        off = L - nfp
        l_xyz = [xyz] + [t[:B] for t in geo["nxyz"]]
        for i in range(L):
            call("pa_furthestsampling_gather", B, npts[i], npts[i + 1], ptr(l_xyz[i]), ptr(geo["cidx"][i]), ptr(geo["nxyz"][i]))
            if first_level_only:
                return
            if not samplings_only:
                call("pa_knnquery", B, npts[i], npts[i + 1], self.knn[i], ptr(l_xyz[i]), ptr(geo["nxyz"][i]), ptr(geo["nbr"][i]), ptr(geo["d2"][i]))
        if samplings_only:
            return
        for j in range(nfp - 1, -1, -1):
            call("pa_three_nn_weights", B, npts[j + off], npts[j + off + 1], ptr(l_xyz[j + off]), ptr(l_xyz[j + off + 1]), ptr(geo["w3"][j]), ptr(geo["idx3"][j]))

    def backbone(self, xyz, early=None, s0=None, geo=None):
        """xyz (B, N, 3) -> point-major features per level + level-0 centre indices.  early(l_feat): called once the decoder has written every
        level but the finest (the engine issues the coarse NetVLAD scales there).

        Everything that depends on coordinates only -- sampling, kNN and 3-NN of every level -- is independent of the feature chains.
        Level 0 (the long one: 1024 serial FPS rounds) has to come first; the coarser levels' sampling / kNN and all three 3-NN launches
        (0.17 ms of small, latency-bound kernels at B = 32) can then run on a side stream under the first set-abstraction chain instead
        of between the chains (fork / join with events).  Latency mode only (self.geo_overlap, see __init__); off while a stage timer is
        attached (profiling wants one stream) and while the stream is being captured into a hipGraph."""
        B = xyz.shape[0]
        dev = self.device
        L, nfp = len(self.sa), len(self.fp)
        npts = [xyz.shape[1]] + list(self.sampling[:L])
        # outputs of the geometry kernels, allocated on the main stream before any fork
        off = L - nfp                              # FP level j interpolates level j + off + 1's features onto level j + off's points
        searched = geo is not None and "nbr" in geo      # geo may hold the samplings of every level only (compute_geometry(samplings_only=True)): the searches then run here
        if geo is not None:                        # computed ahead (compute_geometry): read in place
            cidx, nxyz = list(geo["cidx"]), list(geo["nxyz"])
        else:
            cidx = [torch.empty((B, npts[i + 1]), dtype=torch.int32, device=dev) for i in range(L)]
            nxyz = [torch.empty((B, npts[i + 1], 3), dtype=torch.float32, device=dev) for i in range(L)]
        if searched:
            nbr, d2, w3, idx3 = (list(geo[k]) for k in ("nbr", "d2", "w3", "idx3"))
        else:
            nbr = [torch.empty((B, npts[i + 1], self.knn[i]), dtype=torch.int32, device=dev) for i in range(L)]
            d2 = [torch.empty((B, npts[i + 1], self.knn[i]), dtype=torch.float32, device=dev) for i in range(L)]
            w3 = [torch.empty((B, npts[j + off], 3), dtype=torch.float32, device=dev) for j in range(nfp)]
            idx3 = [torch.empty((B, npts[j + off], 3), dtype=torch.int32, device=dev) for j in range(nfp)]
        if s0 is not None:                         # first level already sampled (sample_first_level on another stream)
            cidx[0], nxyz[0] = s0
        l_xyz = [xyz] + nxyz

        def fps(i):
            if geo is None:
                call("pa_furthestsampling_gather", B, npts[i], npts[i + 1], ptr(l_xyz[i]), ptr(cidx[i]), ptr(nxyz[i]))

        # first level: the input cloud's cell sort does not depend on the centres, so it is issued BEFORE the sampling chain (one workgroup per
        # cloud, ~15 us, next to nothing in CU-time); the neighbour search then copies the record instead of sorting in each of its workgroups
        cells = None
        if not searched and self._presort_ok(npts[0], npts[1], self.knn[0]):
            cells = torch.empty(_lib.lib().pa_cloud_cellsort_floats(B, npts[0]), dtype=torch.float32, device=dev)
            call("pa_cloud_cellsort", B, npts[0], ptr(xyz), ptr(cells))

        def knn(i):
            if searched:
                return
            if i == 0 and cells is not None:
                call("pa_knnquery_presorted", B, npts[0], npts[1], self.knn[0], ptr(xyz), ptr(nxyz[0]), ptr(cells), ptr(nbr[0]), ptr(d2[0]))
                return
            call("pa_knnquery", B, npts[i], npts[i + 1], self.knn[i], ptr(l_xyz[i]), ptr(nxyz[i]), ptr(nbr[i]), ptr(d2[i]))

        def tnn(j):      # patch_aug_net.py:350-353
            if searched:
                return
            call("pa_three_nn_weights", B, npts[j + off], npts[j + off + 1], ptr(l_xyz[j + off]), ptr(l_xyz[j + off + 1]), ptr(w3[j]), ptr(idx3[j]))

        overlap = self.geo_overlap and self.timer is None and L > 1 and geo is None and not torch.cuda.is_current_stream_capturing()
        ev_sa, ev_fp = [None] * L, [None] * nfp
        main = torch.cuda.current_stream(dev)
        # Latency mode, first level in CHUNKS: the sampling order is prefix-stable, so the first quarter of the centres is final when a quarter
        # of the rounds has run -- its neighbour search and its set-abstraction chain run on a second stream under the NEXT quarter's sampling
        # (pa_furthestsampling_range / pa_knnquery_window / pa_sa_group_window).  Only when the level runs the kernels that take windows
        # (the 4096-point configurations of both models); bit-identical to the one-launch form.
        y0 = None
        chunks = self._first_level_chunks(npts, self.knn[0]) if overlap and s0 is None else None
        if chunks:
            side = self._geo_streams.get(("sa0", main.cuda_stream))
            if side is None:
                side = self._geo_streams[("sa0", main.cuda_stream)] = torch.cuda.Stream(device=dev)
            temp = torch.empty((B, npts[0]), dtype=torch.float32, device=dev)
            y0 = torch.empty((B * npts[1], self.sa[0].n_last), dtype=torch.float32, device=dev)
            for j0, j1 in zip(chunks[:-1], chunks[1:]):
                call("pa_furthestsampling_range", B, npts[0], npts[1], j0, j1, ptr(xyz), ptr(temp), ptr(cidx[0]), ptr(nxyz[0]))
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    call("pa_knnquery_window", B, npts[0], npts[1], self.knn[0], j0, j1 - j0, ptr(xyz), ptr(nxyz[0]), ptr(nbr[0]), ptr(d2[0]))
                    self.sa[0].sa(xyz, xyz, cidx[0], nbr[0], 3, pooled=True, window=(j1 - j0, j0), out=y0)
            ev_y0 = torch.cuda.Event()
            ev_y0.record(side)
            for t in (temp, y0, cidx[0], nxyz[0], nbr[0], d2[0], xyz):
                t.record_stream(side)
        elif s0 is None:
            fps(0)
        self._mark("sa0.fps")
        if overlap:
            gstream = self._geo_streams.get(main.cuda_stream)
            if gstream is None:
                gstream = self._geo_streams[main.cuda_stream] = torch.cuda.Stream(device=dev)
            ev0 = torch.cuda.Event()
            ev0.record(main)
            gstream.wait_event(ev0)
            with torch.cuda.stream(gstream):
                for i in range(1, L):
                    fps(i)
                    knn(i)
                    ev_sa[i] = torch.cuda.Event()
                    ev_sa[i].record(gstream)
                for j in range(nfp - 1, -1, -1):           # coarsest first: the order the FP chains consume them
                    tnn(j)
                    ev_fp[j] = torch.cuda.Event()
                    ev_fp[j].record(gstream)
        l_feat, l_c, c_feat = [xyz], [], 3
        for i, chain in enumerate(self.sa):
            src = l_xyz[i]
            m, ns = npts[i + 1], self.knn[i]
            if i == 0 and y0 is not None:          # chunked first level: neighbour search and chain already ran (side stream)
                main.wait_event(ev_y0)
                l_feat.append(y0.view(B, m, chain.n_last))
                l_c.append(cidx[0])
                c_feat = chain.n_last
                continue
            if i == 0:
                knn(0)
            elif overlap:
                main.wait_event(ev_sa[i])
            else:
                fps(i)
                self._mark(f"sa{i}.fps")
                knn(i)
            self._mark(f"sa{i}.knn")
            feat = l_feat[i]
            if chain.pooled_ok(B * m, ns):
                y = chain.sa(src, feat, cidx[i], nbr[i], c_feat, pooled=True)                      # (B*m, C')
            elif chain.atomic_pool_ok(B * m, ns) and self._atomic_pool:
                # wide hidden layers, few groups: 16-row tiles in group order with the max folded into the last layer's epilogue (atomics)
                y = chain.sa(src, feat, cidx[i], nbr[i], c_feat, pooled=2)
            else:  # rows in group order, then the max over each group's ns rows
                full = chain.sa(src, feat, cidx[i], nbr[i], c_feat, pooled=False)                  # (B*m*ns, C')
                y = torch.empty((B * m, chain.n_last), dtype=torch.float32, device=self.device)
                call("pa_rowgroup_max", B * m, ns, chain.n_last, ptr(full), ptr(y))
            self._mark(f"sa{i}.chain")
            if self.attn[i] is not None:
                y = self.attn[i].run(y, B, m)
                self._mark(f"sa{i}.attn")
            l_feat.append(y.view(B, m, chain.n_last))
            l_c.append(cidx[i])
            c_feat = chain.n_last
        sa_feat = l_feat[1:]                       # the list entries are replaced (not written) by the decoder levels below
        for i in range(-1, -(nfp + 1), -1):
            chain = self.fp[nfp + i]
            unknown, known = l_xyz[i - 1], l_xyz[i]
            n_u, m_k = unknown.shape[1], known.shape[1]
            if overlap:
                main.wait_event(ev_fp[nfp + i])
            else:
                tnn(nfp + i)
            w3_l, idx3_l = w3[nfp + i], idx3[nfp + i]
            self._mark(f"fp{nfp + i}.3nn")
            skip = l_feat[i - 1]
            if i == -nfp and not self.use_origin:
                skip = None
            known_feat = l_feat[i]
            c2 = known_feat.shape[-1]
            c1 = skip.shape[-1] if skip is not None else 0
            # c1 <= 4 (xyz skip): always; wider skips only at levels with enough points per cloud to amortise the extra pre-multiply
            # launch (a per-level rule, NOT a function of the batch size: results must not depend on how clouds are batched)
            if self._fold_static[nfp + i] and (c1 <= 4 or n_u >= 512) and n_u >= 2 * m_k:
                y = chain.fp_premul(known_feat.contiguous(), idx3_l, w3_l, skip.contiguous(), B, n_u, m_k, c2, c1,
                                    mark=lambda k=nfp + i: self._mark(f"fp{k}.premul"),
                                    out16=bool(self._fp0_half and nfp + i == 0 and chain._premul["g16"]))
            else:
                y = chain.fp(known_feat.contiguous(), idx3_l, w3_l, skip.contiguous() if skip is not None else None, B, n_u, m_k, c2, c1)
            self._mark(f"fp{nfp + i}.chain")
            l_feat[i - 1] = y.view(B, n_u, chain.n_last)
            if early is not None and nfp + i == 1:
                early(l_feat)        # every decoder level but the finest exists: the coarse NetVLAD scales read them while they are cache-resident
        if self.keep_geometry:
            self.last_geometry = {"center_idx": list(cidx), "sample_idx": list(nbr), "sa_features": list(sa_feat)}
        return l_feat, l_c

    def forward(self, x, views=True, s0=None, geo=None):
        """-> desc (B, 256), (fp_features views, level-0 centre indices); views=False skips the index mapping (descriptor-only callers).
        s0: (indices, centres) of the first level when sample_first_level already ran for this x."""
        if x.device.index is not None and x.device.index != torch.cuda.current_device():
            with torch.cuda.device(x.device):        # the C ABI launches on the CURRENT device's stream: follow the tensor
                return self._forward(x, views, s0, geo)
        return self._forward(x, views, s0, geo)

    def _forward(self, x, views, s0=None, geo=None):
        xyz = x.squeeze(1).contiguous()
        self._mark("start")
        nfp = len(self.fp)
        ktot = sum(v.k for v in self.vlads)
        v = torch.empty((x.shape[0], ktot, 256), dtype=torch.float32, device=self.device)     # cluster-major rows
        pyr = self.pyramid
        # coarse scales right behind the decoder level that completes their inputs (their feature maps are still cache-resident; after the
        # finest level's 134 MB they are not) -- when every scale but the finest is a <= 16-cluster scale, i.e. both shipped models
        split = pyr is not None and self.timer is None and nfp >= 2 and all(pyr.small[:-1]) and not pyr.small[-1] and self._vlad_early
        st = pyr.begin(x.shape[0], self.device) if pyr is not None else None

        def early(l_feat):
            coarse = [l_feat[j].contiguous() for j in range(nfp - 1, 0, -1)]
            pyr.launch(st, coarse + [None], v, 1)
        # fp16 path, descriptor-only call: the finest map is read by the fp16 NetVLAD kernel alone -> written (and read) as fp16 rows
        fine = self.vlads[-1] if self.vlads else None
        self._fp0_half = bool(not views and pyr is not None and pyr.f16 and self._fp0_half_ok and fine is not None and fine.k > 48 and fine.c == 256
                              and xyz.shape[1] >= 2048 and (getattr(self.fp[0], "_premul", None) or {}).get("g16"))
        l_feat, l_c = self.backbone(xyz, early if split else None, s0, geo)
        half_map = self._fp0_half and l_feat[0].dtype == torch.float16
        self._fp0_half = False
        feats = [l_feat[j] for j in range(nfp - 1, -1, -1)]                                   # coarse -> fine, (B, N_i, 256)
        if pyr is not None:
            fc = [f.contiguous() for f in feats]
            pyr.launch(st, ([None] * (nfp - 1) + fc[-1:]) if split else fc, v, 6 if split else 7, x16_mask=(1 << (nfp - 1)) if half_map else 0)
        else:
            koff = 0
            for vl, f in zip(self.vlads, feats):
                vl.run(f.contiguous(), v, ktot, koff, rows=True)
                koff += vl.k
        self._mark("vlad")
        if self.head_kind == "afa":
            desc = self.afa.run_fused(v) if self._afa_fused else self.afa.run_rows(v)
        elif self.head_kind == "fc":
            desc = self.head.run(v)
        else:
            desc = torch.empty((x.shape[0], 256), dtype=torch.float32, device=self.device)
            call("pa_vlad_maxpool", x.shape[0], ktot, 256, ptr(v), 1, ptr(desc))
        if self.gate is not None:
            desc = self.gate.run(desc)
        self._mark("afa")
        return desc, (self._views(feats, l_c) if views else (None, None))

    @staticmethod
    def _views(feats, l_c):
        c_o = [l_c[0]]
        for i in range(1, len(l_c)):
            c_o.append(torch.gather(c_o[i - 1], -1, l_c[i].long()))
        return [f.transpose(1, 2).unsqueeze(-1) for f in feats], c_o                         # (B, 256, N_i, 1) views


def engine_for(model, device):
    """The model's fused engine for `device`, (re)built when absent or stale.  Construction runs under a device guard: the C ABI
    launches on the CURRENT device's stream, so packing kernels for a model on cuda:1 must not be issued while cuda:0 is current."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("the fused engine runs on the MI355X only (got %s); there is no CPU path" % device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    eng = getattr(model, "_engine", None)
    if eng is None or eng.device != device or eng._key != eng._params_key(model):
        with torch.cuda.device(device):
            eng = PatchAugNetEngine(model, device)
        model._engine = eng
    return eng
