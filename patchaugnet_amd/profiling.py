"""Per-stage device timing of one descriptor-extraction step (HIP events on the current torch stream)."""
import torch


class StageTimer:
    def __init__(self):
        self.marks = []

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append((name, e))

    def result(self):
        torch.cuda.synchronize()
        out = {}
        for (_, e0), (name, e1) in zip(self.marks[:-1], self.marks[1:]):
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        return out


def engine_stage_times(model, x, iters=5):
    """Average ms per stage of the fused engine (marks recorded by engine.PatchAugNetEngine)."""
    model(x, return_feat=False)                  # builds the engine
    eng = model._engine
    acc = {}
    for _ in range(iters):
        eng.timer = StageTimer()
        model(x, return_feat=False)
        t, eng.timer = eng.timer, None
        for k, ms in t.result().items():
            acc[k] = acc.get(k, 0.0) + ms / iters
    acc["total"] = sum(acc.values())
    return acc


def fp0_launch_time_ms(model, x, repeat=9, iters=3):
    """Average duration of ONE launch of the finest feature-propagation chain (the dominant kernel), HIP events on the launch stream:
    the stage is timed with the launch issued once and `repeat` times back to back; the difference / (repeat - 1) is the launch's own
    duration, free of the two event records and the dispatch gap that a single bracketed launch includes (~15 us on a 0.3 ms kernel).
    None when the level does not run the pre-multiplied chain."""
    model(x, return_feat=False)
    eng = model._engine
    chain = eng.fp[0]
    if getattr(chain, "_premul", None) is None:
        return None

    def stage(rep):
        chain.bench_repeat = rep
        try:
            tot = 0.0
            for _ in range(iters):
                eng.timer = StageTimer()
                model(x, return_feat=False)
                t, eng.timer = eng.timer, None
                tot += t.result().get("fp0.chain", 0.0) / iters
            return tot
        finally:
            chain.bench_repeat = 1
    stage(1)
    one, many = stage(1), stage(repeat)
    return (many - one) / (repeat - 1)


def stage_times(model, x, iters=5):
    """Average ms per stage over `iters` steps of the module path (stages: fps, knn, group+mlp, 3nn+interp+mlp, vlad, afa)."""
    if getattr(model, "fused_eval", False):
        return engine_stage_times(model, x, iters)
    from . import pointops
    bb, agg = model.backbone, model.aggregation
    acc = {}
    for _ in range(iters):
        t = StageTimer()
        t.mark("start")
        xyz = x.squeeze(1)
        l_xyz, l_feat = [xyz], [xyz.transpose(1, 2).contiguous()]
        for i, sa in enumerate(bb.SA_modules):
            ci = pointops.furthestsampling(l_xyz[i], sa.npoint)
            t.mark(f"sa{i}.fps")
            nx = pointops.gathering(l_xyz[i].transpose(1, 2).contiguous(), ci).transpose(1, 2).contiguous()
            cf = pointops.gathering(l_feat[i], ci)
            t.mark(f"sa{i}.gather")
            idx = pointops.knnquery(sa.groupers[0].nsample, l_xyz[i], nx)
            t.mark(f"sa{i}.knn")
            g, _ = sa.groupers[0](l_xyz[i], nx, l_feat[i], cf, idx=idx)
            t.mark(f"sa{i}.group")
            y = sa.mlps[0](g).max(dim=3)[0]
            t.mark(f"sa{i}.mlp+pool")
            l_xyz.append(nx)
            l_feat.append(y)
        n = len(bb.FP_modules)
        for i in range(-1, -(n + 1), -1):
            fp = bb.FP_modules[i]
            dist, idx = pointops.nearestneighbor(l_xyz[i - 1], l_xyz[i])
            t.mark(f"fp{n + i}.3nn")
            r = 1.0 / (dist + 1e-8)
            w = r / r.sum(dim=2, keepdim=True)
            f = pointops.interpolation(l_feat[i], idx, w)
            t.mark(f"fp{n + i}.interp")
            l_feat[i - 1] = fp.mlp(torch.cat([f, l_feat[i - 1]], dim=1).unsqueeze(-1)).squeeze(-1)
            t.mark(f"fp{n + i}.mlp")
        feats = [l_feat[j].unsqueeze(-1) for j in range(n - 1, -1, -1)]
        v = [vl(f) for vl, f in zip(agg.vlads, feats)]
        t.mark("vlad")
        agg.afa(torch.cat(v, dim=-1))
        t.mark("afa")
        for k, ms in t.result().items():
            acc[k] = acc.get(k, 0.0) + ms / iters
    acc["total"] = sum(acc.values())
    return acc
