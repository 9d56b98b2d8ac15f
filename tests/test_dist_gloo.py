"""The N > 1 path on CPU: two processes, gloo backend.  Sharding, the descriptor all-gather and the sharded recall
exchange are exercised with a deterministic CPU stand-in for the extractor and a numpy kNN (the HIP kernels need the
MI355X; their parity is covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StandInModel:
    """(B,1,N,3) -> (B,256): a fixed random projection of per-cloud moments; deterministic, batch-independent."""

    def __init__(self):
        g = torch.Generator().manual_seed(3)
        self.w = torch.randn(9, 256, generator=g)

    def __call__(self, x, return_feat=False):
        p = x.squeeze(1)
        f = torch.cat([p.mean(1), p.std(1), p.abs().amax(1)], dim=1)
        return torch.nn.functional.normalize(torch.stack([(r[:, None] * self.w).sum(0) for r in f]))   # row by row: bit-identical for any batch split


def _clouds(lo, hi):
    out = []
    for i in range(lo, hi):
        g = torch.Generator().manual_seed(1000 + i)
        out.append(torch.rand(1, 64, 3, generator=g) * (1 + 0.01 * i))
    return torch.stack(out)


def _numpy_knn(database, queries, k):
    out = []
    for lo in range(0, queries.shape[0], 64):                    # 64 queries at a time: the (nq, nd, dim) float64 cube stays small
        d = ((queries[lo:lo + 64].double()[:, None, :] - database.double()[None, :, :]) ** 2).sum(-1).numpy()
        out.append(np.lexsort((np.broadcast_to(np.arange(d.shape[1]), d.shape), d), axis=1)[:, :k])
    return torch.from_numpy(np.concatenate(out).copy() if out else np.zeros((0, k), dtype=np.int64))


def _worker(rank, world, port, n_total, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.gen_recall_golden import synthetic_route
        from patchaugnet_amd import distributed, retrieval
        descs = distributed.extract_dataset(_StandInModel(), _clouds, n_total, batch_size=4, n_streams=0, device=torch.device("cpu"))
        sizes = [40, 33, 28]
        _, desc, tuples = synthetic_route(5, sizes)
        res = retrieval.get_recall_precision(torch.from_numpy(desc), sizes, tuples, top_k=10, knn=_numpy_knn)
        if rank == 0:
            ret["descs"] = descs.numpy()
            ret["recall"] = {k: (v[0], v[2], v[6]) for k, v in res.items()}
        ret[f"bounds{rank}"] = distributed.shard_bounds(n_total, rank, world)
        # bench.py's cross-rank bookkeeping: rank 1 is the slow one in repetition 0, rank 0 in repetition 2
        rep_dt = [1.0 + rank, 2.0, 3.0 - rank]
        ret[f"stats{rank}"] = distributed.run_stats(rep_dt, [t * 0.5 for t in rep_dt], 640, torch.device("cpu"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [21, 8])
def test_two_rank_extraction_and_recall_match_single_process(n_total):
    from oracle.gen_recall_golden import synthetic_route
    from patchaugnet_amd import distributed, retrieval
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        ret = dict(ret)
    single = distributed.extract_dataset(_StandInModel(), _clouds, n_total, batch_size=4, n_streams=0, device=torch.device("cpu"))
    assert ret["descs"].shape == (n_total, 256) and np.array_equal(ret["descs"], single.numpy())
    lo0, hi0 = ret["bounds0"]
    lo1, hi1 = ret["bounds1"]
    assert (lo0, hi1) == (0, n_total) and hi0 == lo1                      # contiguous, complete, disjoint
    for r in range(2):      # slowest rank per repetition, both rank ids seen through the all-gather, each rank's own median rate -- on every rank
        dt_max, seen, per_rank = ret[f"stats{r}"]
        assert dt_max == [2.0, 2.0, 3.0] and seen == 2 and per_rank == [640 / 1.0, 640 / 1.0]
    sizes = [40, 33, 28]
    _, desc, tuples = synthetic_route(5, sizes)
    ref = retrieval.get_recall_precision(torch.from_numpy(desc), sizes, tuples, top_k=10, knn=_numpy_knn)
    assert sorted(ret["recall"]) == sorted(ref)
    for k, v in ref.items():
        assert np.array_equal(ret["recall"][k][0], v[0]) and ret["recall"][k][1:] == (v[2], v[6])


def test_run_stats_without_a_process_group():
    from patchaugnet_amd.distributed import run_stats
    assert run_stats([3.0, 1.0, 2.0], [1.5, 0.5, 1.0], 100, torch.device("cpu")) == ([3.0, 1.0, 2.0], 1, [100.0])


def test_shard_bounds_cover_everything():
    from patchaugnet_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 9, 3000):
        for w in (1, 2, 4, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) <= -(-n // w) if n else True
