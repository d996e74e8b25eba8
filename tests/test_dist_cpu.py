"""world_size-2 gloo test of the gallery-sharded ranking orchestration (sprc_amd/dist.py).
The compute callables are oracle-backed test doubles here; on the GPU box the same orchestration runs
with the HIP kernels over RCCL (tests/test_e2e_gpu.py covers world_size 1; bench.py --gpus N covers N>1)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sprc_oracle as O


def _cpu_sim(fusion, feats):
    return O.similarity(fusion, feats)


def _cpu_topk(sim, k, gidx, idx_base):
    """Oracle-backed double of sprc_topk: smallest keys (fl32(1-sim), index) with explicit global indices."""
    s = sim.numpy().astype(np.float32)
    nq, N = s.shape
    gi = gidx.numpy().astype(np.int64) if gidx is not None else (np.arange(N, dtype=np.int64)[None, :] + idx_base).repeat(nq, 0)
    d = O.distances(s)
    order = np.lexsort((gi, d), axis=-1)[:, :k]                      # primary d, secondary global index
    vals = np.full((nq, k), -np.inf, dtype=np.float32)
    idx = np.full((nq, k), -1, dtype=np.int32)
    kk = min(k, N)
    vals[:, :kk] = np.take_along_axis(s, order, 1)[:, :kk]
    idx[:, :kk] = np.take_along_axis(gi, order, 1)[:, :kk]
    return torch.from_numpy(vals), torch.from_numpy(idx)


def _worker(rank, world, port, n_total, nq_local, k, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sprc_amd.dist import ShardedRanker, shard_bounds
    g = torch.Generator().manual_seed(7)
    feats = torch.nn.functional.normalize(torch.randn((n_total, 32, 16), generator=g), dim=-1)
    feats = (feats * 8).round() / 8                                  # coarse grid -> exact ties across shards
    fusion = torch.nn.functional.normalize(torch.randn((world * nq_local, 16), generator=g), dim=-1)
    fusion = (fusion * 8).round() / 8
    lo, hi = shard_bounds(n_total, world, rank)
    rk = ShardedRanker(feats[lo:hi].contiguous(), lo, sim_fn=_cpu_sim, topk_fn=_cpu_topk)
    vals, idx = rk.rank(fusion[rank * nq_local:(rank + 1) * nq_local].contiguous(), k)
    want_v, want_i = O.topk_stable(O.similarity(fusion, feats).numpy(), min(k, n_total))
    ok = np.array_equal(idx.numpy()[:, :want_i.shape[1]], want_i.astype(np.int32)) and \
        np.array_equal(vals.numpy()[:, :want_i.shape[1]], want_v)
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_ranking_equals_single_process_world2():
    for n_total, nq_local, k in [(37, 3, 10), (5, 2, 8)]:            # second case: shards smaller than k
        with mp.Manager() as mgr:
            out = mgr.dict()
            mp.spawn(_worker, args=(2, _free_port(), n_total, nq_local, k, out), nprocs=2, join=True)
            assert out[0] and out[1], (n_total, nq_local, k, dict(out))


def test_sharded_ranking_equals_single_process_world3_uneven():
    """three ranks, gallery sizes that do not divide (shards of 34/33/33 and 3/2/2 images, the latter smaller than k)."""
    for n_total, nq_local, k in [(100, 2, 12), (7, 1, 5)]:
        with mp.Manager() as mgr:
            out = mgr.dict()
            mp.spawn(_worker, args=(3, _free_port(), n_total, nq_local, k, out), nprocs=3, join=True)
            assert out[0] and out[1] and out[2], (n_total, nq_local, k, dict(out))


def test_local_scores_are_ranked_in_query_blocks_under_a_memory_budget():
    """ShardedRanker.sim_budget_bytes: the local sim[nq, n_local] is produced and consumed in blocks of query rows when it would be
    larger than the budget (config C5: 10 000 x 125 000 fp32 = 5 GB per GPU); top-k and the listed scores are the same bits as from
    the whole matrix (one process, no group: the local half of `rank`)."""
    from sprc_amd.dist import ShardedRanker
    g = torch.Generator().manual_seed(11)
    feats = ((torch.nn.functional.normalize(torch.randn((41, 32, 16), generator=g), dim=-1) * 8).round() / 8)
    fusion = ((torch.nn.functional.normalize(torch.randn((23, 16), generator=g), dim=-1) * 8).round() / 8)
    listed = torch.randint(-1, 41, (23, 6), generator=g)
    calls = []

    def sim_fn(f, t):
        calls.append(f.shape[0])
        return _cpu_sim(f, t)

    whole = ShardedRanker(feats, 100, sim_fn=sim_fn, topk_fn=_cpu_topk).rank(fusion, 9, listed=listed + 100 * (listed >= 0))
    assert calls == [23]
    calls.clear()
    blocks = ShardedRanker(feats, 100, sim_fn=sim_fn, topk_fn=_cpu_topk, sim_budget_bytes=5 * 41 * 4).rank(
        fusion, 9, listed=listed + 100 * (listed >= 0))
    assert calls == [5, 5, 5, 5, 3]
    for a, b in zip(whole, blocks):
        assert torch.equal(a, b)
    want_v, want_i = O.topk_stable(O.similarity(fusion, feats).numpy(), 9)
    assert np.array_equal(blocks[1].numpy(), want_i.astype(np.int32) + 100) and np.array_equal(blocks[0].numpy(), want_v)
