"""Gallery-sharded retrieval across the GPUs of one node.

The reference is single-process (src/utils.py:14-17); this is new design.  One process per GPU,
`torch.distributed` with backend "nccl" (= RCCL over xGMI).  Gallery images are independent units:
rank r encodes and keeps the contiguous slice [r*n_local, (r+1)*n_local) resident; a composed
query is fused on the rank that owns its reference image (its raw embeds, 1.45 MB, never move).
The only exchange steps are
  1. all_gather of the fused query vectors  [nq_local, 256] fp32   (nq*1 KiB in total),
  2. ONE all_gather of per-shard top-k      [nq, k] (fp32 score, int32 global index) = nq*k*8 B / rank, carrying in
     the same payload the scores of the <= 7 listed items per query (CIRR target / subset members) their owner holds,
followed by a k*R-candidate merge on every rank.  Both payloads are tiny (latency-bound); no
all-reduce, no ring over the feature tensors.  Because every comparison uses the integer key
(fl32(1 - sim), global index), the merged result is bit-identical to the single-GPU ranking.

The compute callables are injected so the orchestration can be exercised with world_size-2 gloo
tests on CPU (tests/test_dist_cpu.py passes oracle-backed functions); the product default is the
HIP engine and raises without it.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

SimFn = Callable[[torch.Tensor, torch.Tensor], torch.Tensor]                       # (fusion[nq,E], feats[n,32,E]) -> sim[nq,n]
TopkFn = Callable[[torch.Tensor, int, Optional[torch.Tensor], int], Tuple[torch.Tensor, torch.Tensor]]


def _hip_sim(fusion, feats):
    from . import engine as E
    return E.sim_max(fusion.contiguous(), feats.contiguous())


def _hip_topk(sim, k, gidx, idx_base):
    from . import engine as E
    return E.topk(sim, k, gidx=gidx, idx_base=idx_base)


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced gallery slices; the first (n_total % world) ranks hold one extra image."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def owner_of(index: torch.Tensor, n_total: int, world: int) -> torch.Tensor:
    """Rank that owns each global gallery index under `shard_bounds`."""
    base, rem = divmod(n_total, world)
    big = rem * (base + 1)
    idx = index.to(torch.int64)
    small_owner = rem + (idx - big) // max(base, 1)
    return torch.where(idx < big, idx // (base + 1), small_owner).to(torch.int64)


def offsets_of(counts) -> torch.Tensor:
    """[world+1] exclusive prefix sums of per-rank shard sizes (general form of `shard_bounds`: a rank's slice may have
    lost items its dataset failed to load, data_utils.py:191-192)."""
    c = torch.as_tensor(list(counts), dtype=torch.int64)
    return torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(c, 0)])


def owner_from_offsets(index: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    """Rank whose slice [offsets[r], offsets[r+1]) holds each global gallery index."""
    return torch.searchsorted(offsets[1:].contiguous(), index.to(torch.int64).contiguous(), right=True)


def _all_gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    """all_gather of equally-shaped tensors along dim 0 (one collective).  Over RCCL the payload stays on the device; with
    the gloo backend (CPU tests, and the oversubscribed N-ranks-on-one-GPU test mode) device tensors are staged through
    the host."""
    world = dist.get_world_size(group)
    if x.is_cuda and dist.get_backend(group) != "gloo":
        out = torch.empty((world * x.shape[0], *x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out
    h = x.detach().contiguous().cpu()
    out = torch.empty((world * h.shape[0], *h.shape[1:]), dtype=h.dtype)
    dist.all_gather(list(out.chunk(world, dim=0)), h, group=group)
    return out.to(x.device)


class ShardedRanker:
    """Ranks queries against a gallery sharded over the ranks of `group`."""

    def __init__(self, local_feats: torch.Tensor, index_base: int, sim_fn: SimFn = _hip_sim, topk_fn: TopkFn = _hip_topk,
                 group=None, always_exchange: Optional[bool] = None, sim_budget_bytes: int = 2 << 30):
        """always_exchange (default: environment SPRC_DIST_ALWAYS_EXCHANGE=1): run both all_gathers and the merge even in a
        one-rank group -- same result, and the RCCL path of a 1-GPU box is then the path an 8-GPU node takes.
        sim_budget_bytes: the local score matrix sim[nq, n_local] fp32 is never held whole when it is larger than this: the queries
        are ranked in blocks of budget / (4 n_local) rows -- the reference materialises nq x N (validate_blip.py:253-254; config C5's
        10 000 x 125 000 per GPU would be 5 GB), here top-k and the listed scores are taken block by block: same bits, bounded memory."""
        self.feats, self.base, self.sim_fn, self.topk_fn, self.group = local_feats, int(index_base), sim_fn, topk_fn, group
        self.sim_budget_bytes = int(sim_budget_bytes)
        if always_exchange is None:
            import os
            always_exchange = os.environ.get("SPRC_DIST_ALWAYS_EXCHANGE", "0") == "1"
        self.always_exchange = bool(always_exchange)

    def rank(self, fusion_local: torch.Tensor, k: int, listed: Optional[torch.Tensor] = None,
             select: Optional[torch.Tensor] = None):
        """fusion_local [nq_local,E] (same nq_local on every rank; pad with zero rows if ragged) ->
        (sim[nq,k], global idx[nq,k]) for the nq queries, identical on every rank.

        select: optional int64 [nq] positions in the gathered [world*nq_local] fusion rows (drops padding rows and
                puts the queries in the caller's order; default = all gathered rows in rank order).
        listed: optional [nq,L] GLOBAL gallery indices (-1 = none) whose scores the caller needs besides the top-k
                (CIRR subset members / targets): the owner of each contributes its score in the SAME all_gather as
                the per-shard top-k; returns a third tensor listed_sim[nq,L] (-inf where -1)."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        exchange = world > 1 or (self.always_exchange and dist.is_initialized())
        fusion = _all_gather_rows(fusion_local, self.group) if exchange else fusion_local       # exchange 1
        if select is not None:
            fusion = fusion.index_select(0, select.to(fusion.device))
        nq, n_local = fusion.shape[0], self.feats.shape[0]
        fusion = fusion.contiguous()
        qb = max(1, min(max(nq, 1), self.sim_budget_bytes // (4 * max(n_local, 1))))           # query rows per block of local scores
        vals_b, idx_b, lv_b = [], [], []
        for s in range(0, max(nq, 1), qb):
            sim = self.sim_fn(fusion[s:s + qb], self.feats)
            v, i = self.topk_fn(sim, k, None, self.base)
            vals_b.append(v)
            idx_b.append(i)
            if listed is not None:                                                              # scores of the listed items this rank owns
                col = listed[s:s + qb].to(device=sim.device, dtype=torch.int64) - self.base
                own = (col >= 0) & (col < n_local)
                lv_b.append(torch.where(own, sim.gather(1, col.clamp(0, max(n_local - 1, 0))) if n_local else torch.zeros_like(col, dtype=sim.dtype),
                                        torch.full(col.shape, float("-inf"), dtype=sim.dtype, device=sim.device)))
            del sim
        vals, idx = (vals_b[0], idx_b[0]) if len(vals_b) == 1 else (torch.cat(vals_b), torch.cat(idx_b))
        lv = None if listed is None else (lv_b[0] if len(lv_b) == 1 else torch.cat(lv_b))
        if not exchange:
            return (vals, idx) if listed is None else (vals, idx, lv)
        # exchange 2: ONE all_gather of [nq, k (score bits) + k (global index) + L (listed score bits)] int32 per rank
        parts = [vals.contiguous().view(torch.int32), idx] + ([lv.contiguous().view(torch.int32)] if lv is not None else [])
        got = _all_gather_rows(torch.cat(parts, dim=1).contiguous(), self.group).view(world, nq, -1)
        cand_v = got[:, :, :k].permute(1, 0, 2).reshape(nq, world * k).contiguous().view(torch.float32)
        cand_i = got[:, :, k:2 * k].permute(1, 0, 2).reshape(nq, world * k).contiguous()
        # unused slots (shard smaller than k) carry sim=-inf / idx=-1: give them the worst possible key
        cand_i = torch.where(cand_i < 0, torch.full_like(cand_i, 2**31 - 1), cand_i)
        mv, mi = self.topk_fn(cand_v, k, cand_i, 0)
        mi = torch.where(torch.isinf(mv) & (mv < 0), torch.full_like(mi, -1), mi)
        if listed is None:
            return mv, mi
        ls = got[:, :, 2 * k:].contiguous().view(torch.float32).max(dim=0).values               # exactly one owner per item
        return mv, mi, ls


def rank_logical_shards(feats: torch.Tensor, fusion: torch.Tensor, k: int, shards: int, sim_fn: SimFn = _hip_sim, topk_fn: TopkFn = _hip_topk,
                        sim_budget_bytes: int = 2 << 30):
    """ONE process, `shards` logical gallery shards (contiguous slices under `shard_bounds`): every shard is ranked by its own ShardedRanker
    (blocks of query rows, top-k per shard) and the shards' candidates are merged exactly as exchange 2 of `ShardedRanker.rank` merges the
    all_gather'ed payloads (k x shards candidates per query, integer keys) -- the 8-GPU data flow of BASELINE config C5 without the
    collectives.  -> (sim[nq,k], global idx[nq,k]), bit-identical to a global pass over the whole gallery."""
    n = feats.shape[0]
    cand_v, cand_i = [], []
    for r in range(shards):
        lo, hi = shard_bounds(n, shards, r)
        v, i = ShardedRanker(feats[lo:hi], index_base=lo, sim_fn=sim_fn, topk_fn=topk_fn, always_exchange=False,
                             sim_budget_bytes=sim_budget_bytes).rank(fusion, k)
        cand_v.append(v)
        cand_i.append(i)
    cv, ci = torch.cat(cand_v, 1).contiguous(), torch.cat(cand_i, 1).contiguous()
    ci = torch.where(ci < 0, torch.full_like(ci, 2**31 - 1), ci)
    mv, mi = topk_fn(cv, k, ci, 0)
    return mv, torch.where(torch.isinf(mv) & (mv < 0), torch.full_like(mi, -1), mi)
