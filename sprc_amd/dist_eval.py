"""Gallery-sharded evaluation: the CIRR / FashionIQ protocols of src/validate_blip.py:232-285, :24-57 and
src/cirr_test_submission.py:61-132 with the gallery split over the ranks of one node.

The reference is single-process (src/utils.py:14-17); what must be preserved is its OUTPUT: the seven CIRR numbers, the
two FashionIQ numbers per category, the two submission dicts.  Per rank (one process per GPU, `torchrun`):

  1. encode the rank's contiguous gallery slice (`dist.shard_bounds`; no communication) and keep the raw ViT embeddings
     of the local images that some query uses as its reference image;
  2. fuse the queries whose reference image this rank owns (`dist.owner_of`): a query's 1.45 MB of raw embeddings never
     leave the GPU that produced them; ragged per-rank query counts are padded to the maximum (known without
     communication: every rank holds the whole query list, which is metadata);
  3. `ShardedRanker.rank`: all_gather of the fused vectors, local similarity + top-k, then ONE all_gather that carries
     the per-shard top-k and -- from their owners -- the scores of each query's target / subset members, merged with the
     integer-key top-k kernel.  Every rank ends up with identical (top-k scores, top-k global indices, listed scores);
  4. metrics from those: Recall@K for K <= 50 needs the target's position in the merged top-51 (one slot for the
     reference image the protocol removes), subset recall needs the keys (fl32(1 - sim), index) of <= 6 members.

The numbers equal the single-process harness (sprc_amd/harness.py) on the same scores bit for bit: every comparison is
on the integer key.  Compute callables are injectable (world_size-2 gloo tests on CPU use oracle-backed doubles; the
`-m gpu` test runs two ranks on ONE GPU with the HIP kernels).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import dist as D
from .processors import fiq_compose_caption

K_TOP = 51          # top-50 + the reference image the CIRR protocol deletes (validate_blip.py:258-261)


# ---- query metadata --------------------------------------------------------------------------------------------------
@dataclass
class QuerySet:
    """The relative dataset as plain metadata (its val/test items carry no images: data_utils.py:236-262,171-186)."""
    ref_names: List[str]
    captions: List[str]
    target_names: Optional[List[str]] = None
    group_members: Optional[List[List[str]]] = None
    pair_ids: Optional[list] = None

    def __len__(self):
        return len(self.ref_names)


def cirr_val_queries(relative_val_dataset, txt_processors) -> QuerySet:
    """items: (ref_name, target_name, caption, members)   (validate_blip.py:380-410)"""
    q = QuerySet([], [], [], [])
    for item in _items(relative_val_dataset):
        ref, tgt, cap, members = item
        q.ref_names.append(ref); q.target_names.append(tgt); q.group_members.append(list(members))
        q.captions.append(txt_processors["eval"](cap))
    return q


def cirr_test_queries(relative_test_dataset, txt_processors) -> QuerySet:
    """items: (pair_id, ref_name, caption, members)   (cirr_test_submission.py:160-190)"""
    q = QuerySet([], [], None, [], [])
    for item in _items(relative_test_dataset):
        pid, ref, cap, members = item
        q.pair_ids.append(pid); q.ref_names.append(ref); q.group_members.append(list(members))
        q.captions.append(txt_processors["eval"](cap))
    return q


def fiq_val_queries(relative_val_dataset, txt_processors) -> QuerySet:
    """items: (ref_name, target_name, [cap1, cap2])   (validate_blip.py:170-207; composition :180-184)"""
    q = QuerySet([], [], [])
    for item in _items(relative_val_dataset):
        ref, tgt, caps = item
        q.ref_names.append(ref); q.target_names.append(tgt)
        q.captions.append(txt_processors["eval"](fiq_compose_caption(caps[0], caps[1])))
    return q


def _items(dataset):
    for i in range(len(dataset)):
        item = dataset[i]
        if item is not None:                 # datasets swallow per-item errors (data_utils.py:191-192,277-278)
            yield item


def gallery_names(dataset) -> List[str]:
    for attr in ("names", "image_names"):
        if hasattr(dataset, attr):
            return list(getattr(dataset, attr))
    raise AttributeError("classic dataset exposes neither .names nor .image_names")


# ---- the sharded gallery ---------------------------------------------------------------------------------------------
@dataclass
class GalleryShard:
    """This rank's slice of the encoded gallery + what every rank knows about the whole."""
    feats: torch.Tensor                      # [n_local, 32, E] fp32 unit rows (device)
    raw: Dict[str, torch.Tensor]             # name -> [tokens, D] raw ViT embeddings of the LOCAL reference images
    names: List[str]                         # ALL gallery names in global index order
    offsets: torch.Tensor                    # [world+1] slice boundaries (global index space)
    rank: int = 0
    world: int = 1
    name_to_index: Dict[str, int] = field(default_factory=dict)

    def __post_init__(self):
        if not self.name_to_index:
            self.name_to_index = {n: i for i, n in enumerate(self.names)}

    @property
    def base(self) -> int:
        return int(self.offsets[self.rank])

    def owner(self, index: torch.Tensor) -> torch.Tensor:
        n = len(self.names)
        lo_hi = [D.shard_bounds(n, self.world, r) for r in range(self.world)]
        balanced = all(int(self.offsets[r]) == lo_hi[r][0] for r in range(self.world))
        return D.owner_of(index, n, self.world) if balanced else D.owner_from_offsets(index, self.offsets)


def _world_rank(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def encode_gallery_shard(classic_dataset, blip_model, reference_names: Optional[Sequence[str]] = None, group=None,
                         batch_size: int = 64, num_workers: int = 2, extract_fn: Optional[Callable] = None) -> GalleryShard:
    """Encode gallery rows [lo, hi) of `classic_dataset` on this rank (src/utils.py:46-77 over a slice).
    reference_names: keep raw embeddings only for these images (None = for every local image)."""
    from torch.utils.data import Subset
    if extract_fn is None:
        from .harness import extract_index_blip_features as extract_fn
    world, rank = _world_rank(group)
    n_total = len(classic_dataset)
    lo, hi = D.shard_bounds(n_total, world, rank)
    keep = None if reference_names is None else set(reference_names)
    (feats, raw_store), names_local = extract_fn(Subset(classic_dataset, range(lo, hi)), blip_model, batch_size=batch_size,
                                                 num_workers=num_workers, keep_raw=keep if keep is not None else True)
    import os
    if world > 1 or (dist.is_initialized() and os.environ.get("SPRC_DIST_ALWAYS_EXCHANGE", "0") == "1"):
        # a slice may have lost unreadable images: exchange the actual name lists
        gathered: List[Optional[List[str]]] = [None] * world
        dist.all_gather_object(gathered, list(names_local), group=group)
    else:
        gathered = [list(names_local)]
    names = [n for part in gathered for n in part]
    offsets = D.offsets_of(len(part) for part in gathered)
    raw = {n: r for n, r in zip(names_local, raw_store) if r is not None}
    return GalleryShard(feats=feats, raw=raw, names=names, offsets=offsets, rank=rank, world=world)


# ---- ranking ---------------------------------------------------------------------------------------------------------
def sharded_rank(shard: GalleryShard, queries: QuerySet, blip_model=None, listed: Optional[np.ndarray] = None,
                 k: int = K_TOP, group=None, batch_size: int = 32, fuse_fn: Optional[Callable] = None,
                 sim_fn=None, topk_fn=None):
    """-> (top_sim[nq,k], top_idx[nq,k] global, listed_sim[nq,L] or None), identical on every rank.

    fuse_fn(ref_embeds[B,T,D], captions: list[str]) -> fusion[B,E]; default = blip_model.tokenizer + blip_model.fuse."""
    world, rank = shard.world, shard.rank
    nq = len(queries)
    ref_idx = torch.tensor([shard.name_to_index[n] for n in queries.ref_names], dtype=torch.int64)
    owner = shard.owner(ref_idx) if nq else torch.zeros(0, dtype=torch.int64)
    counts = torch.bincount(owner, minlength=world)
    nq_pad = max(int(counts.max()) if nq else 0, 1)
    # slot of every query inside its owner's (padded) block of the gathered fusion rows: rank-major, query order within
    slot = torch.zeros(nq, dtype=torch.int64)
    for r in range(world):
        sel = owner == r
        slot[sel] = torch.arange(int(sel.sum()), dtype=torch.int64)
    select = owner * nq_pad + slot
    mine = torch.nonzero(owner == rank).flatten().tolist()
    if fuse_fn is None:
        def fuse_fn(ref, caps):
            tok = blip_model.tokenizer(caps, padding="max_length", truncation=True, max_length=blip_model.max_txt_len,
                                       return_tensors="pt").to(blip_model.device)
            return blip_model.fuse(ref, tok.input_ids, tok.attention_mask)
    dev = shard.feats.device
    E = shard.feats.shape[-1]
    fusion_local = torch.zeros((nq_pad, E), dtype=torch.float32, device=dev)
    for s in range(0, len(mine), batch_size):
        ids = mine[s:s + batch_size]
        ref = torch.stack([shard.raw[queries.ref_names[i]] for i in ids]).to(dev)
        fusion_local[s:s + len(ids)] = fuse_fn(ref, [queries.captions[i] for i in ids]).to(torch.float32)
    kw = {}
    if sim_fn is not None:
        kw["sim_fn"] = sim_fn
    if topk_fn is not None:
        kw["topk_fn"] = topk_fn
    ranker = D.ShardedRanker(shard.feats, shard.base, group=group, **kw)
    lt = None if listed is None else torch.from_numpy(np.ascontiguousarray(listed, dtype=np.int64))
    out = ranker.rank(fusion_local, k, listed=lt, select=select if world > 1 else select)
    return out if listed is not None else (*out, None)


# ---- metrics from (top-k, listed scores): the integer-key order of the single-process harness ----------------------------
def _key_less(sim_a, idx_a, sim_b, idx_b):
    """(fl32(1 - sim_a), idx_a) < (fl32(1 - sim_b), idx_b): the ranking contract."""
    da, db = np.float32(1.0) - sim_a.astype(np.float32), np.float32(1.0) - sim_b.astype(np.float32)
    return (da < db) | ((da == db) & (idx_a < idx_b))


def _pct(hits: np.ndarray) -> float:
    return float(np.float32(hits.sum()) / np.float32(len(hits))) * 100


def _position(top_idx: np.ndarray, wanted: np.ndarray) -> np.ndarray:
    """position of wanted[q] in top_idx[q] (k where absent)."""
    hit = (top_idx == wanted[:, None]) & (top_idx >= 0)          # -1 = an unused slot of a gallery with < k rows, never a hit
    return np.where(hit.any(1), hit.argmax(1), top_idx.shape[1])


def cirr_val_metrics_from_topk(top_idx, listed_sim, ref_idx, tgt_idx, group_idx):
    """listed = [target | members]: -> (group_recall@1,2,3, recall@1,5,10,50) as validate_blip.py:253-285."""
    top_idx = np.asarray(top_idx, dtype=np.int64)
    ls = np.asarray(listed_sim, dtype=np.float32)
    ref_idx, tgt_idx, group_idx = (np.asarray(a, dtype=np.int64) for a in (ref_idx, tgt_idx, group_idx))
    assert (tgt_idx != ref_idx).all() and np.isfinite(ls[:, 0]).all(), \
        "every query needs its target in the gallery, distinct from the reference"
    pos_t = _position(top_idx, tgt_idx)
    pos_r = _position(top_idx, ref_idx)
    rank_t = pos_t - (pos_r < pos_t)                                 # drop the reference (validate_blip.py:258-261)
    assert ((group_idx == tgt_idx[:, None]).sum(1) == 1).all(), "target must appear exactly once among the group members"
    member_ok = (group_idx != ref_idx[:, None]) & (group_idx >= 0)
    ahead = _key_less(ls[:, 1:], group_idx, ls[:, :1], tgt_idx[:, None]) & member_ok
    pos_in_group = ahead.sum(1)                                      # :268-271
    return (_pct(pos_in_group < 1), _pct(pos_in_group < 2), _pct(pos_in_group < 3),
            _pct(rank_t < 1), _pct(rank_t < 5), _pct(rank_t < 10), _pct(rank_t < 50))


def fiq_metrics_from_topk(top_idx, tgt_idx) -> Tuple[float, float]:
    tgt_idx = np.asarray(tgt_idx, dtype=np.int64)
    assert (tgt_idx >= 0).all(), "every query needs its target in the gallery (validate_blip.py:51 asserts one label per query)"
    pos = _position(np.asarray(top_idx, dtype=np.int64), tgt_idx)
    return _pct(pos < 10), _pct(pos < 50)                            # validate_blip.py:44-57 (reference kept)


def cirr_test_dicts_from_topk(top_idx, listed_sim, ref_idx, group_idx, pairs_id, index_names):
    """listed = members: -> (pairid -> top-50 names, pairid -> top-3 subset names), cirr_test_submission.py:114-130."""
    top_idx = np.asarray(top_idx, dtype=np.int64)
    ls = np.asarray(listed_sim, dtype=np.float32)
    names = np.asarray(index_names)
    n = len(index_names)
    top, sub = {}, {}
    for q, pid in enumerate(pairs_id):
        row = [i for i in top_idx[q] if i >= 0 and i != ref_idx[q]][: min(50, n - 1)]
        top[str(int(pid))] = names[row].tolist()
        mem = [(np.float32(1.0) - ls[q, j], int(group_idx[q][j])) for j in range(len(group_idx[q]))
               if group_idx[q][j] >= 0 and group_idx[q][j] != ref_idx[q]]
        mem.sort()
        sub[str(int(pid))] = names[[m[1] for m in mem[:3]]].tolist()
    return top, sub


# ---- the three protocols -----------------------------------------------------------------------------------------------
def _idx(shard: GalleryShard, names: Sequence[str]) -> np.ndarray:
    return np.asarray([shard.name_to_index.get(n, -1) for n in names], dtype=np.int64)


def compute_cirr_val_metrics_sharded(relative_val_dataset, classic_val_dataset, blip_model, txt_processors, group=None,
        gallery_batch_size: int = 64, num_workers: int = 2, **kw):
    """Sharded counterpart of harness.compute_cirr_val_metrics (validate_blip.py:232-285): same seven numbers on every rank."""
    q = cirr_val_queries(relative_val_dataset, txt_processors)
    shard = encode_gallery_shard(classic_val_dataset, blip_model, reference_names=q.ref_names, group=group,
                                 batch_size=gallery_batch_size, num_workers=num_workers)
    ref, tgt = _idx(shard, q.ref_names), _idx(shard, q.target_names)
    grp = np.stack([_idx(shard, g) for g in q.group_members]) if len(q) else np.zeros((0, 6), np.int64)
    listed = np.concatenate([tgt[:, None], grp], axis=1)
    _, top_idx, ls = sharded_rank(shard, q, blip_model, listed=listed, group=group, **kw)
    return cirr_val_metrics_from_topk(top_idx.cpu().numpy(), ls.cpu().numpy(), ref, tgt, grp)


def compute_fiq_val_metrics_sharded(relative_val_dataset, classic_val_dataset, blip_model, txt_processors, group=None,
        gallery_batch_size: int = 64, num_workers: int = 2, **kw):
    q = fiq_val_queries(relative_val_dataset, txt_processors)
    shard = encode_gallery_shard(classic_val_dataset, blip_model, reference_names=q.ref_names, group=group,
                                 batch_size=gallery_batch_size, num_workers=num_workers)
    _, top_idx, _ = sharded_rank(shard, q, blip_model, group=group, **kw)
    return fiq_metrics_from_topk(top_idx.cpu().numpy(), _idx(shard, q.target_names))


def generate_cirr_test_dicts_sharded(relative_test_dataset, classic_test_dataset, blip_model, txt_processors, group=None,
        gallery_batch_size: int = 64, num_workers: int = 2, **kw):
    q = cirr_test_queries(relative_test_dataset, txt_processors)
    shard = encode_gallery_shard(classic_test_dataset, blip_model, reference_names=q.ref_names, group=group,
                                 batch_size=gallery_batch_size, num_workers=num_workers)
    ref = _idx(shard, q.ref_names)
    grp = np.stack([_idx(shard, g) for g in q.group_members]) if len(q) else np.zeros((0, 6), np.int64)
    _, top_idx, ls = sharded_rank(shard, q, blip_model, listed=grp, group=group, **kw)
    return cirr_test_dicts_from_topk(top_idx.cpu().numpy(), ls.cpu().numpy(), ref, grp, q.pair_ids, shard.names)


def init_from_env(device_index: Optional[int] = None):
    """`torchrun` plumbing shared by the entry points: one process per GPU, backend nccl (= RCCL over xGMI).  When more
    ranks than visible GPUs are launched (test mode: N ranks on one GPU), ranks share devices and the tiny exchanges go
    through gloo -- RCCL refuses two ranks on one device."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("sprc_amd needs an MI355X: the HIP kernels are the only compute path")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", (local if device_index is None else device_index) % ndev)
    torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world <= ndev:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    return dev, world, int(os.environ.get("RANK", "0"))
