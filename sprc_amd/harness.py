"""Evaluation harness of the retrieval path: gallery feature extraction, query prediction loops and
Recall@K metrics, with the reference's function names, arguments and return values
(src/utils.py:46-77,141-148; src/validate_blip.py:24-57,149-207,232-285,359-410;
src/cirr_test_submission.py:61-190) so the reference's scripts can import them unchanged.

What differs is HOW the metrics are computed: the reference sorts every row of `1 - sim` with
torch.argsort and then runs O(nq*N) numpy string comparisons (validate_blip.py:253-271).  Here the
names are mapped to gallery indices once and the ranks come from the HIP ranking kernels
(sprc_rank_of / sprc_topk): integer-exact, under the stable tie rule (fl32(1-sim), index), with no
full sort and no nq x N host traffic.
"""
from __future__ import annotations

from operator import itemgetter
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import engine as E
from .processors import fiq_compose_caption

try:  # progress bars as in the reference, optional
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **k):
        return x


def collate_fn(batch: list):
    """Drop `None` items (datasets swallow per-item errors, data_utils.py:191-192,277-278; utils.py:141-148)."""
    batch = [b for b in batch if b is not None]
    return torch.utils.data.dataloader.default_collate(batch)


def _collate_ragged(batch: list):
    """(names, [uint8 HWC tensors of different sizes]) with `None` items dropped."""
    batch = [b for b in batch if b is not None]
    return [b[0] for b in batch], [b[1] for b in batch]


def _decode_only(dataset):
    """-> (the dataset's on-device transform, a shallow copy of the dataset that only decodes); Subset-aware."""
    import copy
    from torch.utils.data import Subset
    from .data_utils import DecodeRGB
    if isinstance(dataset, Subset):
        tf, inner = _decode_only(dataset.dataset)
        return tf, Subset(inner, dataset.indices)
    clone = copy.copy(dataset)
    tf = dataset.preprocess
    clone.preprocess = DecodeRGB(getattr(tf, "ratio", 1.25), getattr(tf, "dim", 224))
    return tf, clone


def _pin(dataset) -> bool:
    """pin_memory for the loader unless the dataset's transform already returns device tensors (data_utils.GpuTargetPad)."""
    while hasattr(dataset, "dataset"):              # torch.utils.data.Subset
        dataset = dataset.dataset
    return not getattr(getattr(dataset, "preprocess", None), "on_device", False)


class RawStore:
    """`index_features[1]` when raw ViT embeddings are kept for a SUBSET of the gallery (the
    protocol returns raw[N,257,D] fp32 for every image, 1.45 MB each -- 3.3 GB for CIRR val, 1.4 TB for a 1 M gallery --
    but only the images some query uses as its reference are ever looked up, validate_blip.py:377,395-399).
    Indexable by gallery position and iterable in gallery order like the stacked tensor (rows that were not kept
    yield None), so `dict(zip(index_names, index_features[1]))` keeps working."""

    def __init__(self, n: int, rows: Dict[int, torch.Tensor]):
        self.n, self.rows = int(n), rows

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        i = int(i)
        if i < 0:
            i += self.n
        return self.rows.get(i)

    def __iter__(self):
        return (self.rows.get(i) for i in range(self.n))

    def to(self, *a, **k):
        return RawStore(self.n, {i: r.to(*a, **k) for i, r in self.rows.items()})

    def cpu(self):
        return self.to("cpu")

    @property
    def kept(self) -> int:
        return len(self.rows)

    def nbytes(self) -> int:
        return sum(r.numel() * r.element_size() for r in self.rows.values())


class _ThreadLoader:
    """Decode-only gallery loader on a THREAD pool: yields (names, [uint8 HWC tensors]) batches in dataset order, `None` items dropped
    (collate_fn's contract, utils.py:141-148).  Pillow releases the GIL while it inflates / decodes, so threads decode in parallel --
    without the cost of starting worker processes (a fork-server worker imports torch: ~1 s each, 11 s for twelve on the MI355X
    box's host, for a gallery the engine encodes in 1.6 s) and without pickling every decoded image through a pipe."""

    def __init__(self, dataset, batch_size: int, threads: int, ahead: int = 3, first: int = 32):
        """first: size of the FIRST batch (the rest of that batch follows as the second one, every later batch keeps its boundaries):
        the GPU starts on 32 images after ~25 ms instead of waiting ~150 ms for 128 to be decoded and transformed -- the pipeline's
        fill was all of the difference between a gallery pass from files and the engine on resident tensors (tools/c2_e2e.py)."""
        self.ds, self.bs, self.threads, self.ahead = dataset, batch_size, max(1, threads), ahead
        n = len(dataset)
        cuts = list(range(0, n, batch_size)) + [n]
        if 0 < first < min(batch_size, n):
            cuts.insert(1, first)
        self.spans = list(zip(cuts[:-1], cuts[1:]))

    def __len__(self):
        return len(self.spans)

    def __iter__(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        n = len(self.ds)
        with ThreadPoolExecutor(max_workers=self.threads) as pool:
            pending = deque()
            spans = iter(self.spans)

            def submit():
                s = next(spans, None)
                if s is not None:
                    pending.append([pool.submit(self.ds.__getitem__, i) for i in range(*s)])

            for _ in range(self.ahead):
                submit()
            while pending:
                items = [f.result() for f in pending.popleft()]
                submit()
                yield _collate_ragged(items)


def usable_cores() -> int:
    """Host cores this process may use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def extract_index_blip_features(dataset, blip_model, save_memory: bool = False, batch_size: Optional[int] = None, num_workers: int = 2,
                                keep_raw=True, raw_dtype: Optional[torch.dtype] = None):
    """-> ((feats[N,32,256], raw), names[N])   (src/utils.py:46-77)

    keep_raw=True (the reference's behaviour): raw = the stacked [N,257,D] fp32 tensor.
    keep_raw=<collection of names> (or False): raw = a `RawStore` holding the embeddings of those images only
    (`raw_dtype=torch.bfloat16` halves them; they are cast back to fp32 when a query is fused)."""
    # pixel work on the GPU (data_utils.GpuTargetPad) and workers asked for: the workers only DECODE (a copy of the dataset whose
    # transform is DecodeRGB), the transform runs here on what they hand over -- PNG / JPEG decoding is the slow part of a gallery
    # pass (2297 files: 5.2 s on one core against 1.6 s of GPU work)
    gpu_tf = None
    if num_workers > 0 and not _pin(dataset):
        gpu_tf, dataset = _decode_only(dataset)
        # image decoding is then the only host work and the slowest stage of a real gallery pass (one core decodes ~450 PNG files / s,
        # the engine encodes ~1600 images / s): decode on a thread pool over the cores the host has (_ThreadLoader; the reference's
        # loaders fork 2 processes: utils.py:54), at the batch the engine is benchmarked at
        batch_size = batch_size or 128
    batch_size = batch_size or 64                                    # utils.py:54
    # decode workers come from a fork SERVER (a small process started once): forking them from this process -- GPU context,
    # gigabytes of mapped memory -- cost ~24 s per loader on the MI355X box (tools/c2_e2e.py); SPRC_LOADER_CONTEXT overrides
    ctx = (os.environ.get("SPRC_LOADER_CONTEXT") or "forkserver") if num_workers > 0 else None
    if gpu_tf is not None and os.environ.get("SPRC_LOADER_THREADS", "1") != "0":
        loader = _ThreadLoader(dataset, batch_size, threads=int(os.environ.get("SPRC_DECODE_THREADS") or max(2, min(12, usable_cores() - 2))))
    else:
        loader = DataLoader(dataset=dataset, batch_size=batch_size, num_workers=num_workers, pin_memory=gpu_tf is None and _pin(dataset),
                            collate_fn=_collate_ragged if gpu_tf is not None else collate_fn, multiprocessing_context=ctx,
                            persistent_workers=False)
    feats, raws, names = [], [], []
    split = getattr(dataset, "split", "")
    print(f"extracting {type(dataset).__name__} {split} index features")
    dev = blip_model.device
    subset = keep_raw is not True
    wanted = set() if keep_raw in (False, None) else (set(keep_raw) if subset else None)
    rows: Dict[int, torch.Tensor] = {}
    tf_stream = None
    trace = [] if os.environ.get("SPRC_TRACE_GALLERY") else None     # per batch: host seconds waiting for the loader / issuing the transform / the encode
    t_prev = time.perf_counter()
    for batch_names, images in tqdm(loader):
        t_got = time.perf_counter()
        if gpu_tf is not None:       # uint8 images go through the GPU transform; items a worker already transformed (modes the GPU
            from .data_utils import is_transformed      # path does not reproduce: palette, alpha, ...) are finished tensors
            # on a side stream: a host-to-device copy from pageable memory blocks the host until everything queued before it on
            # ITS stream is done -- on the compute stream that is the previous batch's 80 ms of encoding, and the next batch could
            # not even be prepared meanwhile
            if tf_stream is None:
                tf_stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(tf_stream):
                images = torch.stack([im.to(dev, non_blocking=True) if is_transformed(im) else gpu_tf(im) for im in images])
            torch.cuda.current_stream(dev).wait_stream(tf_stream)
            images.record_stream(torch.cuda.current_stream(dev))
        images = images.to(dev, non_blocking=True)
        t_tf = time.perf_counter()
        f, r = blip_model.extract_target_features(images, mode="mean")
        if trace is not None:
            trace.append((t_got - t_prev, t_tf - t_got, time.perf_counter() - t_tf))
            t_prev = time.perf_counter()
        if raw_dtype is not None:
            r = r.to(raw_dtype)
        if save_memory:
            f, r = f.cpu(), r.cpu()
        feats.append(f)
        if subset:
            for j, n in enumerate(batch_names):
                if n in wanted:
                    rows[len(names) + j] = r[j].clone()
        else:
            raws.append(r)
        names.extend(batch_names)
    if trace:
        w, t, e = (sum(x[i] for x in trace) for i in range(3))
        print(f"[gallery trace] {len(trace)} batches: host waited {w * 1e3:.0f} ms for the loader (first batch {trace[0][0] * 1e3:.0f}), spent "
              f"{t * 1e3:.0f} ms issuing transforms (first {trace[0][1] * 1e3:.0f}) and {e * 1e3:.0f} ms issuing encodes (first {trace[0][2] * 1e3:.0f})")
    if subset:
        return (torch.vstack(feats), RawStore(len(names), rows)), names
    return (torch.vstack(feats), torch.vstack(raws)), names


class ReferenceKVMap:
    """name -> row of precomputed cross-attention K|V projections of the reference images (`build_reference_kv`): stands in for the
    name -> raw-embedding dict of the prediction loops when `reuse_reference_kv` is on."""

    def __init__(self, kv: torch.Tensor, rows: Dict[str, int]):
        self.kv, self.rows = kv, rows

    def get(self, name):
        return self.rows.get(name)


def build_reference_kv(blip_model, index_names: List[str], index_features, reference_names: Sequence[str]) -> ReferenceKVMap:
    """K|V projections (Engine.encode_kv: Qformer.py:191-193 for all six cross-attention layers) of every DISTINCT reference image,
    once: CIRR-val has 4181 queries over ~2000 distinct reference images, and the fusion pass otherwise projects a reference
    image's 257 tokens once per query (6.7 of a query's 29.3 GFLOP).  4.7 MB per image in the 16-bit dtype."""
    name_to_feat = dict(zip(index_names, index_features[1]))
    uniq = list(dict.fromkeys(reference_names))
    eng = blip_model.engine()
    if not eng.is16:                             # (checked BEFORE the allocation: the fused-K|V entry point exists for the 16-bit engines only)
        raise ValueError("reuse_reference_kv needs a 16-bit engine (fp16 / bf16): sprc_qformer_fuse_kv is not built for the fp32 parity engine")
    need = len(uniq) * eng.cfg.vit.tokens * eng.kv_width * torch.empty((), dtype=eng.tdt).element_size()
    free = torch.cuda.mem_get_info(eng.device)[0]
    if need > 0.8 * free:                        # ~4.7 MB per distinct reference image (9.5 GB for CIRR-val): refuse rather than thrash
        raise MemoryError(f"reuse_reference_kv: the K|V cache of {len(uniq)} distinct reference images needs {need / 2**30:.1f} GiB, "
                          f"{free / 2**30:.1f} GiB are free -- evaluate without reuse_reference_kv, or in query chunks")
    kv = torch.empty((len(uniq), eng.cfg.vit.tokens, eng.kv_width), dtype=eng.tdt, device=eng.device)
    for s in range(0, len(uniq), 64):
        eng.encode_kv(_stack_refs(name_to_feat, uniq[s:s + 64]).to(eng.device), out=kv[s:s + 64])
    return ReferenceKVMap(kv, {n: i for i, n in enumerate(uniq)})


def _stack_refs(name_to_feat, names: Sequence[str]):
    if isinstance(name_to_feat, ReferenceKVMap):
        from .model import ReferenceKV
        rows = [name_to_feat.get(n) for n in names]
        if any(r is None for r in rows):
            raise KeyError("reference image without precomputed K|V projections")
        return ReferenceKV(name_to_feat.kv, torch.tensor(rows, dtype=torch.int32))
    missing = [n for n in names if name_to_feat.get(n) is None]
    if missing:
        raise KeyError(f"no raw embeddings kept for reference image(s) {missing[:3]}: pass their names in `keep_raw`")
    if len(names) == 1:
        return name_to_feat[names[0]].unsqueeze(0).float()
    return torch.stack(itemgetter(*names)(name_to_feat)).float()


# ---- CIRR validation ---------------------------------------------------------------------------------
def _query_loader(dataset, batch_size: int, num_workers: int, **kw) -> DataLoader:
    """Loader of a RELATIVE split: its items are names and captions, no pixels.  The reference forks 2-4 workers for it
    (validate_blip.py:164,373; cirr_test_submission.py:149); forked from a process that holds a GPU context and a few GB of
    pinned / mapped memory that start-up alone cost 26-36 s of a 50-s CIRR-val evaluation (tools/c2_e2e.py) for work that
    takes 0.4 s in the main process.  `num_workers` is accepted for signature compatibility and ignored."""
    del num_workers
    return DataLoader(dataset=dataset, batch_size=batch_size, num_workers=0, pin_memory=False, **kw)


def generate_cirr_val_predictions(blip_model, relative_val_dataset, index_names: List[str], index_features, txt_processors,
                                  batch_size: int = 32, num_workers: int = 2, reference_kv: Optional[ReferenceKVMap] = None):
    """-> (sim[nq,N], reference_names, target_names, group_members, captions)   (validate_blip.py:359-410)
    reference_kv (optional, `build_reference_kv`): the fusion pass reads precomputed K|V projections of the reference images."""
    print("Compute CIRR validation predictions")
    loader = _query_loader(relative_val_dataset, batch_size, num_workers, collate_fn=collate_fn)
    name_to_feat = reference_kv if reference_kv is not None else dict(zip(index_names, index_features[1]))
    sims, target_names, group_members, reference_names, captions_all = [], [], [], [], []
    dev = blip_model.device
    for batch_refs, batch_tgts, captions, batch_groups in tqdm(loader):
        batch_groups = np.array(batch_groups).T.tolist()
        captions = [txt_processors["eval"](c) for c in captions]
        ref_feats = _stack_refs(name_to_feat, batch_refs).to(dev)
        sims.append(blip_model.inference(ref_feats, index_features[0].to(dev), captions))
        captions_all += captions
        target_names.extend(batch_tgts)
        group_members.extend(batch_groups)
        reference_names.extend(batch_refs)
    return torch.vstack(sims), reference_names, target_names, group_members, captions_all


def _pct(hits: np.ndarray) -> float:
    # reference: (torch.sum(labels[:, :k]) / len(labels)).item() * 100 -- an fp32 division
    return float(np.float32(hits.sum()) / np.float32(len(hits))) * 100


def cirr_metrics_from_sim(sim: torch.Tensor, ref_idx, tgt_idx, group_idx) -> Tuple[float, ...]:
    """Recall@{1,5,10,50} with the reference image removed + subset Recall@{1,2,3} from exact ranks."""
    ref_idx, tgt_idx, group_idx = (np.asarray(a, dtype=np.int64) for a in (ref_idx, tgt_idx, group_idx))
    listed = np.concatenate([tgt_idx[:, None], ref_idx[:, None], group_idx], axis=1)
    ranks = E.rank_of(sim.contiguous(), torch.from_numpy(listed)).cpu().numpy().astype(np.int64)
    r_t, r_ref, r_g = ranks[:, 0], ranks[:, 1], ranks[:, 2:]
    assert (tgt_idx != ref_idx).all() and (r_t >= 0).all(), "every query needs its target in the gallery, distinct from the reference"
    rank_t = r_t - (r_ref < r_t)                                        # validate_blip.py:258-261: drop the reference
    in_group = (group_idx == tgt_idx[:, None]).sum(1)
    assert (in_group == 1).all(), "target must appear exactly once among the group members"     # :273-274
    member_ok = (group_idx != ref_idx[:, None]) & (group_idx >= 0)     # a member missing from the gallery never matches (:268-271)
    pos_in_group = ((r_g < r_t[:, None]) & member_ok).sum(1)            # :268-271
    return (_pct(pos_in_group < 1), _pct(pos_in_group < 2), _pct(pos_in_group < 3),
            _pct(rank_t < 1), _pct(rank_t < 5), _pct(rank_t < 10), _pct(rank_t < 50))


def compute_cirr_val_metrics(relative_val_dataset, blip_model, index_features, index_names: List[str], txt_processors,
                             reuse_reference_kv: bool = False):
    """-> (group_recall@1, @2, @3, recall@1, @5, @10, @50)   (validate_blip.py:232-285)
    reuse_reference_kv (default off; 16-bit engines): project every distinct reference image's tokens to the cross-attention K|V once
    (`build_reference_kv`) instead of once per query -- the same scores, 23 % fewer query-side flops."""
    rkv = None
    if reuse_reference_kv:
        refs = [item[0] for item in (relative_val_dataset[i] for i in range(len(relative_val_dataset))) if item is not None]
        rkv = build_reference_kv(blip_model, index_names, index_features, refs)
    sim, reference_names, target_names, group_members, _ = generate_cirr_val_predictions(
        blip_model, relative_val_dataset, index_names, index_features, txt_processors, reference_kv=rkv)
    print("Compute CIRR validation metrics")
    n2i = {n: i for i, n in enumerate(index_names)}
    ref = [n2i[n] for n in reference_names]
    tgt = [n2i[n] for n in target_names]
    grp = [[n2i.get(n, -1) for n in g] for g in group_members]
    return cirr_metrics_from_sim(sim, ref, tgt, grp)


# ---- FashionIQ validation ----------------------------------------------------------------------------
def generate_fiq_val_predictions(blip_model, relative_val_dataset, index_names: List[str], index_features, txt_processors,
                                 save_memory: bool = False, batch_size: int = 16, num_workers: int = 4):
    """-> (sim[nq,N], target_names, reference_names, captions)   (validate_blip.py:149-207)"""
    print(f"Compute FashionIQ {getattr(relative_val_dataset, 'dress_types', '')} validation predictions")
    loader = _query_loader(relative_val_dataset, batch_size, num_workers, collate_fn=collate_fn, shuffle=False)
    name_to_feat = dict(zip(index_names, index_features[-1]))
    sims, target_names, reference_names, captions_all = [], [], [], []
    dev = blip_model.device
    for batch_refs, batch_tgts, captions in tqdm(loader):
        flat = np.array(captions).T.flatten().tolist()
        composed = [fiq_compose_caption(flat[i], flat[i + 1]) for i in range(0, len(flat), 2)]     # :180-184
        composed = [txt_processors["eval"](c) for c in composed]
        ref_feats = _stack_refs(name_to_feat, batch_refs).to(dev)
        sims.append(blip_model.inference(ref_feats, index_features[0].to(dev), composed))
        captions_all += composed
        target_names.extend(batch_tgts)
        reference_names.extend(batch_refs)
    return torch.vstack(sims), target_names, reference_names, captions_all


def fiq_metrics_from_sim(sim: torch.Tensor, tgt_idx) -> Tuple[float, float]:
    tgt_idx = np.asarray(tgt_idx, dtype=np.int64)
    r = E.rank_of(sim.contiguous(), torch.from_numpy(tgt_idx[:, None])).cpu().numpy()[:, 0]
    assert (r >= 0).all(), "every query needs its target in the gallery"              # validate_blip.py:51
    return _pct(r < 10), _pct(r < 50)


def compute_fiq_val_metrics(relative_val_dataset, blip_model, index_features, index_names: List[str], txt_processors,
                            save_memory: bool = False) -> Tuple[float, float]:
    """-> (recall@10, recall@50); the reference image is NOT removed   (validate_blip.py:24-57)"""
    sim, target_names, _, _ = generate_fiq_val_predictions(blip_model, relative_val_dataset, index_names, index_features,
                                                           txt_processors, save_memory)
    print(f"Compute FashionIQ {getattr(relative_val_dataset, 'dress_types', '')} validation metrics")
    n2i = {n: i for i, n in enumerate(index_names)}
    return fiq_metrics_from_sim(sim, [n2i[n] for n in target_names])


# ---- CIRR test submission ----------------------------------------------------------------------------
def generate_cirr_test_predictions(blip_model, relative_test_dataset, index_names: List[str], index_features, txt_processors,
                                   batch_size: int = 32, num_workers: int = 4):
    """-> (sim, reference_names, group_members, pairs_id, captions, name_to_feat)   (cirr_test_submission.py:135-190)"""
    print("Compute CIRR test predictions")
    loader = _query_loader(relative_test_dataset, batch_size, num_workers)
    name_to_feat = dict(zip(index_names, index_features[1]))
    pairs_id, group_members, reference_names, sims, captions_all = [], [], [], [], []
    dev = blip_model.device
    for batch_pairs, batch_refs, captions, batch_groups in tqdm(loader):
        batch_groups = np.array(batch_groups).T.tolist()
        captions = [txt_processors["eval"](c) for c in captions]
        ref_feats = _stack_refs(name_to_feat, batch_refs).to(dev)     # (the reference's B==1 branch has an `.unqueeze` typo, :175)
        sims.append(blip_model.inference(ref_feats, index_features[0].to(dev), captions))
        captions_all += captions
        group_members.extend(batch_groups)
        reference_names.extend(batch_refs)
        pairs_id.extend(batch_pairs)
    return torch.vstack(sims), reference_names, group_members, pairs_id, captions_all, name_to_feat


RERANK_TOP = 50          # cirr_test_submission.py:91-92 (`step = 50; top = 50`)


def cirr_test_dicts_from_sim(sim: torch.Tensor, ref_idx, group_idx, pairs_id, index_names: Sequence[str], rerank_fn=None):
    """top-50 names (reference removed) and top-3 subset names per pair id, from top-51 + 6 exact ranks.

    rerank_fn(query_rows[list], cand_idx[len, 50] int64) -> P(match)[len, 50]: the stage-2 branch
    (cirr_test_submission.py:88-112): the first 50 entries of every row are re-sorted by (fl32(1 - P), stage-1 position)
    BEFORE the reference image is removed; entries beyond 50 keep their stage-1 order."""
    ref_idx, group_idx = np.asarray(ref_idx, dtype=np.int64), np.asarray(group_idx, dtype=np.int64)
    names = np.asarray(index_names)
    N = sim.shape[1]
    k = min(51, 64)
    _, idx = E.topk(sim.contiguous(), k)
    idx = idx.cpu().numpy()
    g_rank = E.rank_of(sim.contiguous(), torch.from_numpy(group_idx)).cpu().numpy().astype(np.int64)
    if rerank_fn is not None:
        top = min(RERANK_TOP, N)
        for s in range(0, len(pairs_id), 50):
            rows = list(range(s, min(s + 50, len(pairs_id))))
            cand = idx[rows, :top].astype(np.int64)
            prob = rerank_fn(rows, torch.from_numpy(cand)).detach().float().cpu().numpy()
            perm = np.argsort((np.float32(1.0) - prob.astype(np.float32)), axis=1, kind="stable")
            idx[rows, :top] = np.take_along_axis(cand, perm, axis=1)
        # subset order after reranking: a member inside the top-50 takes its new position, others keep their stage-1 rank
        for q in range(len(pairs_id)):
            where = {int(g): p for p, g in enumerate(idx[q, :top])}
            for j in range(group_idx.shape[1]):
                g = int(group_idx[q, j])
                if g in where:
                    g_rank[q, j] = where[g]
    top, sub = {}, {}
    for q, pid in enumerate(pairs_id):
        row = [i for i in idx[q] if i >= 0 and i != ref_idx[q]][: min(50, N - 1)]       # cirr_test_submission.py:116-120,127
        top[str(int(pid))] = names[row].tolist()
        members = [(g_rank[q, j], group_idx[q, j]) for j in range(group_idx.shape[1])
                   if group_idx[q, j] >= 0 and group_idx[q, j] != ref_idx[q]]
        members.sort()
        sub[str(int(pid))] = names[[m[1] for m in members[:3]]].tolist()                 # :122-124,129-130
    return top, sub


def generate_cirr_test_dicts(relative_test_dataset, blip_model, index_features, index_names: List[str], txt_processors,
                             rerank=False):
    """-> (pairid -> top-50 names, pairid -> top-3 subset names)   (cirr_test_submission.py:61-132)"""
    sim, reference_names, group_members, pairs_id, captions, name_to_feat = generate_cirr_test_predictions(
        blip_model, relative_test_dataset, index_names, index_features, txt_processors)
    print("Compute CIRR prediction dicts")
    n2i = {n: i for i, n in enumerate(index_names)}
    ref = [n2i[n] for n in reference_names]
    grp = [[n2i.get(n, -1) for n in g] for g in group_members]
    rerank_fn = None
    if rerank:                              # cirr_test_submission.py:88-112
        print("reranking now")
        if hasattr(blip_model, "rerank_pairs"):
            # K|V projections once per gallery image (the reference recomputes them inside every (query, candidate) pair)
            kv = blip_model.engine().encode_kv(torch.stack([name_to_feat[n] for n in index_names]).float())

            def rerank_fn(rows, cand):
                r = torch.tensor([ref[q] for q in rows])
                return blip_model.rerank_pairs(kv, r, kv, cand, [captions[q] for q in rows])
        else:                               # any model that follows the reference protocol
            def rerank_fn(rows, cand):
                refs = _stack_refs(name_to_feat, [reference_names[q] for q in rows]).to(blip_model.device)
                tg = torch.stack([name_to_feat[index_names[int(i)]] for i in cand.reshape(-1)]).to(blip_model.device)
                return blip_model.inference_rerank(refs, tg, [captions[q] for q in rows]).view(len(rows), -1)
    return cirr_test_dicts_from_sim(sim, ref, grp, pairs_id, index_names, rerank_fn=rerank_fn)
