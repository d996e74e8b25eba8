"""Shared fixtures of the sharded-evaluation tests (CPU gloo doubles and the 2-ranks-on-1-GPU HIP test): a small
synthetic CIRR-like problem -- gallery images (one unreadable), composed queries with targets and 6-member subsets."""
import numpy as np
import torch
from torch.utils.data import Dataset

from sprc_amd import synth
from sprc_amd.tokenizer import TokenBatch

N_IMG, NQ = 70, 23           # > 51 images so that the merged top-51 is a strict subset of the gallery
BAD = 17                     # index of the unreadable image (dropped by collate_fn, like data_utils.py:191-192)


class FakeTokenizer:
    """caption "q<i>" -> row i of pre-drawn (ids, mask): the real WordPiece vocabulary is a network fetch."""

    def __init__(self, ids, mask):
        self.ids, self.mask = ids, mask

    def __call__(self, text, **kw):
        rows = [int(t[1:]) for t in text]
        return TokenBatch(self.ids[rows], self.mask[rows])


class Gallery(Dataset):
    split = "val"

    def __init__(self, images):
        self.images = images
        self.names = [f"img-{i:05d}" for i in range(len(images))]

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return (self.names[i], self.images[i]) if i != BAD else None


class Relative(Dataset):
    def __init__(self, ref, tgt, groups):
        self.ref, self.tgt, self.groups = ref, tgt, groups

    def __len__(self):
        return len(self.ref)

    def __getitem__(self, i):
        return f"img-{self.ref[i]:05d}", f"img-{self.tgt[i]:05d}", f"q{i}", [f"img-{g:05d}" for g in self.groups[i]]


class RelativeTest(Relative):
    def __getitem__(self, i):
        return 9000 + i, f"img-{self.ref[i]:05d}", f"q{i}", [f"img-{g:05d}" for g in self.groups[i]]


def build(seed: int = 0):
    """-> dict(images, keep (indices of readable images), ids, mask, ref, tgt, groups) with ref/tgt/groups as ORIGINAL
    image numbers (all readable); queries are spread unevenly over the gallery so that ranks own different counts."""
    rng = np.random.default_rng(100 + seed)
    images = synth.make_images(N_IMG, seed=200 + seed)
    keep = [i for i in range(N_IMG) if i != BAD]
    ids, mask, _ = synth.make_queries(NQ, N_IMG, seed=300 + seed)
    # two thirds of the references in the first third of the gallery: ragged per-rank query counts
    ref = np.array([keep[int(rng.integers(0, len(keep) // 3))] if q % 3 else keep[int(rng.integers(0, len(keep)))] for q in range(NQ)])
    tgt = np.array([int(rng.choice([k for k in keep if k != ref[q]])) for q in range(NQ)])
    groups = np.stack([rng.permutation(np.array([ref[q], tgt[q], *rng.choice([k for k in keep if k not in (ref[q], tgt[q])], 4, replace=False)]))
                       for q in range(NQ)])
    return dict(images=images, keep=keep, ids=ids, mask=mask, ref=ref, tgt=tgt, groups=groups)


def to_kept_index(case, a):
    """original image numbers -> positions in the gallery after the unreadable image was dropped."""
    pos = {k: i for i, k in enumerate(case["keep"])}
    return np.vectorize(pos.get)(np.asarray(a))


TXT = {"eval": lambda c: c}


# ---- config C4's sizes (CIRR test1: 2 316 gallery images, 4 148 composed queries), images drawn lazily -------------------------------
C4_IMAGES, C4_QUERIES = 2316, 4148


class LazyGallery(Dataset):
    """n images ~ N(0, 1), image i from its own generator (seed * 100003 + i): every rank can read any slice without any rank
    holding the 1.4 GB of pixels."""
    split = "test1"

    def __init__(self, n: int, seed: int):
        self.n, self.seed = n, seed
        self.names = [f"test1-img-{i:05d}" for i in range(n)]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        return self.names[i], torch.randn((3, 224, 224), generator=g)


class C4Test(Dataset):
    """items of the CIRR test split: (pair_id, reference_name, caption, group_members)   (data_utils.py:255-262)"""

    def __init__(self, names, ref, groups):
        self.names, self.ref, self.groups = names, ref, groups

    def __len__(self):
        return len(self.ref)

    def __getitem__(self, i):
        return 12000 + i, self.names[self.ref[i]], f"q{i}", [self.names[g] for g in self.groups[i]]


def build_c4(seed: int = 0):
    rng = np.random.default_rng(500 + seed)
    ids, mask, _ = synth.make_queries(C4_QUERIES, C4_IMAGES, seed=600 + seed)
    ref = rng.integers(0, C4_IMAGES, size=C4_QUERIES)
    groups = np.zeros((C4_QUERIES, 6), dtype=np.int64)
    for q in range(C4_QUERIES):
        others = rng.choice(C4_IMAGES, 8, replace=False)
        groups[q] = rng.permutation(np.array([ref[q], *[o for o in others if o != ref[q]][:5]]))
    return dict(ids=ids, mask=mask, ref=ref, groups=groups)
