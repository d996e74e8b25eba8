"""world_size 2 / 3 gloo tests of the gallery-sharded EVALUATION (sprc_amd/dist_eval.py): encode own slice -> route the
queries to the owner of their reference image -> pad ragged query counts -> fuse -> ShardedRanker (top-k + subset-member
exchange in one all_gather) -> metrics.  Compute callables are oracle-backed doubles here (CPU); tests/test_dist_gpu.py
runs the same orchestration with the HIP kernels on two ranks sharing one GPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sprc_oracle as O
from sprc_amd import synth
from sprc_amd.config import get_config

import _dist_case as DC
from test_dist_cpu import _cpu_sim, _cpu_topk


class OracleModel:
    """CPU double of the model protocol (extract_target_features / fuse) on the fp32 oracle."""
    device = torch.device("cpu")

    def __init__(self, sd, cfg, ids, mask):
        self.sd, self.cfg, self.ids, self.mask = sd, cfg, ids, mask

    def extract_target_features(self, images, mode="mean"):
        with torch.no_grad():
            return O.extract_target_features(self.sd, self.cfg, images)

    def fuse_captions(self, ref, caps):
        rows = [int(c[1:]) for c in caps]
        with torch.no_grad():
            return O.fuse_queries(self.sd, self.cfg, ref, self.ids[rows], self.mask[rows])


def _setup():
    cfg = get_config("pretrain", vit_depth=1, q_layers=2)
    sd = synth.make_state_dict(cfg, seed=9)
    case = DC.build(0)
    return cfg, sd, case, OracleModel(sd, cfg, case["ids"], case["mask"])


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from sprc_amd import dist_eval as DE
    cfg, sd, case, model = _setup()
    gallery = DC.Gallery(case["images"])
    rel = DC.Relative(case["ref"], case["tgt"], case["groups"])
    local = {}

    def sim_fn(fusion, feats):
        s = _cpu_sim(fusion, feats)
        local["sim"] = s.clone()
        return s

    kw = dict(fuse_fn=model.fuse_captions, sim_fn=sim_fn, topk_fn=_cpu_topk)
    q = DE.cirr_val_queries(rel, DC.TXT)
    shard = DE.encode_gallery_shard(gallery, model, reference_names=q.ref_names, num_workers=0, batch_size=8)
    # raw embeddings are kept for local reference images only
    assert set(shard.raw) == {n for n in set(q.ref_names) if shard.base <= shard.name_to_index[n] < int(shard.offsets[rank + 1])}
    cirr = DE.compute_cirr_val_metrics_sharded(rel, gallery, model, DC.TXT, num_workers=0, gallery_batch_size=8, **kw)
    top, sub = DE.generate_cirr_test_dicts_sharded(DC.RelativeTest(case["ref"], case["tgt"], case["groups"]), gallery, model,
                                                   DC.TXT, num_workers=0, gallery_batch_size=8, **kw)
    ref, tgt, grp = (DC.to_kept_index(case, case[k]) for k in ("ref", "tgt", "groups"))
    _, top_idx, _ = DE.sharded_rank(shard, q, listed=None, **kw)
    out[rank] = dict(cirr=cirr, top=top, sub=sub, sim=local["sim"].numpy(), lo=shard.base, names=shard.names,
                     offsets=shard.offsets.tolist(), fiq=DE.fiq_metrics_from_topk(top_idx.numpy(), tgt))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world):
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        return {r: dict(out[r]) for r in range(world)}


def _check(world):
    res = _run(world)
    cfg, sd, case, model = _setup()
    keep = case["keep"]
    names = [f"img-{i:05d}" for i in keep]
    ref, tgt, grp = (DC.to_kept_index(case, case[k]) for k in ("ref", "tgt", "groups"))
    # every rank saw the same global picture and computed the same numbers
    for r in range(world):
        assert res[r]["names"] == names and res[r]["offsets"] == res[0]["offsets"]
        assert res[r]["cirr"] == res[0]["cirr"] and res[r]["top"] == res[0]["top"] and res[r]["sub"] == res[0]["sub"]
    assert res[0]["offsets"][-1] == len(keep) and len(res[0]["offsets"]) == world + 1
    # the shards' local score blocks tile the full [nq, N] matrix: metrics on it (oracle, single process) must be identical
    sim = np.concatenate([res[r]["sim"] for r in range(world)], axis=1)
    assert sim.shape == (DC.NQ, len(keep))
    assert res[0]["cirr"] == O.cirr_metrics(sim, ref, tgt, grp)
    assert res[0]["fiq"] == O.fiq_metrics(sim, tgt)
    o_top, o_sub = O.cirr_test_dicts(sim, ref, grp, [9000 + i for i in range(DC.NQ)], names)
    assert res[0]["top"] == o_top and res[0]["sub"] == o_sub
    # and those scores are the single-process scores (fp32 round-off of different batch compositions at most)
    with torch.no_grad():
        feats_o, raw_o = O.extract_target_features(sd, cfg, case["images"][keep])
        sim_o = O.inference(sd, cfg, raw_o[torch.from_numpy(ref)], feats_o, case["ids"], case["mask"]).numpy()
    np.testing.assert_allclose(sim, sim_o, atol=2e-6, rtol=0)


def test_sharded_evaluation_world2():
    _check(2)


def test_sharded_evaluation_world3_ragged():
    _check(3)


def test_owner_routing_and_offsets():
    from sprc_amd import dist as D
    for n, world in [(69, 2), (69, 3), (7, 8), (2297, 8), (100, 1)]:
        idx = torch.arange(n)
        own = D.owner_of(idx, n, world)
        offs = D.offsets_of(D.shard_bounds(n, world, r)[1] - D.shard_bounds(n, world, r)[0] for r in range(world))
        assert torch.equal(own, D.owner_from_offsets(idx, offs))
        for r in range(world):
            lo, hi = D.shard_bounds(n, world, r)
            assert (own[lo:hi] == r).all()
    offs = D.offsets_of([3, 0, 5])                       # an empty slice in the middle
    assert D.owner_from_offsets(torch.arange(8), offs).tolist() == [0, 0, 0, 2, 2, 2, 2, 2]


def test_missing_targets_and_padding_slots_are_not_hits():
    """a target that is not in the gallery maps to index -1, which is also the filler of unused top-k slots when the
    gallery has fewer than k rows -- it must fail loudly (the reference asserts one label per query), never count as a hit."""
    import pytest
    from sprc_amd.dist_eval import _position, fiq_metrics_from_topk
    top = np.array([[2, 0, 1, -1, -1], [1, 2, 0, -1, -1]], dtype=np.int64)            # a 3-image gallery, k = 5
    assert _position(top, np.array([1, -1])).tolist() == [2, 5]
    assert fiq_metrics_from_topk(top, np.array([1, 0])) == (100.0, 100.0)
    with pytest.raises(AssertionError):
        fiq_metrics_from_topk(top, np.array([1, -1]))
