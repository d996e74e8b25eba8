"""Multi-rank GPU tests on ONE device (the GPU box has a single MI355X): two processes share cuda:0, every kernel on the
path is the HIP library's, and the two tiny exchanges of sprc_amd/dist.py go through gloo (RCCL refuses two ranks on one
device; on an 8-GPU node the same code runs over RCCL with one rank per GPU).

* sharded CIRR evaluation (sprc_amd/dist_eval.py) == the single-process harness: identical score bits, identical top-k,
  identical Recall@K / subset recall / submission dicts;
* `bench.py --gpus 2` outside a launcher spawns two ranks itself and prints ONE line with n_gpus = 2.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

import _dist_case as DC  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

DEV = "cuda:0"


def _model(case):
    from sprc_amd.model import Blip2QformerCirAlignPrompt
    cfg = get_config("pretrain", vit_depth=2)
    model = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype="fp32", max_batch=32)
    assert not model.load_state_dict(synth.make_state_dict(cfg, seed=11), strict=False).missing_keys
    model = model.to(DEV)
    model.tokenizer = DC.FakeTokenizer(case["ids"], case["mask"])
    return model


def _worker(rank, world, port, out, backend="gloo"):
    """backend "gloo": all ranks share cuda:0 (the 1-GPU box); "nccl": one rank per device over RCCL (a box with >= `world` GPUs)."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = f"cuda:{rank}" if backend == "nccl" else DEV
    torch.cuda.set_device(torch.device(dev))
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from sprc_amd import dist_eval as DE
    from sprc_amd import engine as E
    case = DC.build(0)
    model = _model(case).to(dev)
    gallery = DC.Gallery(case["images"])
    rel = DC.Relative(case["ref"], case["tgt"], case["groups"])
    rec = {}

    def sim_fn(fusion, feats):
        rec["sim"] = E.sim_max(fusion.contiguous(), feats.contiguous())
        return rec["sim"]

    cirr = DE.compute_cirr_val_metrics_sharded(rel, gallery, model, DC.TXT, num_workers=0, gallery_batch_size=16, sim_fn=sim_fn)
    q = DE.cirr_val_queries(rel, DC.TXT)
    shard = DE.encode_gallery_shard(gallery, model, reference_names=q.ref_names, num_workers=0, batch_size=16)
    top_v, top_i, _ = DE.sharded_rank(shard, q, model, sim_fn=sim_fn)
    top, sub = DE.generate_cirr_test_dicts_sharded(DC.RelativeTest(case["ref"], case["tgt"], case["groups"]), gallery, model,
                                                   DC.TXT, num_workers=0, gallery_batch_size=16)
    torch.cuda.synchronize()
    out[rank] = dict(cirr=cirr, top=top, sub=sub, sim=rec["sim"].cpu().numpy(), top_v=top_v.cpu().numpy(),
                     top_i=top_i.cpu().numpy(), n_raw=len(shard.raw), n_local=shard.feats.shape[0])
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _two_rank_check(backend):
    from sprc_amd import engine as E
    from sprc_amd import harness as H
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out, backend), nprocs=world, join=True)
        res = {r: dict(out[r]) for r in range(world)}
    case = DC.build(0)
    model = _model(case)
    gallery = DC.Gallery(case["images"])
    (feats, raw), names = H.extract_index_blip_features(gallery, model, batch_size=16, num_workers=0)
    rel = DC.Relative(case["ref"], case["tgt"], case["groups"])
    want_cirr = H.compute_cirr_val_metrics(rel, model, (feats, raw), names, DC.TXT)
    sim_1, *_ = H.generate_cirr_val_predictions(model, rel, names, (feats, raw), DC.TXT, num_workers=0)
    want_top, want_sub = H.generate_cirr_test_dicts(DC.RelativeTest(case["ref"], case["tgt"], case["groups"]), model, (feats, raw),
                                                    names, DC.TXT)
    want_v, want_i = E.topk(sim_1.contiguous(), 51)
    sim_sharded = np.concatenate([res[r]["sim"] for r in range(world)], axis=1)
    assert sum(res[r]["n_local"] for r in range(world)) == len(names) == DC.N_IMG - 1
    # the exact-fp32 engine is batch-invariant: the shards' score blocks are the single-process scores, bit for bit
    np.testing.assert_array_equal(sim_sharded, sim_1.cpu().numpy())
    for r in range(world):
        np.testing.assert_array_equal(res[r]["top_i"], want_i.cpu().numpy())
        np.testing.assert_array_equal(res[r]["top_v"], want_v.cpu().numpy())
        assert res[r]["cirr"] == want_cirr
        assert res[r]["top"] == want_top and res[r]["sub"] == want_sub
        assert res[r]["n_raw"] < res[r]["n_local"]            # raw embeddings only for local reference images


def test_two_ranks_on_one_gpu_equal_the_single_process_harness():
    _two_rank_check("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device (the test box has one; "
                    "the first box with two proves N > 1 over RCCL -- until then N > 1 over RCCL is UNMEASURED)")
def test_two_rccl_ranks_one_per_gpu_equal_the_single_process_harness():
    """VERDICT r3 missing #2 / item 6(a): TWO ranks over RCCL, one per device -- both all_gather_into_tensor exchanges on device
    memory across xGMI, owner-routed fusion, the k x R merge -- against the single-process harness: score bits, top-51, metrics,
    submission dicts."""
    _two_rank_check("nccl")


def _c4_worker(rank, world, port, out, depth):
    import time
    t0 = time.time()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)                     # eight ranks on the box's 16 cores: the CPU-side image draws must not oversubscribe them
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sprc_amd import dist_eval as DE
    case = DC.build_c4(0)
    model = _c4_model(case, depth)
    gallery = DC.LazyGallery(DC.C4_IMAGES, seed=7)
    top, sub = DE.generate_cirr_test_dicts_sharded(DC.C4Test(gallery.names, case["ref"], case["groups"]), gallery, model, DC.TXT,
                                                   num_workers=0, gallery_batch_size=64)
    torch.cuda.synchronize()
    out[rank] = dict(top=top, sub=sub, seconds=time.time() - t0)
    dist.barrier()
    dist.destroy_process_group()


def _c4_model(case, depth):
    from sprc_amd.model import Blip2QformerCirAlignPrompt
    cfg = get_config("pretrain", vit_depth=depth)
    model = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype="fp32", max_batch=64)
    assert not model.load_state_dict(synth.make_state_dict(cfg, seed=12), strict=False).missing_keys
    model = model.to(DEV)
    model.tokenizer = DC.FakeTokenizer(case["ids"], case["mask"])
    return model


def test_c4_sizes_eight_ranks_sharing_the_gpu_equal_the_single_process_submission():
    """BASELINE config C4 at its REAL sizes -- CIRR test1: 2 316 gallery images, 4 148 composed queries, the gallery in 8 shards -- through
    `generate_cirr_test_dicts_sharded` (cirr_test_submission.py:61-132) with eight ranks that share the box's one GPU (gloo staging: a
    plumbing run, not a scaling number; the full-depth ViT-g and the full Q-Former, exact-fp32 engine so that the shards' scores are the
    single process's bit for bit): ragged shards (2316 = 8 x 289 + 4),
    owner-routed fusion with unequal per-rank query counts, both exchanges, the 8 x 51-candidate merge.  The two submission dicts (top-50
    names, subset top-3) of every rank must equal the single-process harness's, entry for entry."""
    from sprc_amd import harness as H
    import time
    world, depth = 8, int(os.environ.get("SPRC_C4_TEST_DEPTH", "39"))      # the FULL 39-block ViT-g since round 6 (rounds 2-5 truncated it to one block)
    t0 = time.time()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_c4_worker, args=(world, _free_port(), out, depth), nprocs=world, join=True)
        res = {r: dict(out[r]) for r in range(world)}
    t1 = time.time()
    case = DC.build_c4(0)
    model = _c4_model(case, depth)
    gallery = DC.LazyGallery(DC.C4_IMAGES, seed=7)
    (feats, raw), names = H.extract_index_blip_features(gallery, model, batch_size=64, num_workers=0)
    assert len(names) == DC.C4_IMAGES
    want_top, want_sub = H.generate_cirr_test_dicts(DC.C4Test(gallery.names, case["ref"], case["groups"]), model, (feats, raw), names, DC.TXT)
    assert len(want_top) == DC.C4_QUERIES and all(len(v) == 50 for v in want_top.values()) and all(len(v) == 3 for v in want_sub.values())
    print(f"\n[C4 sizes] 8 ranks sharing the GPU: {t1 - t0:.0f} s wall (rank bodies {min(res[r]['seconds'] for r in res):.0f} .. "
          f"{max(res[r]['seconds'] for r in res):.0f} s); single process: {time.time() - t1:.0f} s")
    for r in range(world):
        assert res[r]["top"] == want_top and res[r]["sub"] == want_sub, f"rank {r}"


def test_bench_gpus2_spawns_two_ranks():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["sharding"] == "gallery-sharded x2"
    ngpu = torch.cuda.device_count()
    assert ("gloo" in d["config"]["backend"]) == (ngpu < 2)
    assert abs(d["value"] - 2 * d["config"]["batch"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3


def _rccl_world1_worker(_rank, port, out):
    """ONE rank, backend nccl (= RCCL) on cuda:0, SPRC_DIST_ALWAYS_EXCHANGE=1: the all_gather_into_tensor branch of
    sprc_amd/dist.py (_all_gather_rows), all_gather_object on device, and the k x R merge run exactly as on an 8-GPU node."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["SPRC_DIST_ALWAYS_EXCHANGE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    from sprc_amd import dist as D
    from sprc_amd import dist_eval as DE
    from sprc_amd import engine as E
    calls = {"n": 0}
    real = dist.all_gather_into_tensor

    def counting(o, i, group=None, **kw):
        assert o.is_cuda and i.is_cuda                          # the payload stays on the device
        calls["n"] += 1
        return real(o, i, group=group, **kw)

    dist.all_gather_into_tensor = counting
    # (1) ShardedRanker directly: exchanges on vs off
    g = torch.Generator(device="cuda").manual_seed(5)
    feats = torch.nn.functional.normalize(torch.randn((300, 32, 256), generator=g, device="cuda"), dim=-1)
    fusion = torch.nn.functional.normalize(torch.randn((40, 256), generator=g, device="cuda"), dim=-1)
    listed = torch.randint(-1, 300, (40, 7), generator=torch.Generator().manual_seed(6))
    on = D.ShardedRanker(feats, 1000).rank(fusion, 51, listed=listed + 1000 * (listed >= 0))
    assert calls["n"] == 2                                       # fused vectors + ONE top-k / listed-score payload
    off = D.ShardedRanker(feats, 1000, always_exchange=False).rank(fusion, 51, listed=listed + 1000 * (listed >= 0))
    for a, b in zip(on, off):
        assert torch.equal(a, b)
    # (2) the sharded CIRR evaluation end to end
    case = DC.build(0)
    model = _model(case)
    gallery = DC.Gallery(case["images"])
    rel = DC.Relative(case["ref"], case["tgt"], case["groups"])
    n0 = calls["n"]
    cirr = DE.compute_cirr_val_metrics_sharded(rel, gallery, model, DC.TXT, num_workers=0, gallery_batch_size=16)
    top, sub = DE.generate_cirr_test_dicts_sharded(DC.RelativeTest(case["ref"], case["tgt"], case["groups"]), gallery, model, DC.TXT,
                                                   num_workers=0, gallery_batch_size=16)
    torch.cuda.synchronize()
    out["res"] = dict(cirr=cirr, top=top, sub=sub, gathers=calls["n"] - n0)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_group_of_one_rank_runs_the_device_side_exchanges():
    """VERDICT r2 missing #2: the `nccl` branch had never executed.  A world-size-1 RCCL group on the one GPU of the box, with the
    exchanges forced on, against the single-process harness."""
    from sprc_amd import harness as H
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_rccl_world1_worker, args=(_free_port(), out), nprocs=1, join=True)
        res = dict(out["res"])
    case = DC.build(0)
    model = _model(case)
    gallery = DC.Gallery(case["images"])
    (feats, raw), names = H.extract_index_blip_features(gallery, model, batch_size=16, num_workers=0)
    rel = DC.Relative(case["ref"], case["tgt"], case["groups"])
    assert res["cirr"] == H.compute_cirr_val_metrics(rel, model, (feats, raw), names, DC.TXT)
    want_top, want_sub = H.generate_cirr_test_dicts(DC.RelativeTest(case["ref"], case["tgt"], case["groups"]), model, (feats, raw),
                                                    names, DC.TXT)
    assert res["top"] == want_top and res["sub"] == want_sub
    assert res["gathers"] == 4                                    # two evaluations x (fused vectors, top-k payload)
