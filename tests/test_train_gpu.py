"""Training forward: the three losses of Blip2QformerCirAlignPrompt.forward on the HIP engine
against the numbers the unmodified REFERENCE produced (tests/golden/train_eva.npz, eval mode) and against the oracle on a
ViT-L case; the building-block loss kernels against plain torch expressions."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sprc_oracle as O  # noqa: E402
from sprc_amd import _lib as L  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402
from sprc_amd.model import Blip2QformerCirAlignPrompt  # noqa: E402
from sprc_amd.tokenizer import TokenBatch  # noqa: E402

DEV = "cuda:0"


class _Tok:
    def __init__(self, ids, mask):
        self.ids, self.mask = ids, mask

    def __call__(self, text, **kw):
        return TokenBatch(self.ids, self.mask)


@pytest.mark.parametrize("dtype,tol", [("fp32", 5e-5), ("bf16", 2e-2)])
def test_forward_losses_match_reference(golden_dir, dtype, tol):
    g = np.load(golden_dir / "train_eva.npz", allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    B = int(g["batch"])
    model = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype=dtype, max_batch=8)
    assert not model.load_state_dict(sd, strict=False).missing_keys
    model = model.to(DEV)
    model.tokenizer = _Tok(torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    images = synth.make_images(2 * B, seed=int(g["seed"]))
    out = model({"image": images[:B].to(DEV), "target": images[B:].to(DEV), "text_input": ["caption"] * B})
    torch.cuda.synchronize()
    got = {k: float(v) for k, v in out.items()}
    print(f"\n[train forward {dtype}]", {k: round(v, 6) for k, v in got.items()}, "reference:", {k: round(float(g[k]), 6) for k in got})
    assert set(got) == {"loss_itc", "loss_rtc", "loss_align"}
    for k in got:
        assert got[k] == pytest.approx(float(g[k]), abs=tol), k
    assert not any(v.requires_grad for v in out.values())                     # forward only: no autograd history


def test_forward_losses_vitl_against_the_oracle():
    cfg = get_config("pretrain_vitL", vit_depth=1)
    sd = synth.make_state_dict(cfg, seed=8)
    B = 7
    images = synth.make_images(2 * B, seed=9)
    ids, mask, _ = synth.make_queries(B, B, seed=10)
    with torch.no_grad():
        want = O.training_losses(sd, cfg, images[:B], images[B:], ids, mask)
    from sprc_amd import engine as E
    eng = E.Engine(cfg, sd, DEV, dtype="fp32", max_batch=8)
    got = eng.training_losses(images[:B].to(DEV), images[B:].to(DEV), ids, mask, temp=float(sd["temp"]))
    for k in want:
        assert float(got[k]) == pytest.approx(float(want[k]), abs=5e-5), k


def test_loss_kernels():
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    for B in (1, 5, 64, 200):
        sim = torch.rand((B, B + 3), generator=g) * 2 - 1
        want = torch.nn.functional.cross_entropy(sim[:, :B].double() / 0.07, torch.arange(B))
        d, out = sim.to(DEV), torch.zeros(1, device=DEV)
        L.check(lib.sprc_contrastive_ce(d.data_ptr(), B + 3, B, 0.07, out.data_ptr(), st))
        assert float(out) == pytest.approx(float(want), rel=2e-6, abs=1e-6)
    h, prompt = torch.randn((6, 64, 768), generator=g), torch.randn((32, 768), generator=g)
    want = torch.nn.functional.mse_loss(h[:, :32].double().mean(1), prompt.double().mean(0).expand(6, -1))
    hd, pd, out = h.to(DEV), prompt.to(DEV), torch.zeros(1, device=DEV)
    L.check(lib.sprc_align_mse(hd.data_ptr(), 64 * 768, 32, 768, pd.data_ptr(), 6, out.data_ptr(), st))
    assert float(out) == pytest.approx(float(want), rel=1e-5)
