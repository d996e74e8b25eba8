"""`Blip2QformerCirAlignPrompt` -- the SPRC model protocol on the MI355X HIP engine.

Drop-in for lavis/models/blip2_models/blip2_qformer_cir_align_prompt.py on the retrieval path:
same class name (it is the checkpoint key, src/utils.py:218-222; src/blip_validate.py:107-109), same
registry name ("blip2_cir_align_prompt", align_prompt.py:25), same state-dict keys
(SURVEY.md section 8(b)), and the two methods the evaluation harness calls:

    extract_target_features(image, mode="mean") -> (feats[B,32,256], raw[B,257,D])     align_prompt.py:364-386
    inference(reference_embeds, target_feats, text) -> sim[B,N]                        align_prompt.py:312-361

All compute goes through libsprc_hip.so (sprc_amd/engine.py); there is no torch/CPU fallback:
calling these methods on a CPU-resident model raises.

Deliberate differences from the reference (documented in DESIGN.md):
  * `extract_target_features` / `inference` / `inference_rerank` have eval semantics always (dropout = identity).  The reference's
    CIRR scripts leave the Q-Former in train mode, which makes their features stochastic.  The TRAINING
    forward honours `model.train()` as the reference does (blip_fine_tune_2.py:290): Q-Former dropout p = 0.1 at the reference's
    four sites, reproducible counter-based masks (`dropout_seed`).
  * `inference` always returns a 2-D [B,N] tensor (the reference's `.squeeze()` collapses B=1 / N=1).
  * `forward` (the three training losses, align_prompt.py:95-200) carries autograd history: its backward runs the HIP backward
    kernels (sprc_amd/train.py, csrc/train.hip), so the reference's training loop runs against this class.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from . import synth
from .train import TrainStep
from .config import MODEL_TYPES, SprcConfig, get_config


class _Node(nn.Module):
    """Bare container so that parameters get the reference's dotted state-dict names."""


def _register(root: nn.Module, dotted: str, shape, device="cpu") -> None:
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    # what the reference trains: everything but the ViT trunk (align_prompt.py:64-69)
    mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape, device=device), requires_grad=not dotted.startswith("visual_encoder.")))


class ReferenceKV:
    """`reference_embeds` of `inference` as rows of precomputed cross-attention K|V projections (`Engine.encode_kv`) instead of raw ViT
    embeddings: kv [n, 257, kv_width], index [B] = the row of each query's reference image.  Optional fast path of the evaluation harness
    (`reuse_reference_kv`): an image that is the reference of several queries is projected once."""

    def __init__(self, kv: torch.Tensor, index: torch.Tensor):
        self.kv, self.index = kv, index
        self.shape = (int(index.shape[0]),) + tuple(kv.shape[1:])

    def to(self, *a, **k):
        return self


class Blip2QformerCirAlignPrompt(nn.Module):
    PRETRAINED_MODEL_CONFIG_DICT = {k: k for k in MODEL_TYPES}     # align_prompt.py:38-42 ("coco" has no CIR use)

    def __init__(self, model_type: str = "pretrain", compute_dtype: str = "fp16", rank_dtype: str = "fp32",
                 cfg: Optional[SprcConfig] = None, max_batch: int = 128, tokenizer=None, device="cpu", train_vit_dtype: str = "fp32",
                 train_products: str = "fp32"):
        """train_vit_dtype: dtype of the FROZEN ViT trunk inside a training step -- "fp32" (gradients within 1e-4 of the reference's fp32
        graph) or "fp16" (what the reference's loop does: the trunk runs under `torch.cuda.amp.autocast`, blip_fine_tune_2.py:293; 64
        ViT-g images per step then cost 45 ms instead of 500).  The Q-Former's forward and backward are fp32 either way."""
        super().__init__()
        self.cfg = cfg if cfg is not None else get_config(model_type)
        self.compute_dtype, self.rank_dtype, self.max_batch = compute_dtype, rank_dtype, max_batch
        if train_vit_dtype not in ("fp32", "fp16"):
            raise ValueError(f"train_vit_dtype {train_vit_dtype!r}")
        self.train_vit_dtype = train_vit_dtype
        # train_products: the products of the TRAINABLE part of a training step -- "fp32" (exact-fp32 MFMA: the parity mode) or "fp16"
        # (fp16 operand copies, fp32 accumulate / outputs / master weights: the reference's autocast arithmetic, ~2.3 x faster per step)
        if train_products not in ("fp32", "fp16"):
            raise ValueError(f"train_products {train_products!r}")
        self.train_products = train_products
        self.max_txt_len = self.cfg.max_txt_len
        for name, shape, _ in synth.param_specs(self.cfg):
            _register(self, name, shape, device)
        self.register_parameter("temp", nn.Parameter(0.07 * torch.ones([], device=device), requires_grad=True))
        self._engine: Optional[E.Engine] = None
        self._tokenizer = tokenizer
        # training mode is nn.Module's default, as for the reference class (lavis load_model_and_preprocess(is_eval=False) leaves it on)
        self.dropout_p = 0.1                       # hidden_dropout_prob = attention_probs_dropout_prob of bert-base-uncased (blip2.py:48)
        self.dropout_seed = 0                      # masks of step n: counter-based, seeded by (dropout_seed, n) -- set it for reproducible runs
        self._drop_step = 0

    # ---- construction helpers (lavis/models/base_model.py:58-80) --------------------------------
    @classmethod
    def from_pretrained(cls, model_type: str, **kw) -> "Blip2QformerCirAlignPrompt":
        assert model_type in cls.PRETRAINED_MODEL_CONFIG_DICT, "Unknown model type {}".format(model_type)
        return cls(model_type=model_type, **kw)

    @classmethod
    def from_config(cls, cfg: dict) -> "Blip2QformerCirAlignPrompt":
        vit = cfg.get("vit_model", "eva_clip_g")                  # align_prompt.py:502-529
        return cls(model_type="pretrain" if vit == "eva_clip_g" else "pretrain_vitL")

    def init_synthetic(self, seed: int = 0) -> "Blip2QformerCirAlignPrompt":
        """Seeded random weights (no checkpoint is reachable offline; see sprc_amd/synth.py)."""
        dev = self.device
        gen_dev = "cpu" if dev.type == "cpu" else str(dev)
        with torch.no_grad():
            params = dict(self.named_parameters())
            for name, t in synth.iter_state_dict(self.cfg, seed, device=gen_dev):
                params[name].copy_(t)
        self._drop_engines()
        return self

    # ---- nn.Module plumbing ---------------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        return list(self.parameters())[0].device                 # base_model.py:25-27

    def _drop_engines(self) -> None:
        """Both packed engines are snapshots of the parameters: the inference engine AND the frozen trunk of the training step
        (`_train_engine`) are rebuilt after anything that replaces or moves the weights."""
        self._engine = None
        self._tengine = None

    def load_state_dict(self, state_dict, strict: bool = True):
        self._drop_engines()
        return super().load_state_dict(state_dict, strict=strict)

    def _apply(self, fn, *a, **k):
        self._drop_engines()
        return super()._apply(fn, *a, **k)

    # `train()` / `eval()` are nn.Module's: the flag decides whether the differentiable `forward` runs the Q-Former's dropout
    # (blip_fine_tune_2.py:290 trains in train mode, :322 validates in eval mode); the inference methods ignore it.

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from .tokenizer import BertWordPieceTokenizer
            self._tokenizer = BertWordPieceTokenizer()
        return self._tokenizer

    @tokenizer.setter
    def tokenizer(self, tok):
        self._tokenizer = tok

    def engine(self) -> E.Engine:
        if self._engine is None:
            if self.device.type != "cuda":
                raise L.SprcError("Blip2QformerCirAlignPrompt runs on the MI355X HIP engine only; move the model to a "
                                  "GPU with .to('cuda') (there is no CPU fallback)")
            sd = {k: v for k, v in self.state_dict().items()}
            self._engine = E.Engine(self.cfg, sd, self.device, dtype=self.compute_dtype, max_batch=self.max_batch)
        return self._engine

    # ---- the protocol ---------------------------------------------------------------------------
    @torch.no_grad()
    def extract_target_features(self, image: torch.Tensor, mode: str = "mean") -> Tuple[torch.Tensor, torch.Tensor]:
        eng = self.engine()
        raw = eng.vit_forward(image)
        feats, _ = eng.qformer_image(raw)
        return feats, raw

    @torch.no_grad()
    def fuse(self, reference_embeds: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """fusion_feats[B,256]: the query side of `inference` with pre-tokenised text."""
        if isinstance(reference_embeds, ReferenceKV):
            fusion, _ = self.engine().qformer_fuse_kv(reference_embeds.kv, reference_embeds.index, input_ids, attention_mask)
            return fusion
        fusion, _ = self.engine().qformer_fuse(reference_embeds, input_ids, attention_mask)
        return fusion

    @torch.no_grad()
    def similarity(self, fusion: torch.Tensor, target_feats: torch.Tensor) -> torch.Tensor:
        dev = self.device
        fusion = fusion.to(device=dev, dtype=torch.float32).contiguous()
        target_feats = target_feats.to(device=dev, dtype=torch.float32).contiguous()
        if target_feats.dim() != 3 or target_feats.shape[1] != 32:
            raise ValueError("target_feats must be [N,32,embed_dim]")
        if self.rank_dtype in ("bf16", "fp16"):
            rdt = torch.bfloat16 if self.rank_dtype == "bf16" else torch.float16
            return E.sim_max(fusion.to(rdt), target_feats.to(rdt))
        return E.sim_max(fusion, target_feats)

    @torch.no_grad()
    def inference_ids(self, reference_embeds, target_feats, input_ids, attention_mask) -> torch.Tensor:
        return self.similarity(self.fuse(reference_embeds, input_ids, attention_mask), target_feats)

    @torch.no_grad()
    def inference(self, reference_embeds: torch.Tensor, target_feats: torch.Tensor, text: List[str]) -> torch.Tensor:
        if isinstance(text, str):
            text = [text]
        if len(text) != reference_embeds.shape[0]:
            raise ValueError("one caption per reference image is required")
        tok = self.tokenizer(text, padding="max_length", truncation=True, max_length=self.max_txt_len,
                             return_tensors="pt").to(self.device)
        return self.inference_ids(reference_embeds, target_feats, tok.input_ids, tok.attention_mask)

    @torch.no_grad()
    def inference_rerank(self, reference_embeds: torch.Tensor, target_embeds: torch.Tensor, text: List[str]) -> torch.Tensor:
        """Stage-2 rerank, `Blip2QformerCirRerank.inference_rerank` (blip2_qformer_cir_rerank.py:399-445):
        reference_embeds [B,257,D], target_embeds [B*T,257,D] (T candidates per query, query-major), text: B captions
        -> P(match) [B*T].  K|V projections of the image tokens are computed once per image instead of once per pair."""
        if isinstance(text, str):
            text = [text]
        B, BT = reference_embeds.shape[0], target_embeds.shape[0]
        if len(text) != B or BT % B:
            raise ValueError("one caption per reference image and T candidates per query are required")
        T = BT // B
        tok = self.tokenizer(text, padding="max_length", truncation=True, max_length=self.max_txt_len,
                             return_tensors="pt").to(self.device)
        eng = self.engine()
        kv_ref, kv_tgt = eng.encode_kv(reference_embeds), eng.encode_kv(target_embeds)
        ia = torch.arange(B, device=self.device).repeat_interleave(T)
        ib = torch.arange(BT, device=self.device)
        return eng.itm(kv_ref, ia, kv_tgt, ib, tok.input_ids.repeat_interleave(T, dim=0), tok.attention_mask.repeat_interleave(T, dim=0))

    @torch.no_grad()
    def rerank_pairs(self, kv_ref: torch.Tensor, ref_index: torch.Tensor, kv_gallery: torch.Tensor, cand_index: torch.Tensor,
                     text: List[str]) -> torch.Tensor:
        """The cached form the harness uses: kv_* from `engine().encode_kv`; query q = (kv_ref[ref_index[q]], text[q]) is
        scored against gallery rows cand_index[q, :] -> P(match) [nq, T]."""
        nq, T = cand_index.shape
        tok = self.tokenizer(text, padding="max_length", truncation=True, max_length=self.max_txt_len,
                             return_tensors="pt").to(self.device)
        ia = ref_index.to(self.device).repeat_interleave(T)
        prob = self.engine().itm(kv_ref, ia, kv_gallery, cand_index.reshape(-1), tok.input_ids.repeat_interleave(T, dim=0),
                                 tok.attention_mask.repeat_interleave(T, dim=0))
        return prob.view(nq, T)

    def forward(self, samples):
        """Training step, align_prompt.py:95-200: {"image", "target", "text_input"} -> {"loss_itc", "loss_rtc", "loss_align"}.
        Dropout: with autograd enabled AND the module in train mode (`model.train()`, the nn.Module default) the Q-Former's dropout runs
        (p = 0.1 at Qformer.py:113,264,293,379; counter-based masks, regenerated in backward); in eval mode, and under torch.no_grad() in
        EITHER mode, dropout is the identity -- no_grad means "measure the losses", as the reference's validation does under model.eval().
        With autograd enabled the losses are outputs of a torch.autograd.Function whose backward
        runs the HIP backward kernels (sprc_amd/train.py) and hands every trainable parameter its gradient, so the reference's loop
        (blip_fine_tune_2.py:293-304: weighted sum, `scaler.scale(loss).backward()`, AdamW step) runs as written; the training graph
        is evaluated on the exact-fp32 engine.  Under torch.no_grad() the losses come from the inference engine in its compute dtype."""
        image, target, text = samples["image"], samples["target"], samples["text_input"]
        tok = self.tokenizer(list(text), padding="max_length", truncation=True, max_length=self.max_txt_len,
                             return_tensors="pt").to(self.device)
        if not torch.is_grad_enabled():
            with torch.no_grad():
                return self.engine().training_losses(image, target, tok.input_ids, tok.attention_mask, temp=float(self.temp))
        names = [n for n, p_ in self.named_parameters() if p_.requires_grad and TrainStep._trains(n)]
        params = dict(self.named_parameters())
        out = _TrainFn.apply(self, image, target, tok.input_ids, tok.attention_mask, names, *[params[n] for n in names])
        self._engine = None                       # the optimizer is about to move the weights the inference engine has packed
        return {"loss_itc": out[0], "loss_rtc": out[1], "loss_align": out[2]}

    def _train_engine(self) -> E.Engine:
        """engine for the frozen ViT trunk of the training step, in `train_vit_dtype` (built once: the trunk does not train)."""
        if getattr(self, "_tengine", None) is None:
            if self.device.type != "cuda":
                raise L.SprcError("training runs on the MI355X HIP engine only; move the model to a GPU with .to('cuda') (there is no CPU fallback)")
            self._tengine = E.Engine(self.cfg, dict(self.state_dict()), self.device, dtype=self.train_vit_dtype, max_batch=self.max_batch,
                                     qformer_x3=0)
        return self._tengine


class _TrainFn(torch.autograd.Function):
    """losses = f(trainable parameters): forward and backward are the HIP training step (sprc_amd/train.py)."""

    @staticmethod
    def forward(ctx, model, image, target, input_ids, attention_mask, names, *params):
        P = {n: p_.detach() for n, p_ in model.named_parameters()}
        # train mode: the Q-Former's dropout (Qformer.py:113,264,293,379; p from the BERT config: 0.1), a fresh mask every step
        drop_p = model.dropout_p if model.training else 0.0
        model._drop_step += 1
        seed = (int(model.dropout_seed) * 0x9E3779B1 + model._drop_step) & 0xFFFFFFFFFFFFFFFF
        step = TrainStep(model.cfg, {n: t.float().contiguous() for n, t in P.items()}, model._train_engine(), dropout_p=drop_p, seed=seed,
                         products=model.train_products)
        losses = step.forward(image.to(model.device), target.to(model.device), input_ids, attention_mask)
        ctx.step, ctx.names = step, names
        return losses["loss_itc"].clone(), losses["loss_rtc"].clone(), losses["loss_align"].clone()

    @staticmethod
    def backward(ctx, g_itc, g_rtc, g_align):
        w = {"loss_itc": float(g_itc), "loss_rtc": float(g_rtc), "loss_align": float(g_align)}     # (GradScaler's scale rides in here)
        G = ctx.step.backward(w)
        ctx.step = None
        return (None,) * 6 + tuple(G[n].view_as(G[n]) for n in ctx.names)


# ---- registry + loader (lavis/common/registry.py:83-110, lavis/models/__init__.py:204-249) -----------
class Blip2QformerCirRerank(Blip2QformerCirAlignPrompt):
    """The stage-2 model class (blip2_qformer_cir_rerank.py:26-27): same trunk and Q-Former, `inference_rerank` + `itm_head`; its
    frozen Q-Former copy (Fformer) is training-only and not loaded.  Its STAGE-1 score differs from align_prompt's: `inference`
    (:373-397) ignores the reference image and the query tokens -- the Q-Former encodes the caption alone, text_proj of its [CLS]
    row is matched against the gallery features."""

    @torch.no_grad()
    def fuse(self, reference_embeds, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        return self.engine().qformer_text(input_ids, attention_mask)      # reference_embeds unused, as in the reference (:373-390)

    def forward(self, samples):
        """The rerank class trains a DIFFERENT objective in the reference (blip2_qformer_cir_rerank.py:100-371: ITM over hard negatives
        mined with its frozen Fformer copy; itm_head receives the gradient) -- not built here.  Inheriting align_prompt's
        differentiable forward would silently train the wrong losses, so a call with autograd enabled is refused; under
        torch.no_grad() the align_prompt losses are still available as a diagnostic."""
        if torch.is_grad_enabled():
            raise NotImplementedError("Blip2QformerCirRerank.forward: the stage-2 training objective (ITM with the frozen Fformer, "
                                      "blip2_qformer_cir_rerank.py:100-371) is not implemented; train stage 1 with "
                                      "blip2_cir_align_prompt, or call under torch.no_grad()")
        return super().forward(samples)


_MODEL_REGISTRY: Dict[str, type] = {"blip2_cir_align_prompt": Blip2QformerCirAlignPrompt,
                                    "blip2_cir_rerank": Blip2QformerCirRerank}


def get_model_class(name: str):
    if name not in _MODEL_REGISTRY:
        raise KeyError(f"model '{name}' is not registered (available: {sorted(_MODEL_REGISTRY)}); the reference "
                       f"scripts' default --blip-model-name points at sources it does not ship (SURVEY.md 2 row 25)")
    return _MODEL_REGISTRY[name]


def load_model_and_preprocess(name: str, model_type: str, is_eval: bool = False, device="cpu", **model_kw):
    """-> (model, vis_processors, txt_processors), as lavis.models.load_model_and_preprocess."""
    from .processors import BlipCaptionProcessor
    model = get_model_class(name).from_pretrained(model_type=model_type, **model_kw)
    if is_eval:
        model.eval()
    txt = {"train": BlipCaptionProcessor(), "eval": BlipCaptionProcessor()}
    return model.to(device), {"train": None, "eval": None}, txt
