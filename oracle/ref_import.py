"""Import the Python reference (/root/reference) UNMODIFIED inside the build container.

TEST INFRASTRUCTURE.  Used only by oracle/gen_golden.py (which writes tests/golden/) and
by the optional cross-check tests that skip when /root/reference is absent (it does not
exist on the GPU box).  Nothing here is copied from the reference: this file only
installs the import shims that SURVEY.md section 8(c) lists, because the container lacks
timm / omegaconf / iopath / fairscale / torchvision / cv2 and has transformers 5.x
instead of the pinned 4.36.2 (requirements.txt:9).
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import torch
import torch.nn as nn

REF = Path("/root/reference")
SRC = REF / "src"


def available() -> bool:
    return (SRC / "lavis" / "models" / "eva_vit.py").is_file()


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _ns(name: str, path: Path) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = [str(path)]
    sys.modules[name] = m
    return m


_installed = False


def install_shims() -> None:
    """Make lavis.models.{eva_vit,clip_vit,base_model}, blip2_models.{Qformer,blip2,
    blip2_qformer_cir_align_prompt} and src/{validate_blip,utils,data_utils,
    cirr_test_submission}.py importable without executing the lavis package __init__s."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("/root/reference is not present")
    sys.dont_write_bytecode = True                      # reference tree is read-only by policy
    import transformers                                  # noqa: F401  (before the fake timm: it probes find_spec)
    import transformers.modeling_utils as mu
    from transformers.pytorch_utils import apply_chunking_to_forward
    import transformers.models.bert.configuration_bert   # noqa: F401
    L = SRC / "lavis"
    _ns("lavis", L)
    _ns("lavis.models", L / "models")
    _ns("lavis.common", L / "common")
    _ns("lavis.models.blip2_models", L / "models" / "blip2_models")
    _ns("lavis.models.blip_models", L / "models" / "blip_models")
    _ns("lavis.processors", L / "processors")

    # timm / fairscale / omegaconf / iopath stand-ins (import-time names only)
    def drop_path(x, drop_prob: float = 0.0, training: bool = False):
        return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", drop_path=drop_path, to_2tuple=to_2tuple,
         trunc_normal_=lambda t, std=1.0, **k: nn.init.trunc_normal_(t, std=std))
    _mod("timm.models.registry", register_model=lambda f: f)
    _mod("timm.models.hub", download_cached_file=None, get_cache_dir=None)
    _mod("fairscale")
    _mod("fairscale.nn")
    _mod("fairscale.nn.checkpoint")
    _mod("fairscale.nn.checkpoint.checkpoint_activations", checkpoint_wrapper=lambda m, *a, **k: m)

    class _OmegaConf:
        @staticmethod
        def load(path):
            raise RuntimeError("OmegaConf is not available (shim)")

        @staticmethod
        def create(*a, **k):
            return {}

    _mod("omegaconf", OmegaConf=_OmegaConf)
    _mod("lavis.common.utils", is_url=lambda s: False, get_abs_path=lambda s: str(L / s))
    _mod("lavis.common.dist_utils", download_cached_file=lambda *a, **k: (_ for _ in ()).throw(
        RuntimeError("no network")), is_dist_avail_and_initialized=lambda: False, is_main_process=lambda: True,
        get_rank=lambda: 0, get_world_size=lambda: 1, main_process=lambda f: f)
    _mod("lavis.common.logger", MetricLogger=object)

    # transformers 5.x drift (Qformer.py:39-44, :703, :943)
    if not hasattr(mu, "apply_chunking_to_forward"):
        mu.apply_chunking_to_forward = apply_chunking_to_forward
    for nm in ("find_pruneable_heads_and_indices", "prune_linear_layer"):
        if not hasattr(mu, nm):
            setattr(mu, nm, lambda *a, **k: None)

    import importlib
    registry = importlib.import_module("lavis.common.registry")
    base_model = importlib.import_module("lavis.models.base_model")
    sys.modules["lavis.models"].BaseModel = base_model.BaseModel          # registry.py:95
    # blip_outputs (dataclasses only) imports cleanly
    _installed = True


def build_reference_model(cfg, state_dict, eval_mode: bool = True, variant: str = "align_prompt", gpu_numerics: bool = False):
    """Construct the reference's Blip2QformerCirAlignPrompt (CPU, fp32) with `cfg` depths and load
    `state_dict` into it.  Returns the nn.Module.  Network-only constructors are replaced by the
    same constructor calls without the download (SURVEY.md 8(c) shim 8).

    gpu_numerics=True reproduces, on the CPU, the arithmetic the reference runs on a GPU (vit_precision="fp16", the class default):
    the trunk's Conv / Linear weights go through the reference's OWN convert_weights_to_fp16 (eva_vit.py:410-425 / the clip_vit.py
    twin, called where create_eva_vit_g :452-454 calls it: after the weights are loaded), the Q-Former, ln_vision and the heads stay
    fp32, and `maybe_autocast` (blip2.py:36-44: "if on cpu, don't use autocast") is replaced by the CPU autocast context of the same
    dtype -- the only line of behaviour that is changed.  Under it linear / conv / matmul run on fp16 operands and return fp16, the
    residual stream, LayerNorm statistics and softmax stay fp32 (their inputs are fp32), as under torch.cuda.amp.autocast."""
    install_shims()
    import importlib
    from functools import partial
    from transformers import BertConfig

    eva = importlib.import_module("lavis.models.eva_vit")
    clipv = importlib.import_module("lavis.models.clip_vit")
    qf = importlib.import_module("lavis.models.blip2_models.Qformer")
    blip2 = importlib.import_module("lavis.models.blip2_models.blip2")

    v = cfg.vit

    def create_eva_vit_g(img_size=224, drop_path_rate=0.4, use_checkpoint=False, precision="fp16"):
        return eva.VisionTransformer(img_size=img_size, patch_size=14, use_mean_pooling=False, embed_dim=1408,
                                     depth=v.depth, num_heads=1408 // 88, mlp_ratio=4.3637, qkv_bias=True,
                                     drop_path_rate=drop_path_rate, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                     use_checkpoint=use_checkpoint)       # eva_vit.py:429-441 minus download

    def create_clip_vit_L(img_size=224, use_checkpoint=False, precision="fp16"):
        return clipv.VisionTransformer(input_resolution=img_size, patch_size=14, width=1024, layers=v.depth,
                                       heads=16, use_grad_checkpointing=use_checkpoint)   # clip_vit.py:243-250

    blip2.create_eva_vit_g = create_eva_vit_g
    blip2.create_clip_vit_L = create_clip_vit_L

    class _FakeTok:
        """Stands in for BertTokenizer (network fetch, blip2.py:30-34): len == 30523; the golden
        script feeds pre-tokenised ids through `set_next`."""
        def __init__(self):
            self._next = None

        def __len__(self):
            return 30523

        def set_next(self, input_ids, attention_mask):
            self._next = (input_ids, attention_mask)

        def __call__(self, text, **kw):
            ids, mask = self._next
            out = types.SimpleNamespace(input_ids=ids, attention_mask=mask)
            out.to = lambda device: out
            return out

    blip2.Blip2Base.init_tokenizer = classmethod(lambda cls, truncation_side="right": _FakeTok())

    qcfg = cfg.qformer
    BertConfig.from_pretrained = classmethod(
        lambda cls, *a, **k: BertConfig(num_hidden_layers=qcfg.layers))   # defaults == bert-base-uncased
    qf.BertPreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    qf.BertModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
    qf.BertModel.invert_attention_mask = lambda self, m: (1.0 - m[:, None, None, :].to(torch.float32)) * \
        torch.finfo(torch.float32).min
    qf.BertLMHeadModel.from_pretrained = classmethod(lambda cls, name, config=None, **k: cls(config))

    vit_model = "eva_clip_g" if v.kind == "eva_g" else "clip_L"
    torch.manual_seed(0)
    if variant == "rerank":            # the stage-2 model class (blip2_qformer_cir_rerank.py): same trunk + a frozen copy it
        rr = importlib.import_module("lavis.models.blip2_models.blip2_qformer_cir_rerank")      # never uses at inference
        model = rr.Blip2QformerCirRerank(vit_model=vit_model, vit_precision="fp32")
        allowed_missing = ("Qformer.cls.", "Qformer.bert.embeddings.position_ids", "Fformer.", "query_tokens_f", "vision_proj_f.", "text_proj_f.")
    else:
        ap = importlib.import_module("lavis.models.blip2_models.blip2_qformer_cir_align_prompt")
        model = ap.Blip2QformerCirAlignPrompt(vit_model=vit_model, vit_precision="fp32")
        allowed_missing = ("Qformer.cls.", "itm_head.", "Qformer.bert.embeddings.position_ids")
    msg = model.load_state_dict(state_dict, strict=False)
    bad = [k for k in msg.missing_keys if not k.startswith(allowed_missing)]
    if variant == "rerank":
        msg.unexpected_keys[:] = [k for k in msg.unexpected_keys if k != "prompt_tokens"]
    if bad or msg.unexpected_keys:
        raise RuntimeError(f"state-dict mismatch: missing={bad[:8]} unexpected={msg.unexpected_keys[:8]}")
    model = model.float()
    if gpu_numerics:
        vmod = eva if v.kind == "eva_g" else clipv
        vmod.convert_weights_to_fp16(model.visual_encoder)
        model.maybe_autocast = lambda dtype=torch.float16: torch.autocast("cpu", dtype=dtype)
    if eval_mode:
        model.eval()
    return model


def import_harness():
    """Import src/validate_blip.py and src/cirr_test_submission.py (metrics code) with stubs for cv2 /
    torchvision / clip and a stub lavis.models.load_model_and_preprocess."""
    install_shims()
    import importlib
    _mod("cv2")
    tv = _mod("torchvision")
    tr = _mod("torchvision.transforms", Compose=object, Resize=object, CenterCrop=object, ToTensor=object,
              Normalize=object)
    tf = _mod("torchvision.transforms.functional", pad=lambda *a, **k: None)
    tv.transforms = tr
    tr.functional = tf
    sys.modules["lavis.models"].load_model_and_preprocess = lambda *a, **k: None
    if str(SRC) not in sys.path:
        sys.path.insert(0, str(SRC))
    vb = importlib.import_module("validate_blip")
    cts = importlib.import_module("cirr_test_submission")
    return vb, cts
