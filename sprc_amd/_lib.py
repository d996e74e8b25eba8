"""ctypes binding of libsprc_hip.so (include/sprc.h).

The library is the only compute path of the product: there is no CPU fallback.  `load()`
raises if the shared object is missing (build it with `python -m sprc_amd.build`).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# SPRC_LIB_PATH: an A/B build of the same library (tools/build_variant.sh); never a different implementation
LIB_PATH = Path(os.environ.get("SPRC_LIB_PATH") or (Path(__file__).resolve().parent / "libsprc_hip.so"))

SPRC_F16X3 = 4                                        # storage layout of split-precision activations: rows [hi fp16 | lo e4m3 | hi e4m3] (sprc.h)
SPRC_F32, SPRC_BF16, SPRC_F16, SPRC_FP8 = 0, 1, 2, 3   # F16: IEEE half (compute dtype since ABI 3); FP8: OCP e4m3fn operands
ABI_VERSION = 6
ACT_NONE, ACT_GELU, ACT_QUICKGELU = 0, 1, 2
FP8_ALL, FP8_MLP = 1, 2                               # sprc_vit_model.fp8: qkv + fc1 + fc2, or fc1 + fc2 only, on e4m3fn operands
DTYPES = {"fp32": SPRC_F32, "f32": SPRC_F32, "bf16": SPRC_BF16, "fp16": SPRC_F16, "f16": SPRC_F16,
          "fp8": SPRC_BF16}     # "fp8" engine: a 16-bit model (Engine(fp8_base=...): bf16 by default) + fp8 ViT GEMMs


def is16(dt: int) -> bool:
    """16-bit MFMA operand engines (bf16, or fp16 = the reference's GPU autocast precision)."""
    return dt in (SPRC_BF16, SPRC_F16)

vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class RowMap(C.Structure):
    _fields_ = [("rows_per_group", i32), ("group_stride", i32), ("group_offset", i32)]


class GemmArgs(C.Structure):
    _fields_ = [("M", i32), ("N", i32), ("K", i32), ("dtype", i32), ("out_dtype", i32), ("act", i32), ("max32", i32),
                ("A", vp), ("lda", i64), ("amap", RowMap), ("W", vp), ("ldw", i64), ("bias", vp),
                ("resid", vp), ("ldr", i64), ("C", vp), ("ldc", i64), ("cmap", RowMap),
                ("scratch", vp), ("scratch_bytes", C.c_size_t), ("w_scale", vp), ("a_scale", f32), ("out_scale", f32),
                ("k_alg", i32), ("k8", i32)]


class LayerNormArgs(C.Structure):
    _fields_ = [("M", i32), ("D", i32), ("out_dtype", i32), ("x", vp), ("ldx", i64), ("xmap", RowMap),
                ("gamma", vp), ("beta", vp), ("eps", f32), ("y32", vp), ("ld32", i64), ("ymap", RowMap),
                ("y16", vp), ("ld16", i64), ("add16", vp), ("ld_add", i64), ("sum32", vp), ("ld_sum", i64),
                ("y16_scale", f32)]


class AttentionArgs(C.Structure):
    _fields_ = [("B", i32), ("H", i32), ("Tq", i32), ("Tk", i32), ("head_dim", i32), ("dtype", i32),
                ("q", vp), ("ldq", i64), ("k", vp), ("ldk", i64), ("v", vp), ("ldv", i64), ("out", vp), ("ldo", i64),
                ("key_mask", vp), ("scale", f32),
                ("k2", vp), ("ldk2", i64), ("v2", vp), ("ldv2", i64), ("Tk2", i32), ("kv_index", vp), ("kv2_index", vp),
                ("drop_p", f32), ("drop_site", C.c_uint32), ("drop_seed", C.c_uint64), ("out_x3", i32)]


class QformerEmbedArgs(C.Structure):
    _fields_ = [("B", i32), ("Lq", i32), ("Lt", i32), ("hidden", i32), ("out_dtype", i32), ("vocab", i32),
                ("query_embeds", vp), ("q_bstride", i64), ("input_ids", vp), ("word_emb", vp), ("pos_emb", vp),
                ("gamma", vp), ("beta", vp), ("eps", f32), ("y32", vp), ("y16", vp), ("no_img", i32)]


class AttentionBwdArgs(C.Structure):
    _fields_ = [("B", i32), ("H", i32), ("Tq", i32), ("Tk", i32), ("head_dim", i32),
                ("q", vp), ("k", vp), ("v", vp), ("dout", vp), ("ldq", i64), ("ldk", i64), ("ldv", i64), ("lddo", i64),
                ("key_mask", vp), ("scale", f32), ("dq", vp), ("dk", vp), ("dv", vp), ("lddq", i64), ("lddk", i64), ("lddv", i64),
                ("scratch", vp), ("scratch_bytes", sz), ("drop_p", f32), ("drop_site", C.c_uint32), ("drop_seed", C.c_uint64)]


class ProfEntry(C.Structure):
    _fields_ = [("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double), ("launches", i64), ("busy_ms", C.c_double),
                ("exec_flops", C.c_double)]


K_CLASSES = ("gemm_bf16", "gemm_f32", "attention", "rowops", "rank")


class Linear(C.Structure):
    _fields_ = [("w", vp), ("b", vp)]


class VitLayer(C.Structure):
    _fields_ = [("ln1_w", vp), ("ln1_b", vp), ("ln2_w", vp), ("ln2_b", vp),
                ("qkv", Linear), ("proj", Linear), ("fc1", Linear), ("fc2", Linear),
                ("qkv_ws", vp), ("fc1_ws", vp), ("fc2_ws", vp), ("s_ln1", f32), ("s_ln2", f32), ("s_mlp", f32)]


class VitModel(C.Structure):
    _fields_ = [("dtype", i32), ("width", i32), ("depth", i32), ("heads", i32), ("head_dim", i32), ("mlp", i32),
                ("act", i32), ("tokens", i32), ("patch_size", i32), ("image", i32), ("patch_k_pad", i32),
                ("has_ln_pre", i32), ("ln_eps", f32), ("ln_vision_eps", f32), ("patch", Linear),
                ("cls", vp), ("pos", vp), ("ln_pre_w", vp), ("ln_pre_b", vp), ("ln_vision_w", vp), ("ln_vision_b", vp),
                ("layers", C.POINTER(VitLayer)), ("fp8", i32), ("calib_amax", vp), ("pre_ln_out", vp), ("patch_x3", i32)]


class QfLayer(C.Structure):
    _fields_ = [("qkv", Linear), ("attn_out", Linear), ("attn_ln_w", vp), ("attn_ln_b", vp),
                ("has_cross", i32), ("cross_index", i32),
                ("cq", Linear), ("cross_out", Linear), ("cross_ln_w", vp), ("cross_ln_b", vp),
                ("ffn_t_in", Linear), ("ffn_t_out", Linear), ("ffn_t_ln_w", vp), ("ffn_t_ln_b", vp),
                ("ffn_q_in", Linear), ("ffn_q_out", Linear), ("ffn_q_ln_w", vp), ("ffn_q_ln_b", vp)]


class QformerModel(C.Structure):
    _fields_ = [("dtype", i32), ("hidden", i32), ("n_layers", i32), ("heads", i32), ("head_dim", i32), ("ffn", i32),
                ("num_query", i32), ("enc_width", i32), ("embed_dim", i32), ("max_txt", i32), ("n_cross", i32),
                ("vocab", i32), ("ln_eps", f32), ("word_emb", vp), ("pos_emb", vp), ("emb_ln_w", vp), ("emb_ln_b", vp),
                ("query_tokens", vp), ("ckv_all", Linear), ("vision_proj", Linear), ("text_proj", Linear),
                ("layers", C.POINTER(QfLayer)), ("x3", i32), ("x3_image", i32), ("x3_fuse", i32)]


# name -> (restype, argtypes); must list every symbol include/sprc.h declares
SIGNATURES = {
    "sprc_version": (i32, []),
    "sprc_last_error": (C.c_char_p, []),
    "sprc_prof_enable": (i32, [i32]),
    "sprc_prof_collect": (i32, [C.POINTER(ProfEntry)]),
    "sprc_stream_create_partition": (i32, [i32, i32, C.POINTER(vp)]),
    "sprc_stream_destroy": (i32, [vp]),
    "sprc_stream_cus": (i32, [vp]),
    "sprc_cast_f32_to_bf16": (i32, [vp, vp, sz, vp]),
    "sprc_cast_f32_to_16": (i32, [vp, vp, sz, i32, vp]),
    "sprc_cast_f32_to_x3": (i32, [vp, vp, i64, i32, vp]),
    "sprc_absmax_bf16": (i32, [vp, sz, vp, vp]),
    "sprc_absmax_16": (i32, [vp, sz, i32, vp, vp]),
    "sprc_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "sprc_gemm_pair": (i32, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), vp]),
    "sprc_layernorm": (i32, [C.POINTER(LayerNormArgs), vp]),
    "sprc_attention": (i32, [C.POINTER(AttentionArgs), vp]),
    "sprc_im2row": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "sprc_vit_assemble": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "sprc_qformer_embed": (i32, [C.POINTER(QformerEmbedArgs), vp]),
    "sprc_l2norm_rows": (i32, [vp, i64, vp, vp, i64, i32, i32, i32, vp]),
    "sprc_qformer_mask": (i32, [vp, vp, i32, i32, i32, vp]),
    "sprc_sim_max": (i32, [vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "sprc_topk": (i32, [vp, i64, vp, i32, i32, i32, i32, vp, vp, vp]),
    "sprc_rank_of": (i32, [vp, i64, vp, i32, i32, i32, vp, vp]),
    "sprc_vit_workspace_bytes": (sz, [C.POINTER(VitModel), i32]),
    "sprc_qformer_workspace_bytes": (sz, [C.POINTER(QformerModel), i32]),
    "sprc_vit_forward": (i32, [C.POINTER(VitModel), vp, i32, vp, vp, sz, vp]),
    "sprc_qformer_image": (i32, [C.POINTER(QformerModel), vp, i32, vp, vp, vp, sz, vp]),
    "sprc_qformer_fuse": (i32, [C.POINTER(QformerModel), vp, i32, vp, vp, i32, vp, vp, vp, sz, vp]),
    "sprc_qformer_fuse_kv": (i32, [C.POINTER(QformerModel), vp, i32, vp, vp, vp, i32, vp, vp, vp, sz, vp]),
    "sprc_qformer_kv_workspace_bytes": (sz, [C.POINTER(QformerModel), i32, i32]),
    "sprc_qformer_encode_kv": (i32, [C.POINTER(QformerModel), vp, i32, i32, vp, vp, sz, vp]),
    "sprc_qformer_itm_workspace_bytes": (sz, [C.POINTER(QformerModel), i32]),
    "sprc_qformer_itm": (i32, [C.POINTER(QformerModel), vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, i32, vp, vp, sz, vp]),
    "sprc_itm_head": (i32, [vp, i64, i32, i32, vp, vp, i32, vp, vp]),
    "sprc_qformer_fuse_train": (i32, [C.POINTER(QformerModel), vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, sz, vp]),
    "sprc_qformer_text_only": (i32, [C.POINTER(QformerModel), vp, vp, vp, i32, vp, vp, vp, sz, vp]),
    "sprc_qformer_text": (i32, [C.POINTER(QformerModel), vp, vp, i32, vp, vp, vp, sz, vp]),
    "sprc_contrastive_ce": (i32, [vp, i64, i32, f32, vp, vp]),
    "sprc_align_mse": (i32, [vp, i64, i32, i32, vp, i32, vp, vp]),
    "sprc_transpose_f32": (i32, [vp, i64, vp, i64, i32, i32, vp]),
    "sprc_transpose_f32_to16": (i32, [vp, i64, vp, i64, i32, i32, i32, vp]),
    "sprc_colsum_f32": (i32, [vp, i64, i32, i32, vp, i32, vp]),
    "sprc_gelu_fwd": (i32, [vp, vp, sz, vp]),
    "sprc_gelu_bwd": (i32, [vp, vp, vp, sz, vp]),
    "sprc_layernorm_bwd_workspace_bytes": (sz, [i32, i32]),
    "sprc_layernorm_bwd": (i32, [vp, i64, vp, vp, i64, f32, i32, i32, vp, i64, vp, vp, vp, sz, vp]),
    "sprc_attention_bwd": (i32, [C.POINTER(AttentionBwdArgs), vp]),
    "sprc_dropout_f32": (i32, [vp, vp, vp, sz, C.c_uint64, C.c_uint32, f32, vp]),
    "sprc_qformer_embed_rows": (i32, [C.POINTER(QformerEmbedArgs), vp, vp]),
    "sprc_qformer_embed_bwd": (i32, [C.POINTER(QformerEmbedArgs), vp, vp, i64, vp, vp, vp]),
    "sprc_sim_max_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "sprc_contrastive_ce_bwd": (i32, [vp, i64, i32, f32, f32, vp, vp, vp]),
    "sprc_l2norm_bwd": (i32, [vp, i64, vp, i64, vp, i64, i32, i32, vp]),
    "sprc_align_mse_bwd": (i32, [vp, i64, i32, i32, vp, i32, f32, vp, i64, vp]),
    "sprc_preprocess_workspace_bytes": (sz, [i32, i32, f32, i32]),
    "sprc_preprocess_targetpad": (i32, [vp, i32, i32, i64, f32, i32, C.POINTER(f32), C.POINTER(f32), vp, vp, sz, vp]),
}

_lib = None


class SprcError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen libsprc_hip.so and type every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SprcError(f"{LIB_PATH} is missing: the HIP extension is the only compute path "
                        f"(build it with `python -m sprc_amd.build`)")
    # torch first: it ships its own libamdhip64; were ours loaded before it, the process would hold two HIP runtimes
    # and our launches would see "no ROCm-capable device" (same SONAME: the loader then reuses torch's copy for us)
    import torch  # noqa: F401
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.sprc_version() != ABI_VERSION:
        raise SprcError(f"{LIB_PATH} speaks ABI {lib.sprc_version()}, this binding {ABI_VERSION}: rebuild (python -m sprc_amd.build)")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().sprc_last_error().decode(errors="replace")
        raise SprcError(f"{what or 'sprc call'} failed (rc={rc}): {msg}")
