/* sprc.h -- C ABI of the MI355X-native SPRC retrieval engine (libsprc_hip.so).
 *
 * The reference (chunmeifeng/SPRC) is pure Python and has no FFI; its "plugin API" is the
 * duck-typed model protocol consumed by its evaluation harness.
 * Every entry point below names the reference call site(s) whose computation it replaces
 * (paths relative to /root/reference/src).  The Python host (sprc_amd/model.py) mirrors the
 * reference protocol (`extract_target_features`, `inference`) and binds these symbols with
 * ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C, no torch types: raw device pointers, explicit sizes, a stream handle
 *     (hipStream_t passed as void*; the caller's current stream);
 *   - the caller owns every input, output and workspace buffer; the library never allocates
 *     or frees device memory and never synchronises the stream;
 *   - return 0 on success, a negative SPRC_E* code otherwise; sprc_last_error() returns a
 *     thread-local message; no C++ exception crosses the boundary;
 *   - all matrices are row-major; "ld*" are leading dimensions in ELEMENTS;
 *   - compute dtype: SPRC_BF16 or SPRC_F16 (16-bit MFMA operands, fp32 accumulate, fp32 residual
 *     stream / LayerNorm / softmax) or SPRC_F32 (exact fp32 MFMA, parity mode).
 */
#ifndef SPRC_H
#define SPRC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPRC_ABI_VERSION 6

enum { SPRC_OK = 0, SPRC_EINVAL = -1, SPRC_ELAUNCH = -2, SPRC_EWORKSPACE = -3, SPRC_EUNSUPPORTED = -4 };
enum { SPRC_F32 = 0, SPRC_BF16 = 1,
       SPRC_F16 = 2 /* IEEE half.  As a COMPUTE dtype (ABI 3): fp16 MFMA operands (v_mfma_f32_32x32x16_f16, the bf16 rate), fp32
                       accumulate / residual stream / LayerNorm / softmax -- the reference's own GPU precision (fp16 autocast,
                       lavis/models/blip2_models/blip2.py:36-44, eva_vit.py:410-425), 8x finer operand rounding than bf16.
                       Also an OUTPUT dtype of a bf16 sprc_gemm: a residual-branch output ("delta") that sprc_layernorm
                       adds to the fp32 residual stream */,
       SPRC_F16X3 = 4 /* STORAGE layout of split-precision activations (OUTPUT dtype of sprc_gemm / sprc_layernorm / sprc_qformer_embed /
                         sprc_cast_f32_to_x3 / sprc_im2row / sprc_attention.out_x3): a row of a logical [M, K] matrix x takes 4K bytes,
                             [ K x fp16: hi = fp16(x) | K x e4m3: (x - hi) * 2^12 | K x e4m3: x ]
                         (ABI 4; ABI 3 stored three fp16 segments [hi | lo | hi]).  A split product -- sprc_gemm with dtype SPRC_F16,
                         K, and k8 = 2K -- multiplies it with weight rows [ K x fp16: W_hi | K x e4m3: W * 2^6 | K x e4m3: (W - W_hi) * 2^18 ]:
                             x_hi.W_hi  +  2^-18 ( [x_lo 2^12].[W 2^6] + [x].[W_lo 2^18] )   =   x.W  up to the e4m3 rounding of the two
                         CORRECTION terms, which are 2^-11 of the result and need 3-4 significant bits: the product to ~2^-16 instead of
                         2^-11 (measured: relative rms error 1.0e-5 against 2.9e-4 for plain fp16 and 9e-7 for three fp16 segments),
                         at 2 units of matrix time instead of 3 -- the e4m3 K-tiles run on the MX-scaled MFMA at twice the fp16 rate
                         (block scales 2^-9 x 2^-9).  This is what lets the Q-Former of the fp16 engine follow the reference's GPU
                         path, whose Q-Former runs in fp32 OUTSIDE the fp16 autocast (blip2_qformer_cir_align_prompt.py:366-368).
                         A consumer that is not split reads the hi segment alone (K fp16, leading dimension 2K). */,
       SPRC_FP8 = 3 /* OCP e4m3fn (the gfx950 fp8; NOT MI300's fnuz): GEMM operands with a per-tensor activation scale and
                       per-output-channel weight scales, fp32 accumulation (BASELINE.json config C5: "ViT-L, fp8 MFMA") */ };
enum { SPRC_ACT_NONE = 0, SPRC_ACT_GELU = 1, SPRC_ACT_QUICKGELU = 2 };

typedef void* sprc_stream;                    /* hipStream_t */

/* Row indirection shared by GEMM / LayerNorm: logical row m lives at physical row
 *   (m / rows_per_group) * group_stride + (m % rows_per_group) + group_offset
 * (rows_per_group == 0 -> identity).  Lets the Q-Former address "rows [:32]" / "rows [32:]" of a
 * [B,64,768] tensor (Qformer.py:436,455-468) without copies. */
typedef struct { int32_t rows_per_group, group_stride, group_offset; } sprc_rowmap;

/* ------------------------------------------------------------------------------------------
 * Building-block operators (also what the unit parity tests drive)
 * ---------------------------------------------------------------------------------------- */

int         sprc_version(void);
const char* sprc_last_error(void);

/* Per-kernel-class timing with HIP events recorded on the launch stream around every kernel launch
 * (what bench.py's `roofline` leg reads).  enable(1) clears the records; collect() synchronises the
 * recorded events and sums elapsed time, ALGORITHMIC flops and bytes per class. */
enum { SPRC_K_GEMM_BF16 = 0, SPRC_K_GEMM_F32 = 1, SPRC_K_ATTN = 2, SPRC_K_ROWOPS = 3, SPRC_K_RANK = 4, SPRC_K_COUNT = 5 };
typedef struct {
    double ms, flops, bytes;   /* sum of launch durations; ALGORITHMIC flops and bytes (sprc_gemm: 2 M N k_alg, see sprc_gemm_args) */
    int64_t launches;
    double busy_ms;            /* union of the launches' [start,end] intervals: == ms unless launches of the class overlapped */
    double exec_flops;         /* ABI 4: flops the launches EXECUTED (sprc_gemm: 2 M N K with the launched K: split-precision products
                                * reduce over K = 3 k_alg, the patch embedding over its zero-padded K); >= flops */
} sprc_prof_entry;
int sprc_prof_enable(int on);   /* 1 = start afresh, 0 = pause (records are kept for collect), 2 = resume.  Every recorded
                                 * launch costs two stream markers (~7 us of pipeline bubble): sample steps, do not record all */
int sprc_prof_collect(sprc_prof_entry* out /* [SPRC_K_COUNT] */);

/* CU-PARTITION streams (ABI 6).  The reference runs its whole forward on one CUDA stream (blip_validate.py / utils.py:46-77: a plain loop
 * over loader batches); on MI355X every kernel of that loop is either matrix-bound (the GEMMs' K loops, HBM idle) or memory-bound (GEMM
 * epilogues, LayerNorm, attention: matrix pipes idle), and a kernel that owns all 256 CUs keeps them in lockstep -- the whole chip
 * alternates between the two.  sprc_stream_create_partition(part, nparts) returns a HIP stream whose queue is restricted
 * (hipExtStreamCreateWithCUMask) to CUs {c : (c / 8) % nparts == part} of every XCD (mask bit i = CU i/8 of XCD i%8, the KFD's layout), so
 * that `nparts` independent batches run SIDE BY SIDE on disjoint CUs and one partition's memory bursts fall under the others' K loops.
 * The library sizes persistent grids and its tile cost model by the CU count registered for the stream a launch goes to
 * (sprc_stream_cus; the device's CU count for any other stream).  Caller-owned: destroy with sprc_stream_destroy. */
int sprc_stream_create_partition(int32_t part, int32_t nparts, sprc_stream* out);
int sprc_stream_destroy(sprc_stream s);
int sprc_stream_cus(sprc_stream s);   /* CUs a launch on `s` can use: the partition's count, or the device's */

/* amax[0] = max(amax[0], max |x|) over n bf16 values (device float, caller-initialised): fp8 scale calibration. */
int sprc_absmax_bf16(const void* x, size_t n, float* amax, sprc_stream s);
/* the same over n 16-bit values of `dtype` (SPRC_BF16 or SPRC_F16) */
int sprc_absmax_16(const void* x, size_t n, int32_t dtype, float* amax, sprc_stream s);

/* fp32 -> bf16 (round-to-nearest-even) weight/feature packing. */
int sprc_cast_f32_to_bf16(const float* src, uint16_t* dst, size_t n, sprc_stream s);
/* fp32 -> `dtype` (SPRC_BF16 or SPRC_F16), round-to-nearest-even. */
int sprc_cast_f32_to_16(const float* src, void* dst, size_t n, int32_t dtype, sprc_stream s);
/* fp32 [rows, cols] (contiguous) -> SPRC_F16X3 rows (4 cols bytes each, contiguous); cols % 4 == 0. */
int sprc_cast_f32_to_x3(const float* src, void* dst, int64_t rows, int32_t cols, sprc_stream s);

/* C = epilogue(A[M,K] . W[N,K]^T + bias) -- replaces every nn.Linear / F.linear on the path:
 * eva_vit.py:123,146,55-60; clip_vit.py:132-139; Qformer.py:135-137,201-211,291-293,365,377;
 * align_prompt.py:348,385.  A and W are `dtype`; bias/resid fp32; out is `out_dtype`.
 * K % 64 == 0 (bf16, fp16) / K % 32 == 0 (f32); lda, ldw multiples of 8 (16-bit) / 4 (f32) elements.
 * dtype SPRC_F16: outputs fp16, f32 or SPRC_F16X3 (ldc >= 2 N in fp16 units, N % 4 == 0; no residual, no max32), every activation.
 * out_dtype SPRC_F16 with bf16 operands: no activation / residual / max32 (see sprc_layernorm_args.add16).
 * dtype SPRC_FP8: K % 128 == 0; outputs bf16 / fp16 / f32 (plain epilogue, residual allowed) or fp8 (any activation).
 *   out = act(A.W^T + bias) + resid                         (resid optional, fp32, mapped like C)
 * max32 != 0: "similarity" epilogue -- rows of A are query vectors, rows of W are gallery tokens (32 per
 * image); out[m*ldc + n/32] = max over the 32 W-rows of image n/32 (align_prompt.py:353-358). */
typedef struct {
    int32_t M, N, K;
    int32_t dtype, out_dtype, act, max32;
    const void* A;  int64_t lda;  sprc_rowmap amap;
    const void* W;  int64_t ldw;
    const float* bias;
    const float* resid; int64_t ldr;
    void* C;        int64_t ldc;  sprc_rowmap cmap;
    /* optional device scratch the call may use for split-K partial sums (never read afterwards); NULL/0 = none.
     * 8 * 128 * N * 4 bytes lets the <= 128 remainder rows of a K >= 4096 product be reduced by 8 workgroups per tile. */
    void* scratch;  size_t scratch_bytes;
    /* dtype == SPRC_FP8:  out = act(a_scale * w_scale[n] * (A_q . W_q^T) + bias) + resid, A_q / W_q e4m3fn with
     * A = a_scale * A_q (per tensor) and W[n,:] = w_scale[n] * W_q[n,:] (per output channel; fp32 [N], 16-byte aligned).
     * out_dtype == SPRC_FP8: the result is multiplied by out_scale (= 1 / the consumer's a_scale) and saturated to +-448. */
    const float* w_scale; float a_scale; float out_scale;
    /* ABI 4: the LOGICAL reduction length this product stands for, for the profiler's algorithmic flop count only (0 = K).
     * The patch embedding launches its zero-padded K (640 for 588); a split-precision product (k8 below) counts its fp16 K as the
     * algorithmic one and its e4m3 correction elements as executed work only.  Never changes what is computed. */
    int32_t k_alg;
    /* ABI 4, dtype SPRC_F16 only: split-precision product.  Every row of A and W holds K fp16 elements FOLLOWED by k8 e4m3fn elements
     * (k8 == 2K: the SPRC_F16X3 layout above; lda / ldw in fp16 units >= K + k8 / 2); out = epilogue(sum over the fp16 part +
     * 2^-18 * sum over the e4m3 part).  K % 128 == 0, k8 % 128 == 0, k8 >= 512.  0 = a plain product.  Epilogues: bias, GELU,
     * fp32 residual; outputs SPRC_F16 / SPRC_F32 / SPRC_F16X3; sprc_gemm_pair allowed; no max32. */
    int32_t k8;
} sprc_gemm_args;
int sprc_gemm(const sprc_gemm_args* a, sprc_stream s);

/* Two products of identical shape in ONE launch: same A, C, resid, leading dimensions, dtypes, activation and row-map
 * geometry; `b` may differ from `a` in W, bias and the two row-map group offsets only.  This is the query / text FFN pair
 * of a Q-Former layer (Qformer.py:455-475: rows [:32] of every sample through intermediate_query / output_query, rows
 * [32:] through intermediate / output): half-empty grids of the two small products become one full one. */
int sprc_gemm_pair(const sprc_gemm_args* a, const sprc_gemm_args* b, sprc_stream s);

/* y = LayerNorm(x) over the last dim (fp32 statistics, two-pass) -- nn.LayerNorm at
 * eva_vit.py:175-176 (eps 1e-6), clip_vit.py:100-106, blip2.py:193-199 (ln_vision, eps 1e-5),
 * Qformer.py:112,294,380 (eps 1e-12).  Writes an fp32 copy (residual stream) and/or a
 * compute-dtype copy (next GEMM operand); either may be NULL. */
typedef struct {
    int32_t M, D, out_dtype;
    const float* x;  int64_t ldx;  sprc_rowmap xmap;
    const float* gamma; const float* beta; float eps;
    float* y32;      int64_t ld32; sprc_rowmap ymap;
    void*  y16;      int64_t ld16;              /* rows mapped with ymap as well; out_dtype SPRC_F16X3: ld16 >= 2 D (fp16 units; a split row is 4 D bytes) */
    /* Optional fused residual add (bf16 engine): the normalised row is x + add16, with add16 the fp16 output of the branch
     * GEMM (attention proj / MLP fc2 / BERT output.dense: eva_vit.py:178-179, clip_vit.py:137-138, Qformer.py:294,380).
     * The fp32 + residual GEMM epilogue it replaces was an un-overlapped HBM burst (8 B per element with the matrix pipe
     * idle); here the add rides on a pass that reads x anyway.  add16 rows are mapped like x (xmap); it may alias y16
     * (a wave holds its whole row in registers before it stores).  sum32 (optional, rows mapped like x, may alias x)
     * receives x + add16: the pre-LN residual-stream update. */
    const void* add16; int64_t ld_add;
    float* sum32;      int64_t ld_sum;
    /* out_dtype == SPRC_FP8: y16 receives sat(LN(x) * y16_scale) as e4m3fn (ld16 in elements = bytes); y16_scale = 1 / the
     * consumer GEMM's a_scale.  Not combinable with add16. */
    float y16_scale;
} sprc_layernorm_args;
int sprc_layernorm(const sprc_layernorm_args* a, sprc_stream s);

/* out = softmax(scale * Q K^T + key_mask) V per (batch, head) -- eva_vit.py:128-145,
 * nn.MultiheadAttention in clip_vit.py:132-134, Qformer.py:233-268.  Q/K/V/out are `dtype`
 * (bf16 or f32); token t of batch b, head h, lives at  ptr + (b*T + t)*ld + h*head_dim.
 * key_mask: optional additive fp32 [B,Tk] (Qformer.py:806-807: (1-m)*-10000). */
typedef struct {
    int32_t B, H, Tq, Tk, head_dim, dtype;
    const void* q; int64_t ldq;
    const void* k; int64_t ldk;
    const void* v; int64_t ldv;
    void* out;     int64_t ldo;
    const float* key_mask;
    float scale;
    /* Optional second key/value segment: the key axis is cat(segment 1: Tk tokens of (k, v) at batch row kv_index[b],
     * segment 2: Tk2 tokens of (k2, v2) at batch row kv2_index[b]); index arrays are device int32 [B], NULL = b.
     * This is the stage-2 rerank's cross-attention over cat(reference, candidate) image tokens
     * (blip2_qformer_cir_rerank.py:430-437) without materialising the concatenation per (query, candidate) pair.
     * Not combinable with key_mask. */
    const void* k2; int64_t ldk2;
    const void* v2; int64_t ldv2;
    int32_t Tk2;
    const int32_t* kv_index; const int32_t* kv2_index;
    /* Training (dtype SPRC_F32 only): dropout on the attention probabilities (Qformer.py:264), out = (D o softmax(..)) V with
     * D[b, h, i, j] = sprc_dropout keep mask at element index ((b H + h) Tq + i) Tk + j of site `drop_site`, scaled 1 / (1 - drop_p).
     * drop_p == 0: none (inference). */
    float drop_p; uint32_t drop_site; uint64_t drop_seed;
    int32_t out_x3;   /* dtype SPRC_F16 only: `out` rows are stored in the SPRC_F16X3 layout (logical width H * head_dim, ldo >= 2 H head_dim) */
} sprc_attention_args;
int sprc_attention(const sprc_attention_args* a, sprc_stream s);

/* Patch extraction for the 14x14/stride-14 conv (eva_vit.py:196,203; clip_vit.py:160,173-175):
 * images [B,3,S,S] fp32 -> rows [B*G*G, k_pad] (`dtype`), column = c*P*P + i*P + j, zero padded.
 * dtype SPRC_F16X3: split rows of logical width k_pad (4 k_pad bytes each). */
int sprc_im2row(const float* images, void* rows, int32_t B, int32_t image, int32_t patch, int32_t k_pad,
                int32_t dtype, sprc_stream s);

/* x[b,0,:] = cls + pos[0];  x[b,1+p,:] = patch_out[b,p,:] + pos[1+p]
 * (eva_vit.py:328-331; clip_vit.py:176-177).  fp32. */
int sprc_vit_assemble(const float* patch_out, const float* cls, const float* pos, float* x,
                      int32_t B, int32_t tokens, int32_t width, sprc_stream s);

/* Q-Former embeddings (Qformer.py:98-114): rows [0,Lq) = query_embeds (batch stride q_bstride, 0 =
 * broadcast), rows [Lq,Lq+Lt) = word_emb[ids] + pos_emb[0..Lt); one LayerNorm over all rows.
 * input_ids may be NULL (Lt = 0).  Writes fp32 and compute-dtype copies. */
typedef struct {
    int32_t B, Lq, Lt, hidden, out_dtype, vocab;
    const float* query_embeds; int64_t q_bstride;
    const int64_t* input_ids;
    const float* word_emb; const float* pos_emb;
    const float* gamma; const float* beta; float eps;
    float* y32; void* y16;                      /* [B, Lq+Lt, hidden] contiguous (split rows of 4 hidden bytes for out_dtype SPRC_F16X3) */
    int32_t no_img;                             /* 1: the text-only form of Qformer.py:88-104 (training, align_prompt.py:173-179): rows =
                                                 * [text[0] ; the Lq query rows ; text[1:]] and every row gets its absolute position */
} sprc_qformer_embed_args;
int sprc_qformer_embed(const sprc_qformer_embed_args* a, sprc_stream s);

/* y = x / max(||x||_2, 1e-12) per row (F.normalize, align_prompt.py:348-350,385). */
int sprc_l2norm_rows(const float* x, int64_t ldx, float* y32, void* y16, int64_t ldy, int32_t M, int32_t D,
                     int32_t out_dtype, sprc_stream s);

/* additive key mask (Qformer.py:806-807): out[b, j] = j < Lq ? 0 : (1 - mask[b, j-Lq]) * -10000 */
int sprc_qformer_mask(const int64_t* attention_mask, float* out, int32_t B, int32_t Lq, int32_t Lt, sprc_stream s);

/* ------------------------------------------------------------------------------------------
 * Ranking (R7): validate_blip.py:253-254, :44-45; cirr_test_submission.py:82-83
 * ---------------------------------------------------------------------------------------- */

/* sim[nq,N] = max_j <fusion[q,:], feats[n,j,:]>  (align_prompt.py:353-358) -- thin wrapper over
 * sprc_gemm(max32).  fusion [nq,E], feats [N,J=32,E], both `dtype`. */
int sprc_sim_max(const void* fusion, const void* feats, float* sim, int64_t ld_sim,
                 int32_t nq, int32_t N, int32_t E, int32_t dtype, sprc_stream s);

/* Per row of sim[nq,N] (ld in elements): the k smallest keys (fl32(1 - sim[n]), gidx[n]) ascending,
 * i.e. the first k entries of a STABLE argsort of the fp32 distance (the ranking contract,
 * SURVEY.md section 7).  gidx: optional int32 [nq,N] global indices (NULL -> n + idx_base).
 * out_sim [nq,k] fp32, out_idx [nq,k] int32; unused slots (N < k) get sim = -inf, idx = -1.
 * 1 <= k <= 64. */
int sprc_topk(const float* sim, int64_t ld, const int32_t* gidx, int32_t idx_base,
              int32_t nq, int32_t N, int32_t k, float* out_sim, int32_t* out_idx, sprc_stream s);

/* rank[q,l] = #{ n : (d[q,n], n) < (d[q,t], t) },  t = listed[q,l]  (-1 where t < 0): the position
 * of gallery item t in the stable order, without sorting (exact Recall@K for any K). */
int sprc_rank_of(const float* sim, int64_t ld, const int32_t* listed, int32_t nq, int32_t N, int32_t L,
                 int32_t* rank, sprc_stream s);

/* ------------------------------------------------------------------------------------------
 * Composite forward passes (one call per batch; the kernels above, sequenced on `s`)
 * ---------------------------------------------------------------------------------------- */

enum { SPRC_X3_QKV = 1, SPRC_X3_ATTN_OUT = 2, SPRC_X3_CROSS_Q = 4, SPRC_X3_CROSS_OUT = 8, SPRC_X3_FFN_IN = 16, SPRC_X3_FFN_OUT = 32,
       SPRC_X3_CKV = 64, SPRC_X3_HEADS = 128, SPRC_X3_ALL = 255 };      /* layer kinds of sprc_qformer_model.x3 */
enum { SPRC_FP8_ALL = 1, SPRC_FP8_MLP = 2 };                          /* sprc_vit_model.fp8 */
typedef struct { const void* w; const float* b; } sprc_linear;   /* w: [out,in(padded)] compute dtype */

typedef struct {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    sprc_linear qkv, proj, fc1, fc2;            /* qkv bias: [q_bias, 0, v_bias] (eva_vit.py:120-122) */
    /* fp8 ViT (sprc_vit_model.fp8 != 0): qkv / fc1 / fc2 weights are e4m3fn with per-output-channel scales, their inputs are
     * quantised with the static per-tensor scales below (a = s * a_q); proj stays 16-bit (its input is the attention output);
     * with fp8 == SPRC_FP8_MLP qkv is a 16-bit linear like proj */
    const float *qkv_ws, *fc1_ws, *fc2_ws;      /* [3*width], [mlp], [width] fp32 */
    float s_ln1, s_ln2, s_mlp;                  /* activation scales of the qkv / fc1 / fc2 inputs */
} sprc_vit_layer;

typedef struct {
    int32_t dtype, width, depth, heads, head_dim, mlp, act, tokens, patch_size, image, patch_k_pad, has_ln_pre;
    float ln_eps, ln_vision_eps;
    sprc_linear patch;                          /* [width, patch_k_pad]; bias may be NULL (clip) */
    const float *cls, *pos;                     /* [width], [tokens,width] */
    const float *ln_pre_w, *ln_pre_b, *ln_vision_w, *ln_vision_b;
    const sprc_vit_layer* layers;               /* host array [depth] */
    int32_t fp8;                                /* dtype SPRC_BF16 or SPRC_F16 (everything else of the model computes in it) and
                                                 * SPRC_FP8_ALL (1): qkv, fc1 and fc2 of every block run on fp8 operands;
                                                 * SPRC_FP8_MLP (2): fc1 and fc2 only */
    float* calib_amax;                          /* optional device array [depth*3] (16-bit model, fp8 == 0): running max |x| of the qkv / fc1 /
                                                 * fc2 inputs, updated by sprc_vit_forward -- the calibration pass of the fp8 scales */
    float* pre_ln_out;                          /* optional device array [B, tokens, width] fp32: receives the INPUT of ln_vision (the ViT's last
                                                 * residual stream) -- what the training step needs for ln_vision's gradient (blip2.py:81) */
    int32_t patch_x3;                           /* dtype SPRC_F16 only, 1: the patch embedding runs on split-precision operands -- patch.w holds
                                                 * split weight rows [W_hi | W 2^6 | W_lo 2^18] (4 patch_k_pad bytes each), the patch rows are SPRC_F16X3.  Its output IS the
                                                 * residual stream's first value: an fp16 rounding there is carried through every block
                                                 * (tools/fq_vit.py: 27 % of the fp16 ViT's error variance, for 0.1 ms) */
} sprc_vit_model;

typedef struct {
    sprc_linear qkv, attn_out;        const float *attn_ln_w, *attn_ln_b;
    int32_t has_cross, cross_index;   /* cross_index: position of this layer's K|V block in ckv_all */
    sprc_linear cq, cross_out;        const float *cross_ln_w, *cross_ln_b;
    sprc_linear ffn_t_in, ffn_t_out;  const float *ffn_t_ln_w, *ffn_t_ln_b;
    sprc_linear ffn_q_in, ffn_q_out;  const float *ffn_q_ln_w, *ffn_q_ln_b;
} sprc_qf_layer;

typedef struct {
    int32_t dtype, hidden, n_layers, heads, head_dim, ffn, num_query, enc_width, embed_dim, max_txt, n_cross, vocab;
    float ln_eps;
    const float *word_emb, *pos_emb, *emb_ln_w, *emb_ln_b;
    const float* query_tokens;                  /* [num_query, hidden] */
    sprc_linear ckv_all;                        /* [n_cross*2*hidden, enc_width]: K|V of every cross layer */
    sprc_linear vision_proj, text_proj;         /* [embed_dim, hidden] */
    const sprc_qf_layer* layers;                /* host array [n_layers] */
    int32_t x3;                                 /* != 0 (dtype SPRC_F16 only): split-precision Q-Former -- every GEMM input activation is
                                                 * kept in the SPRC_F16X3 layout, and the weight matrices of the layer KINDS whose SPRC_X3_*
                                                 * bit is set are packed as split weight rows [W_hi fp16 | W 2^6 e4m3 | W_lo 2^18 e4m3], 4 `in` bytes each
                                                 * (the others stay [out, in] fp16 and multiply the hi segment only); attention operands (q, k, v, probabilities) stay plain fp16 */
    int32_t x3_image, x3_fuse;                  /* subsets of x3: the kinds that run the split product (k8 = 2 K) in sprc_qformer_image /
                                                 * in the query-side calls (fuse, text_only, encode_kv, itm); the others read hi only */
} sprc_qformer_model;

/* Workspace sizes in bytes for a batch of B (images or queries). */
size_t sprc_vit_workspace_bytes(const sprc_vit_model* m, int32_t B);
size_t sprc_qformer_workspace_bytes(const sprc_qformer_model* m, int32_t B);

/* raw[B,tokens,width] (fp32) = ln_vision(ViT(images[B,3,S,S]))   -- align_prompt.py:366-368,
 * eva_vit.py:324-340 / clip_vit.py:171-185, blip2.py:193-199.
 * Environment SPRC_VIT_STREAMS=2 (read at the first call): the two halves of a batch of >= 16 run the transformer blocks
 * on `s` and on one library-owned helper stream, forked after and joined before the rest of the work on `s`. */
int sprc_vit_forward(const sprc_vit_model* m, const float* images, int32_t B, float* raw,
                     void* ws, size_t ws_bytes, sprc_stream s);

/* feats[B,32,E] = normalize(vision_proj(Qformer(query_tokens, enc = raw)))  -- align_prompt.py:369-385;
 * Q-Former call shape (i).  feats16 (optional) receives the compute-dtype copy for bf16 ranking. */
int sprc_qformer_image(const sprc_qformer_model* m, const float* raw, int32_t B, float* feats, void* feats16,
                       void* ws, size_t ws_bytes, sprc_stream s);

/* fusion[B,E] = normalize(text_proj(pass2[:,32,:]))  -- align_prompt.py:313-350: Q-Former pass 1
 * (ids + reference image, call shape (ii)) then pass 2 (call shape (iii)).
 * ref_embeds [B,tokens,enc_width] fp32, input_ids/attention_mask [B,max_txt] int64. */
int sprc_qformer_fuse(const sprc_qformer_model* m, const float* ref_embeds, int32_t enc_tokens,
                      const int64_t* input_ids, const int64_t* attention_mask, int32_t B,
                      float* fusion, void* fusion16, void* ws, size_t ws_bytes, sprc_stream s);

/* sprc_qformer_fuse with the K|V projections of the reference images GIVEN: kv = rows of an sprc_qformer_encode_kv output
 * ([n, enc_tokens, n_cross*2*hidden], compute dtype), query b reads row kv_index[b] (device int32 [B]; NULL = b).  A gallery
 * image that serves as the reference of several queries (CIRR: the reference image of a query is a gallery image,
 * validate_blip.py:377,395-399) is then projected once instead of once per query -- 6.7 of a query's 29.3 GFLOP
 * (Qformer.py:191-193 inside align_prompt.py:332-339).  Same bits as sprc_qformer_fuse on the same images.  16-bit models.
 * OPTIONAL: bench.py does not use it (BASELINE.md's per-query figure counts the projection). */
int sprc_qformer_fuse_kv(const sprc_qformer_model* m, const void* kv, int32_t enc_tokens, const int32_t* kv_index,
                         const int64_t* input_ids, const int64_t* attention_mask, int32_t B, float* fusion,
                         void* fusion16, void* ws, size_t ws_bytes, sprc_stream s);

/* ------------------------------------------------------------------------------------------
 * Stage-2 rerank: Blip2QformerCirRerank.inference_rerank,
 * lavis/models/blip2_models/blip2_qformer_cir_rerank.py:399-445; caller cirr_test_submission.py:88-112.
 * The reference runs the Q-Former once per (query, candidate) pair over cat(reference, candidate) image tokens and
 * projects those 514 tokens to K|V inside every pair.  K|V projections are per token, so they are computed ONCE per
 * image (sprc_qformer_encode_kv) and a pair only names its two rows (index_a / index_b): 13 of the 19 GFLOP of a
 * pair disappear.
 * ---------------------------------------------------------------------------------------- */

/* kv[B*tokens, n_cross*2*hidden] (compute dtype) = K|V projections of image tokens raw[B,tokens,enc_width] (fp32) for
 * every cross-attention layer (Qformer.py:191-193).  Workspace: the bf16 copy of raw (bf16 engine only). */
size_t sprc_qformer_kv_workspace_bytes(const sprc_qformer_model* m, int32_t B, int32_t tokens);
int sprc_qformer_encode_kv(const sprc_qformer_model* m, const float* raw, int32_t B, int32_t tokens, void* kv,
                           void* ws, size_t ws_bytes, sprc_stream s);

/* prob[p] = softmax(mean over the 32 query rows of itm_head(Qformer(query_tokens, text_p, enc = cat(A[index_a[p]],
 * B[index_b[p]]))))[1] for P (query, candidate) pairs.  kv_a [*, tokens_a, n_cross*2*hidden] / kv_b likewise come from
 * sprc_qformer_encode_kv; index_a / index_b: device int32 [P] (NULL = p); input_ids / attention_mask [P, max_txt] int64
 * (the query's caption repeated for each of its candidates); itm_w [2, hidden], itm_b [2] fp32. */
size_t sprc_qformer_itm_workspace_bytes(const sprc_qformer_model* m, int32_t P);
int sprc_qformer_itm(const sprc_qformer_model* m, const float* itm_w, const float* itm_b,
                     const void* kv_a, int32_t tokens_a, const int32_t* index_a,
                     const void* kv_b, int32_t tokens_b, const int32_t* index_b,
                     const int64_t* input_ids, const int64_t* attention_mask, int32_t P,
                     float* prob, void* ws, size_t ws_bytes, sprc_stream s);

/* building block: prob[p] = softmax(W . mean_j h[p, j, :] + b)[1], j < Lq; h fp32 with `sample_stride` elements between samples */
int sprc_itm_head(const float* h, int64_t sample_stride, int32_t Lq, int32_t D, const float* w, const float* b, int32_t P,
                  float* prob, sprc_stream s);

/* ------------------------------------------------------------------------------------------
 * Image preprocessing: the reference's targetpad_transform(target_ratio, dim)
 * = TargetPad -> Resize(dim, BICUBIC) -> CenterCrop(dim) -> ToTensor -> Normalize, src/data_utils.py:49-72, :91-105.
 * src: ONE decoded image on the device, uint8 RGB, HWC, `src_stride` bytes per row.  out: fp32 [3, dim, dim].
 * Bit-exact with the reference transform (PIL's 8-bit two-pass bicubic resampler): tap tables are computed on the host
 * in double, the kernels do the 22-bit fixed-point arithmetic.  mean / std: host pointers to 3 floats.
 * ---------------------------------------------------------------------------------------- */
size_t sprc_preprocess_workspace_bytes(int32_t src_h, int32_t src_w, float target_ratio, int32_t dim);
int sprc_preprocess_targetpad(const uint8_t* src, int32_t src_h, int32_t src_w, int64_t src_stride, float target_ratio,
                              int32_t dim, const float* mean, const float* std, float* out,
                              void* ws, size_t ws_bytes, sprc_stream s);

/* ------------------------------------------------------------------------------------------
 * Training forward: the three losses of Blip2QformerCirAlignPrompt.forward,
 * lavis/models/blip2_models/blip2_qformer_cir_align_prompt.py:95-200 (eval semantics: dropout = identity).
 * The backward kernels are at the end of this file; sprc_amd/train.py sequences forward + backward, sprc_amd.model.forward
 * wraps them in a torch.autograd.Function.
 * ---------------------------------------------------------------------------------------- */

/* sprc_qformer_fuse + loss_align: additionally writes loss_align[0] = mse(mean over the query rows of the PASS-1 output,
 * mean over the rows of prompt_tokens [num_query, hidden])  (:192-193). */
int sprc_qformer_fuse_train(const sprc_qformer_model* m, const float* ref_embeds, int32_t enc_tokens,
                            const int64_t* input_ids, const int64_t* attention_mask, int32_t B,
                            float* fusion, void* fusion16, const float* prompt_tokens, float* loss_align,
                            void* ws, size_t ws_bytes, sprc_stream s);

/* feat[B,E] = normalize(text_proj(Qformer(text, query_embeds = prompt_tokens, no_img = True)[:, 0, :]))  (:170-179). */
int sprc_qformer_text_only(const sprc_qformer_model* m, const float* prompt_tokens, const int64_t* input_ids,
                           const int64_t* attention_mask, int32_t B, float* feat, void* feat16,
                           void* ws, size_t ws_bytes, sprc_stream s);

/* feat[B,E] = normalize(text_proj(Qformer(text)[:, 0, :])): the Q-Former as a plain text encoder -- no query rows, no image, text FFN
 * on every row (Qformer.py:98-114 with query_embeds None, :469-475).  This is the STAGE-1 score of the rerank model class,
 * Blip2QformerCirRerank.inference (lavis/models/blip2_models/blip2_qformer_cir_rerank.py:373-397), which ignores the reference image. */
int sprc_qformer_text(const sprc_qformer_model* m, const int64_t* input_ids, const int64_t* attention_mask, int32_t B,
                      float* feat, void* feat16, void* ws, size_t ws_bytes, sprc_stream s);

/* loss[0] = F.cross_entropy(sim[B,B] / temp, arange(B))  (:157-167, :181-190);  sim fp32, ld in elements. */
int sprc_contrastive_ce(const float* sim, int64_t ld, int32_t B, float temp, float* loss, sprc_stream s);

/* loss[0] = mse(mean_j h[b, j, :], mean_j prompt[j, :]), j < Lq; h fp32 with `sample_stride` elements between samples. */
int sprc_align_mse(const float* h, int64_t sample_stride, int32_t Lq, int32_t D, const float* prompt, int32_t B, float* loss,
                   sprc_stream s);

/* ------------------------------------------------------------------------------------------
 * Training BACKWARD: the kernels sprc_amd/train.py sequences into the gradient of
 * Blip2QformerCirAlignPrompt.forward (align_prompt.py:95-200) for blip_fine_tune_2.py:293-304.  The ViT is frozen in the
 * reference (align_prompt.py:64-69): what trains is the Q-Former, ln_vision, the ITC heads, query / prompt tokens and temp.
 * Products run on sprc_gemm through transposed operand copies: dX = dY . W as sprc_gemm(A = dY, W = W^T), dW (+)= dY^T . X as
 * sprc_gemm(A = dY^T, W = X^T, resid = dW) -- on the exact-fp32 MFMA (the parity mode), or (ABI 5) on fp16 OPERAND COPIES with fp32
 * accumulation, fp32 outputs and fp32 master weights: the reference's training arithmetic (fp16 autocast + GradScaler,
 * blip_fine_tune_2.py:290-303).  Everything else (LayerNorm, GELU, softmax, losses, every gradient buffer) is fp32.
 * ---------------------------------------------------------------------------------------- */

/* dst[c, r] = src[r, c]; leading dimensions in elements. */
int sprc_transpose_f32(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int32_t rows, int32_t cols, sprc_stream s);
/* ABI 5: dst[c, r] = (dtype) src[r, c], dtype SPRC_F16 | SPRC_BF16: the transposed 16-bit operand copy of the 16-bit training products
 * (replaces the x.t() / grad.t() views torch's autograd hands its fp16 matmuls under autocast; Qformer.py's nn.Linear backward). */
int sprc_transpose_f32_to16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int32_t rows, int32_t cols, int32_t dtype, sprc_stream s);
/* out[n] (+)= sum_m x[m, n]  (bias gradients; fixed summation order). */
int sprc_colsum_f32(const float* x, int64_t ld, int32_t M, int32_t N, float* out, int32_t accumulate, sprc_stream s);
/* y = x Phi(x) (exact erf form, Qformer.py:482-490 ACT2FN["gelu"]) and dx = dy (Phi(x) + x phi(x)). */
int sprc_gelu_fwd(const float* x, float* y, size_t n, sprc_stream s);
int sprc_gelu_bwd(const float* x, const float* dy, float* dx, size_t n, sprc_stream s);
/* LayerNorm backward from the INPUT x: dx (optional) = rstd (g - mean g - xhat mean(g xhat)), g = dy gamma; dgamma += sum dy xhat,
 * dbeta += sum dy (per-block partials in ws, reduced in block order). */
size_t sprc_layernorm_bwd_workspace_bytes(int32_t M, int32_t D);
int sprc_layernorm_bwd(const float* x, int64_t ldx, const float* gamma, const float* dy, int64_t lddy, float eps, int32_t M, int32_t D,
                       float* dx, int64_t lddx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, sprc_stream s);
/* Backward of sprc_attention (fp32, head_dim 64, same token layout): dq, dk, dv from dout; probabilities are recomputed.
 * scratch: 2 * B * H * Tq * Tk floats. */
typedef struct {
    int32_t B, H, Tq, Tk, head_dim;
    const float *q, *k, *v, *dout; int64_t ldq, ldk, ldv, lddo;
    const float* key_mask; float scale;
    float *dq, *dk, *dv; int64_t lddq, lddk, lddv;
    void* scratch; size_t scratch_bytes;
    float drop_p; uint32_t drop_site; uint64_t drop_seed;      /* the forward call's attention-probability dropout (0: none) */
} sprc_attention_bwd_args;
/* Dropout as the reference trains with it (nn.Dropout(p) at Qformer.py:113,264,293,379; blip_fine_tune_2.py:290 `.train()`):
 *   y[i] = keep(seed, site, i) ? x[i] / (1 - p) : 0      (+ resid[i] when resid != NULL: the post-dropout residual add of :294,380)
 * keep is a COUNTER-BASED mask -- z = seed + site * 0x9E3779B97F4A7C15 + i * 0xD1B54A32D192ED03 (mod 2^64), SplitMix64 finaliser,
 * keep <=> (z >> 32) >= floor(p * 2^32) -- so backward regenerates it (dx = sprc_dropout_f32(dy)) instead of storing it, and a CPU
 * restatement (oracle/sprc_oracle.py: drop_keep) injects the SAME masks into the reference for the gradient goldens.  x may alias y. */
int sprc_dropout_f32(const float* x, const float* resid, float* y, size_t n, uint64_t seed, uint32_t site, float p, sprc_stream s);
int sprc_attention_bwd(const sprc_attention_bwd_args* a, sprc_stream s);
/* The rows sprc_qformer_embed normalises, WITHOUT the LayerNorm (pre [B, Lq+Lt, hidden]), and the scatter of their gradient:
 * dquery[b * dq_bstride + row] += (dq_bstride 0: summed over the batch), dword[id] +=, dpos[position] += (atomic adds). */
int sprc_qformer_embed_rows(const sprc_qformer_embed_args* a, float* pre, sprc_stream s);
int sprc_qformer_embed_bwd(const sprc_qformer_embed_args* a, const float* dpre, float* dquery, int64_t dq_bstride, float* dword, float* dpos,
                           sprc_stream s);
/* sim[b, n] = max_j <fusion[b], feats[n, j]>: dfusion[b] += sum_n dsim[b, n] feats[n, j*], dfeats[n, j*] += dsim[b, n] fusion[b],
 * j* = the first argmax (torch.max); jstar: int32 [B, N] scratch. */
int sprc_sim_max_bwd(const float* fusion, const float* feats, const float* dsim, int32_t B, int32_t N, int32_t J, int32_t E, float* dfusion,
                     float* dfeats, int32_t* jstar, sprc_stream s);
/* Backward of sprc_contrastive_ce scaled by `grad`: dsim [B, B] (written), dtemp[0] += (optional). */
int sprc_contrastive_ce_bwd(const float* sim, int64_t ld, int32_t B, float temp, float grad, float* dsim, float* dtemp, sprc_stream s);
/* Backward of sprc_l2norm_rows from its input x. */
int sprc_l2norm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, float* dx, int64_t lddx, int32_t M, int32_t D, sprc_stream s);
/* Backward of sprc_align_mse scaled by `grad` into the first Lq rows of every sample of dh (+=); the prompt side is detached
 * (align_prompt.py:193). */
int sprc_align_mse_bwd(const float* h, int64_t sample_stride, int32_t Lq, int32_t D, const float* prompt, int32_t B, float grad, float* dh,
                       int64_t d_stride, sprc_stream s);

#ifdef __cplusplus
}
#endif
#endif /* SPRC_H */
