// models.hip -- composite forward passes: the kernels of gemm/attention/rowops sequenced on one
// stream, one C call per batch.  Activations live in a caller-provided workspace (bump allocated);
// the residual stream, LayerNorm statistics and softmax stay fp32, GEMM operands are `dtype`.
//
//   sprc_vit_forward    eva_vit.py:324-340 / clip_vit.py:171-185 + ln_vision (blip2.py:193-199)
//   sprc_qformer_image  Qformer.py:810-973 call shape (i)  + vision_proj + normalize (align_prompt.py:369-385)
//   sprc_qformer_fuse   call shapes (ii)+(iii) + text_proj + normalize (align_prompt.py:313-350)
#include "common.hpp"

namespace sprc {

struct Bump {
    char* base; size_t cap, off; bool ok;
    Bump(void* p, size_t c) : base((char*)p), cap(c), off(0), ok(true) {}
    void* take(size_t bytes) {
        const size_t o = align_up(off, 256);
        if (base == nullptr) { off = o + bytes; return nullptr; }     // sizing pass
        if (o + bytes > cap) { ok = false; return base; }
        off = o + bytes;
        return base + o;
    }
};

static const sprc_rowmap ID_MAP = {0, 0, 0};

struct Fp8Scales { const float* w_scale; float a_scale, out_scale; };

static int gemm(hipStream_t st, int dt, int out_dt, int M, int N, int K, const void* A, int64_t lda, const sprc_linear& w,
                void* C, int64_t ldc, int act = SPRC_ACT_NONE, const float* resid = nullptr, int64_t ldr = 0,
                sprc_rowmap amap = ID_MAP, sprc_rowmap cmap = ID_MAP, void* scratch = nullptr, size_t scratch_bytes = 0,
                const Fp8Scales* q = nullptr, int64_t ldw = 0, int k8 = 0, int k_alg = 0) {
    // k_alg: the ALGORITHMIC reduction length for the profiler when it differs from K (the patch embedding launches its zero-padded K)
    sprc_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.k8 = k8; g.k_alg = k_alg;
    if (q != nullptr) { g.w_scale = q->w_scale; g.a_scale = q->a_scale; g.out_scale = q->out_scale; }
    g.M = M; g.N = N; g.K = K; g.dtype = dt; g.out_dtype = out_dt; g.act = act;
    g.A = A; g.lda = lda; g.amap = amap;
    g.W = w.w; g.ldw = ldw > 0 ? ldw : K; g.bias = w.b;
    g.resid = resid; g.ldr = ldr;
    g.C = C; g.ldc = ldc; g.cmap = cmap;
    g.scratch = scratch; g.scratch_bytes = scratch_bytes;
    return sprc_gemm(&g, st);
}

// two products of identical shape in one launch (sprc_gemm_pair): w0 on the rows amap0 -> cmap0, w1 on amap1 -> cmap1
static int gemm2(hipStream_t st, int dt, int out_dt, int M, int N, int K, const void* A, int64_t lda, const sprc_linear& w0,
                 const sprc_linear& w1, void* C, int64_t ldc, int act, const float* resid, int64_t ldr, sprc_rowmap amap0,
                 sprc_rowmap amap1, sprc_rowmap cmap0, sprc_rowmap cmap1, int64_t ldw = 0, int k8 = 0) {
    sprc_gemm_args g[2];
    memset(g, 0, sizeof(g));
    for (int i = 0; i < 2; ++i) {
        g[i].k8 = k8;
        g[i].M = M; g[i].N = N; g[i].K = K; g[i].dtype = dt; g[i].out_dtype = out_dt; g[i].act = act;
        g[i].A = A; g[i].lda = lda; g[i].amap = i ? amap1 : amap0;
        g[i].W = (i ? w1 : w0).w; g[i].ldw = ldw > 0 ? ldw : K; g[i].bias = (i ? w1 : w0).b;
        g[i].resid = resid; g[i].ldr = ldr;
        g[i].C = C; g[i].ldc = ldc; g[i].cmap = i ? cmap1 : cmap0;
    }
    return sprc_gemm_pair(&g[0], &g[1], st);
}

// y = LN(x [+ add16]); sum32 (optional) receives x + add16 (the residual-stream update of a pre-LN block)
static int lnorm(hipStream_t st, int dt, int M, int D, const float* x, const float* gam, const float* bet, float eps,
                 float* y32, void* y16, sprc_rowmap map = ID_MAP, const void* add16 = nullptr, float* sum32 = nullptr,
                 float y16_scale = 0.f) {
    const int64_t ld16 = dt == SPRC_F16X3 ? 2 * (int64_t)D : D;          // fp16 units: a split row is 4 D bytes
    sprc_layernorm_args a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.D = D; a.out_dtype = dt;
    a.x = x; a.ldx = D; a.xmap = map;
    a.gamma = gam; a.beta = bet; a.eps = eps;
    a.y32 = y32; a.ld32 = D; a.ymap = map;
    a.y16 = y16; a.ld16 = ld16;
    a.add16 = add16; a.ld_add = D;
    a.sum32 = sum32; a.ld_sum = D;
    a.y16_scale = y16_scale;
    return sprc_layernorm(&a, st);
}

// where the cross-attention K|V rows of a batch come from: one [B, Ta, ld] buffer (image pass, fusion), or per-sample rows
// of two buffers through index arrays (stage-2 rerank: cat(reference, candidate) tokens)
struct KvSrc {
    const void* a; int Ta; const int32_t* ia;
    const void* b; int Tb; const int32_t* ib;       // b == nullptr: one segment
    int64_t ld;                                      // elements per token row: n_cross * 2 * hidden
};

static int attn(hipStream_t st, int dt, int B, int H, int Tq, int Tk, int dh, const void* q, int64_t ldq, const void* k,
                int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo, const float* mask, float scale,
                const KvSrc* two = nullptr, size_t seg2_off = 0, bool out_x3 = false) {
    sprc_attention_args a;
    memset(&a, 0, sizeof(a));
    a.out_x3 = out_x3 ? 1 : 0;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.head_dim = dh; a.dtype = dt;
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out = out; a.ldo = ldo;
    a.key_mask = mask; a.scale = scale;
    if (two != nullptr) {                            // k / v already point at segment 1; segment 2 at the same byte offsets
        a.kv_index = two->ia;
        if (two->b != nullptr) {
            a.k2 = (const char*)two->b + seg2_off; a.ldk2 = two->ld;
            a.v2 = (const char*)a.k2 + (size_t)((const char*)v - (const char*)k); a.ldv2 = two->ld;
            a.Tk2 = two->Tb; a.kv2_index = two->ib;
        }
    }
    return sprc_attention(&a, st);
}

// SPRC_FUSE_ADD=1 (default 0): fold the residual adds into the LayerNorm kernels (branch GEMMs write fp16 deltas, see
// sprc_layernorm_args.add16).  Measured on MI355X (tools/fuse_ab.sh): GEMM class 83.9 -> 78.5 ms per step (872 -> 932
// TFLOP/s) but the LayerNorms 5.8 -> 10.5 ms -- the same 14 B per element cross HBM either way, they only move from the
// GEMM epilogue burst to the row kernels -- i.e. +0.3 % images/s, while the fp16 rounding of every branch output (2^-11
// relative, not averaged over K like the bf16 operand roundings) takes the full-depth ViT-g cosine error from 4.3e-4 to
// 1.1e-3, over the 1e-3 bar.  Kept as an A/B switch; off.
static bool fuse_add_enabled(int which = 1) {            // bit 0: ViT blocks, bit 1: Q-Former layers
    static const int on = [] { const char* e = getenv("SPRC_FUSE_ADD"); return e ? atoi(e) : 0; }();
    return (on & which) != 0;
}

#define RUN(x)                      \
    do {                            \
        int rc__ = (x);             \
        if (rc__ != SPRC_OK) return rc__; \
    } while (0)

// ------------------------------------------------------------------------------------------------
struct VitBufs { void *rows, *h, *qkv, *ctx, *mlp; float *pout, *x; };

static size_t vit_plan(const sprc_vit_model* m, int B, Bump& b, VitBufs& v) {
    const size_t es = dtype_size(m->dtype);
    const size_t M = (size_t)B * m->tokens, P = (size_t)B * (m->tokens - 1), D = m->width;
    v.rows = b.take(P * m->patch_k_pad * es * (m->patch_x3 ? 2 : 1));          // split rows: 4 bytes per column
    v.pout = (float*)b.take(P * D * 4);
    v.x = (float*)b.take(M * D * 4);
    v.h = b.take(M * D * es);
    v.qkv = b.take(M * 3 * D * es);
    v.ctx = b.take(M * D * es);
    v.mlp = b.take(M * (size_t)m->mlp * es);
    return b.off;
}

// ------------------------------------------------------------------------------------------------
struct QfBufs {
    void *enc, *kv, *h16, *a16, *g16, *qkv, *ctx, *cq, *ffn;
    float *h32, *a32, *g32, *t32, *proj, *mask;
};

static size_t qf_plan(const sprc_qformer_model* m, int B, int enc_tokens, Bump& b, QfBufs& q, bool with_kv = true) {
    const size_t es = dtype_size(m->dtype);
    const size_t S = (size_t)m->num_query + m->max_txt, R = (size_t)B * S, Hd = m->hidden;
    const size_t E = (size_t)B * enc_tokens;
    const size_t kx = m->x3 ? 2 : 1;               // split-precision Q-Former: GEMM input activations are split rows (SPRC_F16X3: 4 B / column)
    q.enc = (with_kv && is16(m->dtype)) ? b.take(E * m->enc_width * es * ((m->x3 & SPRC_X3_CKV) ? 2 : 1)) : nullptr;
    q.kv = with_kv ? b.take(E * (size_t)m->n_cross * 2 * Hd * es) : nullptr;
    q.h32 = (float*)b.take(R * Hd * 4); q.h16 = b.take(R * Hd * es * kx);
    q.a32 = (float*)b.take(R * Hd * 4); q.a16 = b.take(R * Hd * es * kx);
    q.g32 = (float*)b.take(R * Hd * 4); q.g16 = b.take(R * Hd * es * kx);
    q.t32 = (float*)b.take(R * Hd * 4);
    q.qkv = b.take(R * 3 * Hd * es);
    q.ctx = b.take(R * Hd * es * kx);
    q.cq = b.take(R * Hd * es);
    q.ffn = b.take(R * (size_t)m->ffn * es * kx);
    q.proj = (float*)b.take(R * (size_t)m->embed_dim * 4);
    q.mask = (float*)b.take(R * 4);
    return b.off;
}

// which activation buffers of a call with layer-kind mask `cm` are kept in the split layout: a buffer is split when one of the
// layers that read it runs the split product (fp16 + e4m3 segments)
struct X3Layouts { bool ln, ctx, ffn; };
static X3Layouts x3_layouts(int cm) {
    return X3Layouts{(cm & (SPRC_X3_QKV | SPRC_X3_CROSS_Q | SPRC_X3_FFN_IN | SPRC_X3_HEADS)) != 0,
                     (cm & (SPRC_X3_ATTN_OUT | SPRC_X3_CROSS_OUT)) != 0, (cm & SPRC_X3_FFN_OUT) != 0};
}

// one Q-Former encoder stack over x32/x16 [B, S, hidden]; cross-attention + query FFN on rows [:Lq] when
// `kv` is given (Qformer.py:434-468), text FFN on rows [Lq:]; text FFN on all rows otherwise (:469-475).
// Split-precision (fp16 engine): `cm` is the CALL's mask over layer kinds (SPRC_X3_*, a subset of the packed mask m->x3).
// A layer whose kind is in cm runs the split product (sprc_gemm k8 = 2 K) on its input -- kept as SPRC_F16X3 rows [hi fp16 | lo e4m3 |
// hi e4m3], 4 x width bytes -- against split weight rows [W_hi | W 2^6 | W_lo 2^18] (fp16 K-tiles, then e4m3 K-tiles on the MX-scaled
// MFMA: products to ~2^-16); the others reduce over the hi segment only (packed weight rows are then read through their pitch).  The three groups
// of activation buffers (LayerNorm copies, attention outputs, FFN hidden) are split only when one of their consumers is in cm
// (x3_layouts).  q / k / v and the attention probabilities stay plain fp16.
// need_n > 0: only rows [need_lo, need_lo + need_n) of every sample are read after the stack (need_n a power of two; either a subset
// of the query rows when `kv` is given, or -- without `kv` -- any run of rows): the LAST layer then runs everything behind its
// self-attention (output projection, cross-attention, FFNs, LayerNorms) on those rows alone.  The other rows of x32 / x16 keep the
// previous layer's values.  Pass 1 of the fusion is read on its 32 query rows (align_prompt.py:341-346), pass 2 on the one [CLS] row
// (:348-350), the text passes on row 0, the ITM pass on its query rows: work nobody reads, 0.5 ms of kernel time per bench step.
static bool qf_dead_rows_enabled() {
    static const int on = [] { const char* e = getenv("SPRC_QF_DEAD"); return e ? atoi(e) : 1; }();
    return on != 0;
}

static int qf_stack(const sprc_qformer_model* m, hipStream_t st, QfBufs& q, int B, int S, const KvSrc* kvs,
                    const float* mask, float* x32, void* x16, int cm, int need_lo = 0, int need_n = 0) {
    const bool with_enc = kvs != nullptr;
    const int dt = m->dtype, Hd = m->hidden, H = m->heads, dh = m->head_dim, F = m->ffn, Lq = m->num_query;
    const X3Layouts lay = x3_layouts(cm);
    const int adt = lay.ln ? SPRC_F16X3 : dt, fdt = lay.ffn ? SPRC_F16X3 : dt;      // dtype of the LayerNorm copies / the FFN hidden
    const int KH = lay.ln ? 2 * Hd : Hd, KC = lay.ctx ? 2 * Hd : Hd, KF = lay.ffn ? 2 * F : F;   // their leading dimensions in fp16 units (ctx: KC)
    auto k8of = [&](int kind, int width) { return (cm & kind) ? 2 * width : 0; };              // e4m3 correction elements a layer of this call reduces over
    auto wld = [&](int kind, int width) { return (int64_t)((m->x3 & kind) ? 2 * width : width); };   // its weights' leading dimension (fp16 units)
    const int R = B * S;
    const float sc = 1.0f / sqrtf((float)dh);                                   // Qformer.py:250
    const size_t es = dtype_size(dt);
    const sprc_rowmap qmap = {Lq, S, 0}, tmap = {S - Lq, S, Lq};
    const bool split = with_enc && S > Lq;         // rows [:Lq] and [Lq:] take different paths
    // SPRC_FUSE_ADD=1: the post-LN residual adds (Qformer.py:294,380) ride on the LayerNorm kernels (sprc_layernorm add16),
    // the branch GEMMs write fp16 instead of running an fp32 + residual epilogue.  Off by default (see fuse_add_enabled).
    const bool fuse_add = dt == SPRC_BF16 && fuse_add_enabled(2);
    // y = act(in . W^T + b) of layer kind `kind` over `width` input columns, into a compute-dtype (or x3) buffer
    auto lin = [&](int kind, int rows, int N, int width, const void* in, int64_t lda, const sprc_linear& w, int odt, void* out,
                   int64_t ldc, int act, sprc_rowmap amap, sprc_rowmap cmap) -> int {
        return gemm(st, dt, odt, rows, N, width, in, lda, w, out, ldc, act, nullptr, 0, amap, cmap, nullptr, 0, nullptr,
                    wld(kind, width), k8of(kind, width));
    };
    // a = LN(dense(in) + res): branch GEMM of layer kind `kind` over `width` input columns, post-LN into (o32, o16)
    auto branch = [&](int kind, int rows, int width, int64_t lda, const void* in, const sprc_linear& w, const float* res,
                      const float* lw, const float* lb, float* o32, void* o16, sprc_rowmap amap, sprc_rowmap cmap) -> int {
        if (fuse_add) {         // the branch output goes out as fp16 into a16 (dead here) and is added by the LN
            RUN(gemm(st, dt, SPRC_F16, rows, Hd, width, in, lda, w, q.a16, Hd, SPRC_ACT_NONE, nullptr, 0, amap, cmap));
            return lnorm(st, dt, rows, Hd, res, lw, lb, m->ln_eps, o32, o16, cmap, q.a16);
        }
        RUN(gemm(st, dt, SPRC_F32, rows, Hd, width, in, lda, w, q.t32, Hd, SPRC_ACT_NONE, res, Hd, amap, cmap, nullptr, 0,
                 nullptr, wld(kind, width), k8of(kind, width)));
        return lnorm(st, adt, rows, Hd, q.t32, lw, lb, m->ln_eps, o32, o16, cmap);
    };
    const bool prune = qf_dead_rows_enabled() && !fuse_add && need_n > 0 && need_n < S && (need_n & (need_n - 1)) == 0 &&
                       (with_enc ? (split && need_lo == 0 && need_n == Lq) : true);
    for (int l = 0; l < m->n_layers; ++l) {
        const sprc_qf_layer& L = m->layers[l];
        // self-attention over all S rows
        RUN(lin(SPRC_X3_QKV, R, 3 * Hd, Hd, x16, KH, L.qkv, dt, q.qkv, 3 * Hd, SPRC_ACT_NONE, ID_MAP, ID_MAP));
        RUN(attn(st, dt, B, H, S, S, dh, q.qkv, 3 * Hd, (char*)q.qkv + Hd * es, 3 * Hd, (char*)q.qkv + 2 * Hd * es, 3 * Hd,
                 q.ctx, KC, mask, sc, nullptr, 0, lay.ctx));
        if (prune && l == m->n_layers - 1) {
            // last layer, rows nobody reads dropped: nmap = the needed rows of every sample
            const sprc_rowmap nmap = {need_n, S, need_lo};
            const int Rn = B * need_n;
            RUN(branch(SPRC_X3_ATTN_OUT, Rn, Hd, KC, q.ctx, L.attn_out, x32, L.attn_ln_w, L.attn_ln_b, q.a32, q.a16, nmap, nmap));
            if (with_enc) {                       // the needed rows are the query rows: cross-attention + query FFN as in any layer, no text FFN
                if (L.has_cross) {
                    RUN(lin(SPRC_X3_CROSS_Q, Rn, Hd, Hd, q.a16, KH, L.cq, dt, q.cq, Hd, SPRC_ACT_NONE, nmap, ID_MAP));
                    const size_t off = (size_t)L.cross_index * 2 * Hd * es;
                    const char* kp = (const char*)kvs->a + off;
                    const bool plain = kvs->b == nullptr && kvs->ia == nullptr;
                    RUN(attn(st, dt, B, H, Lq, kvs->Ta, dh, q.cq, Hd, kp, kvs->ld, kp + Hd * es, kvs->ld, q.ctx, KC, nullptr, sc,
                             plain ? nullptr : kvs, off, lay.ctx));
                    RUN(branch(SPRC_X3_CROSS_OUT, Rn, Hd, KC, q.ctx, L.cross_out, q.a32, L.cross_ln_w, L.cross_ln_b, q.a32, q.a16, ID_MAP, nmap));
                }
                RUN(lin(SPRC_X3_FFN_IN, Rn, F, Hd, q.a16, KH, L.ffn_q_in, fdt, q.ffn, KF, SPRC_ACT_GELU, nmap, ID_MAP));
                RUN(branch(SPRC_X3_FFN_OUT, Rn, F, KF, q.ffn, L.ffn_q_out, q.a32, L.ffn_q_ln_w, L.ffn_q_ln_b, x32, x16, ID_MAP, nmap));
            } else {
                RUN(lin(SPRC_X3_FFN_IN, Rn, F, Hd, q.a16, KH, L.ffn_t_in, fdt, q.ffn, KF, SPRC_ACT_GELU, nmap, ID_MAP));
                RUN(branch(SPRC_X3_FFN_OUT, Rn, F, KF, q.ffn, L.ffn_t_out, q.a32, L.ffn_t_ln_w, L.ffn_t_ln_b, x32, x16, ID_MAP, nmap));
            }
            break;
        }
        RUN(branch(SPRC_X3_ATTN_OUT, R, Hd, KC, q.ctx, L.attn_out, x32, L.attn_ln_w, L.attn_ln_b, q.a32, q.a16, ID_MAP, ID_MAP));
        if (with_enc) {
            const sprc_rowmap rq = split ? qmap : ID_MAP;
            const int Rq = B * Lq;
            if (L.has_cross) {
                RUN(lin(SPRC_X3_CROSS_Q, Rq, Hd, Hd, q.a16, KH, L.cq, dt, q.cq, Hd, SPRC_ACT_NONE, rq, ID_MAP));
                const size_t off = (size_t)L.cross_index * 2 * Hd * es;          // this layer's K|V block inside a token row
                const char* kp = (const char*)kvs->a + off;
                const bool plain = kvs->b == nullptr && kvs->ia == nullptr;
                RUN(attn(st, dt, B, H, Lq, kvs->Ta, dh, q.cq, Hd, kp, kvs->ld, kp + Hd * es, kvs->ld, q.ctx, KC, nullptr, sc,
                         plain ? nullptr : kvs, off, lay.ctx));
                RUN(branch(SPRC_X3_CROSS_OUT, Rq, Hd, KC, q.ctx, L.cross_out, q.a32, L.cross_ln_w, L.cross_ln_b, q.a32, q.a16, ID_MAP, rq));
            }
            if (split && S - Lq == Lq) {
                // query rows and text rows of every sample go through different FFN weights (Qformer.py:455-475): the two
                // products have the same shape, so each pair is ONE launch (sprc_gemm_pair) -- 360 + 360 tiles instead of two
                // 1.4-round grids for the up projection, 90 + 90 instead of two third-empty grids for the down projection.
                // The hidden activations keep the rows' natural positions in q.ffn [R, F].
                RUN(gemm2(st, dt, fdt, Rq, F, Hd, q.a16, KH, L.ffn_q_in, L.ffn_t_in, q.ffn, KF, SPRC_ACT_GELU, nullptr, 0,
                          qmap, tmap, qmap, tmap, wld(SPRC_X3_FFN_IN, Hd), k8of(SPRC_X3_FFN_IN, Hd)));
                if (fuse_add) {
                    RUN(gemm2(st, dt, SPRC_F16, Rq, Hd, F, q.ffn, F, L.ffn_q_out, L.ffn_t_out, q.a16, Hd, SPRC_ACT_NONE, nullptr, 0, qmap,
                              tmap, qmap, tmap));
                    RUN(lnorm(st, dt, Rq, Hd, q.a32, L.ffn_q_ln_w, L.ffn_q_ln_b, m->ln_eps, x32, x16, qmap, q.a16));
                    RUN(lnorm(st, dt, Rq, Hd, q.a32, L.ffn_t_ln_w, L.ffn_t_ln_b, m->ln_eps, x32, x16, tmap, q.a16));
                } else {
                    RUN(gemm2(st, dt, SPRC_F32, Rq, Hd, F, q.ffn, KF, L.ffn_q_out, L.ffn_t_out, q.t32, Hd, SPRC_ACT_NONE,
                              q.a32, Hd, qmap, tmap, qmap, tmap, wld(SPRC_X3_FFN_OUT, F), k8of(SPRC_X3_FFN_OUT, F)));
                    RUN(lnorm(st, adt, Rq, Hd, q.t32, L.ffn_q_ln_w, L.ffn_q_ln_b, m->ln_eps, x32, x16, qmap));
                    RUN(lnorm(st, adt, Rq, Hd, q.t32, L.ffn_t_ln_w, L.ffn_t_ln_b, m->ln_eps, x32, x16, tmap));
                }
            } else {
                RUN(lin(SPRC_X3_FFN_IN, Rq, F, Hd, q.a16, KH, L.ffn_q_in, fdt, q.ffn, KF, SPRC_ACT_GELU, rq, ID_MAP));
                RUN(branch(SPRC_X3_FFN_OUT, Rq, F, KF, q.ffn, L.ffn_q_out, q.a32, L.ffn_q_ln_w, L.ffn_q_ln_b, x32, x16, ID_MAP, rq));
                if (split) {
                    const int Rt = B * (S - Lq);
                    RUN(lin(SPRC_X3_FFN_IN, Rt, F, Hd, q.a16, KH, L.ffn_t_in, fdt, q.ffn, KF, SPRC_ACT_GELU, tmap, ID_MAP));
                    RUN(branch(SPRC_X3_FFN_OUT, Rt, F, KF, q.ffn, L.ffn_t_out, q.a32, L.ffn_t_ln_w, L.ffn_t_ln_b, x32, x16, ID_MAP, tmap));
                }
            }
        } else {
            RUN(lin(SPRC_X3_FFN_IN, R, F, Hd, q.a16, KH, L.ffn_t_in, fdt, q.ffn, KF, SPRC_ACT_GELU, ID_MAP, ID_MAP));
            RUN(branch(SPRC_X3_FFN_OUT, R, F, KF, q.ffn, L.ffn_t_out, q.a32, L.ffn_t_ln_w, L.ffn_t_ln_b, x32, x16, ID_MAP, ID_MAP));
        }
    }
    return SPRC_OK;
}

// K|V projections of the image tokens for every cross-attention layer in ONE GEMM (Qformer.py:191-193)
static int qf_encode_kv(const sprc_qformer_model* m, hipStream_t st, void* enc16, void* kv_out, const float* enc32, int B,
                        int enc_tokens, int cm) {
    const int E = B * enc_tokens;
    const void* enc = enc32;
    const bool kv3 = (cm & SPRC_X3_CKV) != 0;
    if (kv3) {
        RUN(sprc_cast_f32_to_x3(enc32, enc16, E, m->enc_width, st));
        enc = enc16;
    } else if (is16(m->dtype)) {
        RUN(sprc_cast_f32_to_16(enc32, enc16, (size_t)E * m->enc_width, m->dtype, st));
        enc = enc16;
    }
    const int Nkv = m->n_cross * 2 * m->hidden, Kw = m->enc_width;
    return gemm(st, m->dtype, m->dtype, E, Nkv, Kw, enc, Kw * (kv3 ? 2 : 1), m->ckv_all, kv_out, Nkv, SPRC_ACT_NONE, nullptr, 0, ID_MAP, ID_MAP, nullptr, 0,
                nullptr, (int64_t)Kw * ((m->x3 & SPRC_X3_CKV) ? 2 : 1), kv3 ? 2 * Kw : 0);
}

// out[rows, embed_dim] (fp32) = proj(hidden rows `amap` of x16): the ITC heads (align_prompt.py:348,385)
static int head(const sprc_qformer_model* m, hipStream_t st, int cm, int rows, const void* x16, const sprc_linear& w, float* out,
                sprc_rowmap amap) {
    const int Hd = m->hidden;
    return gemm(st, m->dtype, SPRC_F32, rows, m->embed_dim, Hd, x16, x3_layouts(cm).ln ? 2 * Hd : Hd, w, out, m->embed_dim,
                SPRC_ACT_NONE, nullptr, 0, amap, ID_MAP, nullptr, 0, nullptr, (int64_t)((m->x3 & SPRC_X3_HEADS) ? 2 * Hd : Hd), (cm & SPRC_X3_HEADS) ? 2 * Hd : 0);
}

static int check_qf(const sprc_qformer_model* m) {
    SPRC_REQUIRE(m && m->layers, "qformer: null model");
    SPRC_REQUIRE(is16(m->dtype) || m->dtype == SPRC_F32, "qformer: bad dtype");
    SPRC_REQUIRE(!m->x3 || m->dtype == SPRC_F16, "qformer: the split-precision mode (x3) needs dtype SPRC_F16");
    SPRC_REQUIRE((m->x3_image & ~m->x3) == 0 && (m->x3_fuse & ~m->x3) == 0, "qformer: x3_image / x3_fuse must be subsets of the packed mask x3");
    SPRC_REQUIRE(m->hidden == m->heads * m->head_dim, "qformer: hidden != heads*head_dim");
    SPRC_REQUIRE((m->num_query & (m->num_query - 1)) == 0 && (m->max_txt & (m->max_txt - 1)) == 0,
                 "qformer: num_query and max_txt must be powers of two");
    return SPRC_OK;
}

}  // namespace sprc

using namespace sprc;

extern "C" size_t sprc_vit_workspace_bytes(const sprc_vit_model* m, int32_t B) {
    if (!m || B <= 0) return 0;
    Bump b(nullptr, 0);
    VitBufs v;
    return vit_plan(m, B, b, v) + 256;
}

extern "C" size_t sprc_qformer_workspace_bytes(const sprc_qformer_model* m, int32_t B) {
    if (!m || B <= 0) return 0;
    Bump b(nullptr, 0);
    QfBufs q;
    return qf_plan(m, B, 257, b, q) + 256;
}

extern "C" int sprc_vit_forward(const sprc_vit_model* m, const float* images, int32_t B, float* raw, void* ws,
                                size_t ws_bytes, sprc_stream s) {
    SPRC_REQUIRE(m && m->layers && images && raw && ws, "sprc_vit_forward: null pointer");
    SPRC_REQUIRE(B > 0, "sprc_vit_forward: B=%d", B);
    SPRC_REQUIRE(is16(m->dtype) || m->dtype == SPRC_F32, "sprc_vit_forward: bad dtype");
    SPRC_REQUIRE(m->width == m->heads * m->head_dim, "sprc_vit_forward: width != heads*head_dim");
    SPRC_REQUIRE(((uintptr_t)ws % 256) == 0, "sprc_vit_forward: workspace must be 256-byte aligned");
    Bump b(ws, ws_bytes);
    VitBufs v;
    vit_plan(m, B, b, v);
    if (!b.ok) {
        set_error("sprc_vit_forward: workspace too small (%zu bytes given)", ws_bytes);
        return SPRC_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)s;
    const int dt = m->dtype, D = m->width, T = m->tokens, M = B * T, P = B * (T - 1), F = m->mlp;
    const size_t es = dtype_size(dt);
    const size_t pout_bytes = (size_t)P * D * 4;
    const float scale = 1.0f / sqrtf((float)m->head_dim);                       // eva_vit.py:74
    SPRC_REQUIRE(!m->patch_x3 || dt == SPRC_F16, "sprc_vit_forward: patch_x3 is a mode of the fp16 model");
    const int patch_k = 3 * m->patch_size * m->patch_size;     // the convolution's own reduction length (588; launched zero padded)
    if (m->patch_x3) {                                      // split-precision patch embedding: fp16 product + e4m3 correction segments
        RUN(sprc_im2row(images, v.rows, B, m->image, m->patch_size, m->patch_k_pad, SPRC_F16X3, st));
        RUN(gemm(st, dt, SPRC_F32, P, D, m->patch_k_pad, v.rows, 2 * m->patch_k_pad, m->patch, v.pout, D, SPRC_ACT_NONE, nullptr, 0, ID_MAP,
                 ID_MAP, nullptr, 0, nullptr, 2 * m->patch_k_pad, 2 * m->patch_k_pad, patch_k));
    } else {
        RUN(sprc_im2row(images, v.rows, B, m->image, m->patch_size, m->patch_k_pad, dt, st));
        RUN(gemm(st, dt, SPRC_F32, P, D, m->patch_k_pad, v.rows, m->patch_k_pad, m->patch, v.pout, D, SPRC_ACT_NONE, nullptr, 0, ID_MAP, ID_MAP,
                 nullptr, 0, nullptr, 0, 0, patch_k));
    }
    RUN(sprc_vit_assemble(v.pout, m->cls, m->pos, v.x, B, T, D, st));
    if (m->has_ln_pre) RUN(lnorm(st, dt, M, D, v.x, m->ln_pre_w, m->ln_pre_b, m->ln_eps, v.x, nullptr));
    // ---- transformer blocks.  SPRC_VIT_STREAMS=2 runs the two halves of the batch on two streams (the caller's and one
    // helper stream owned by the library): samples are independent, and with two GEMM streams in flight one kernel's
    // output-write burst, its partial last round of tiles and the HBM-bound LayerNorms overlap the other half's matrix work
    // (measured +4 % images/s end to end, -7 % per ViT layer; four quarters gain nothing).  Everything on the helper stream
    // is ordered after `s` up to here and joined back into `s` before returning, so the call keeps its in-stream semantics.
    // Off by default: with kernels of two streams sharing the chip a per-kernel duration no longer measures the kernel.
    static const int n_streams = [] { const char* e = getenv("SPRC_VIT_STREAMS"); return e ? atoi(e) : 1; }();
    const bool split = n_streams >= 2 && B >= 16 && is16(dt);
    const int B0 = split ? B / 2 : B;
    hipStream_t st2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    if (split) {
        // one helper stream + fork/join events PER DEVICE (streams and events belong to the device they were created on)
        struct Helper { hipStream_t stream; hipEvent_t fork, join; };
        static Helper helpers[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        Helper& hp = helpers[dev >= 0 && dev < 64 ? dev : 0];
        if (hp.stream == nullptr) {
            SPRC_REQUIRE(hipStreamCreateWithFlags(&hp.stream, hipStreamNonBlocking) == hipSuccess, "sprc_vit_forward: cannot create the helper stream");
            (void)hipEventCreateWithFlags(&hp.fork, hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&hp.join, hipEventDisableTiming);
        }
        st2 = hp.stream; ev_fork = hp.fork; ev_join = hp.join;
        (void)hipEventRecord(ev_fork, st);
        (void)hipStreamWaitEvent(st2, ev_fork, 0);
    }
    struct Part { hipStream_t ps; int Bp, Mp; float* x; char *h, *qkv, *ctx, *mlp, *scratch; float* raw; };
    Part parts[2];
    const int nparts = split ? 2 : 1;
    const size_t scratch_bytes = pout_bytes / 2 - 256;      // patch-embedding output, dead by now: split-K partials
    for (int i = 0; i < nparts; ++i) {
        const int b0 = i == 0 ? 0 : B0, Bp = i == 0 ? B0 : B - B0;
        const size_t r0 = (size_t)b0 * T;
        parts[i] = Part{i == 0 ? st : st2, Bp, Bp * T, v.x + r0 * D, (char*)v.h + r0 * D * es, (char*)v.qkv + r0 * 3 * D * es,
                        (char*)v.ctx + r0 * D * es, (char*)v.mlp + r0 * (size_t)F * es,
                        (char*)v.pout + (i == 0 ? 0 : align_up(pout_bytes / 2, 256)), raw + r0 * D};
    }
    // SPRC_FUSE_ADD=1: x += proj(..) and x += fc2(..) (eva_vit.py:178-179, clip_vit.py:137-138) are folded into the LayerNorm
    // that reads x next: the branch GEMM writes its output as fp16 into the h buffer (dead at that point; the LN overwrites
    // it in place with its bf16 result) and the LN kernel adds it to the fp32 residual stream (sum32 = x) before the
    // statistics.  Off by default (see fuse_add_enabled: no net gain, and it costs parity).
    const bool fp8 = m->fp8 != 0, fp8_qkv = m->fp8 == SPRC_FP8_ALL;
    const bool fuse_add = dt == SPRC_BF16 && !fp8 && fuse_add_enabled(1);
    SPRC_REQUIRE(m->fp8 == 0 || m->fp8 == SPRC_FP8_ALL || m->fp8 == SPRC_FP8_MLP, "sprc_vit_forward: fp8 = %d", m->fp8);
    SPRC_REQUIRE(!fp8 || (is16(dt) && D % 128 == 0 && F % 128 == 0), "sprc_vit_forward: the fp8 path needs a 16-bit model with width, mlp %% 128 == 0");
    SPRC_REQUIRE(!(fp8 && m->calib_amax), "sprc_vit_forward: calibrate on the 16-bit model, not on the fp8 one");
    float* calib = (is16(dt) && !fp8) ? m->calib_amax : nullptr;
    for (int l = 0; l < m->depth; ++l) {                    // enqueue layer by layer, alternating streams: both stay fed
        const sprc_vit_layer& L = m->layers[l];
        for (int i = 0; i < nparts; ++i) {
            const Part& q = parts[i];
            if (fp8) {
                // fp8 ViT block: LN -> e4m3 operand (static scale), qkv / fc1 / fc2 on fp8 MFMA with per-channel weight scales,
                // fc1's GELU output quantised in its epilogue; attention and proj stay bf16; residual stream / LN fp32.
                // h (bf16-sized) holds the fp8 operand: half its bytes, leading dimension D bytes.
                const Fp8Scales s_qkv{L.qkv_ws, L.s_ln1, 0.f}, s_fc1{L.fc1_ws, L.s_ln2, 1.0f / L.s_mlp}, s_fc2{L.fc2_ws, L.s_mlp, 0.f};
                if (fp8_qkv) {
                    RUN(lnorm(q.ps, SPRC_FP8, q.Mp, D, q.x, L.ln1_w, L.ln1_b, m->ln_eps, nullptr, q.h, ID_MAP, nullptr, nullptr, 1.0f / L.s_ln1));
                    RUN(gemm(q.ps, SPRC_FP8, dt, q.Mp, 3 * D, D, q.h, D, L.qkv, q.qkv, 3 * D, SPRC_ACT_NONE, nullptr, 0, ID_MAP, ID_MAP,
                             nullptr, 0, &s_qkv));
                } else {
                    RUN(lnorm(q.ps, dt, q.Mp, D, q.x, L.ln1_w, L.ln1_b, m->ln_eps, nullptr, q.h));
                    RUN(gemm(q.ps, dt, dt, q.Mp, 3 * D, D, q.h, D, L.qkv, q.qkv, 3 * D));
                }
                RUN(attn(q.ps, dt, q.Bp, m->heads, T, T, m->head_dim, q.qkv, 3 * D, q.qkv + D * es, 3 * D, q.qkv + 2 * D * es, 3 * D,
                         q.ctx, D, nullptr, scale));
                RUN(gemm(q.ps, dt, SPRC_F32, q.Mp, D, D, q.ctx, D, L.proj, q.x, D, SPRC_ACT_NONE, q.x, D));
                RUN(lnorm(q.ps, SPRC_FP8, q.Mp, D, q.x, L.ln2_w, L.ln2_b, m->ln_eps, nullptr, q.h, ID_MAP, nullptr, nullptr, 1.0f / L.s_ln2));
                RUN(gemm(q.ps, SPRC_FP8, SPRC_FP8, q.Mp, F, D, q.h, D, L.fc1, q.mlp, F, m->act, nullptr, 0, ID_MAP, ID_MAP, nullptr, 0, &s_fc1));
                RUN(gemm(q.ps, SPRC_FP8, SPRC_F32, q.Mp, D, F, q.mlp, F, L.fc2, q.x, D, SPRC_ACT_NONE, q.x, D, ID_MAP, ID_MAP, q.scratch,
                         scratch_bytes, &s_fc2));
                continue;
            }
            if (fuse_add) {
                const bool pend = l > 0;                    // fc2 output of the previous layer waits in h
                RUN(lnorm(q.ps, dt, q.Mp, D, q.x, L.ln1_w, L.ln1_b, m->ln_eps, nullptr, q.h, ID_MAP, pend ? q.h : nullptr,
                          pend ? q.x : nullptr));
            } else {
                RUN(lnorm(q.ps, dt, q.Mp, D, q.x, L.ln1_w, L.ln1_b, m->ln_eps, nullptr, q.h));
            }
            if (calib) RUN(sprc_absmax_16(q.h, (size_t)q.Mp * D, dt, calib + l * 3 + 0, q.ps));
            RUN(gemm(q.ps, dt, dt, q.Mp, 3 * D, D, q.h, D, L.qkv, q.qkv, 3 * D));
            RUN(attn(q.ps, dt, q.Bp, m->heads, T, T, m->head_dim, q.qkv, 3 * D, q.qkv + D * es, 3 * D, q.qkv + 2 * D * es, 3 * D,
                     q.ctx, D, nullptr, scale));
            if (fuse_add) {
                RUN(gemm(q.ps, dt, SPRC_F16, q.Mp, D, D, q.ctx, D, L.proj, q.h, D));
                RUN(lnorm(q.ps, dt, q.Mp, D, q.x, L.ln2_w, L.ln2_b, m->ln_eps, nullptr, q.h, ID_MAP, q.h, q.x));
            } else {
                RUN(gemm(q.ps, dt, SPRC_F32, q.Mp, D, D, q.ctx, D, L.proj, q.x, D, SPRC_ACT_NONE, q.x, D));
                RUN(lnorm(q.ps, dt, q.Mp, D, q.x, L.ln2_w, L.ln2_b, m->ln_eps, nullptr, q.h));
            }
            if (calib) RUN(sprc_absmax_16(q.h, (size_t)q.Mp * D, dt, calib + l * 3 + 1, q.ps));
            RUN(gemm(q.ps, dt, dt, q.Mp, F, D, q.h, D, L.fc1, q.mlp, F, m->act));
            if (calib) RUN(sprc_absmax_16(q.mlp, (size_t)q.Mp * F, dt, calib + l * 3 + 2, q.ps));
            if (fuse_add) {
                RUN(gemm(q.ps, dt, SPRC_F16, q.Mp, D, F, q.mlp, F, L.fc2, q.h, D, SPRC_ACT_NONE, nullptr, 0, ID_MAP, ID_MAP, q.scratch,
                         scratch_bytes));
            } else {
                RUN(gemm(q.ps, dt, SPRC_F32, q.Mp, D, F, q.mlp, F, L.fc2, q.x, D, SPRC_ACT_NONE, q.x, D, ID_MAP, ID_MAP, q.scratch,
                         scratch_bytes));
            }
        }
    }
    if (m->pre_ln_out != nullptr) {
        SPRC_REQUIRE(!fuse_add, "sprc_vit_forward: pre_ln_out is not available with SPRC_FUSE_ADD");
        for (int i = 0; i < nparts; ++i) {
            const size_t r0 = (size_t)(parts[i].x - v.x);
            SPRC_REQUIRE(hipMemcpyAsync(m->pre_ln_out + r0, parts[i].x, (size_t)parts[i].Mp * D * sizeof(float), hipMemcpyDeviceToDevice,
                                        parts[i].ps) == hipSuccess, "sprc_vit_forward: copy of the pre-LayerNorm stream failed");
        }
    }
    for (int i = 0; i < nparts; ++i)
        RUN(lnorm(parts[i].ps, dt, parts[i].Mp, D, parts[i].x, m->ln_vision_w, m->ln_vision_b, m->ln_vision_eps, parts[i].raw, nullptr,
                  ID_MAP, fuse_add && m->depth > 0 ? parts[i].h : nullptr));
    if (split) {
        (void)hipEventRecord(ev_join, st2);
        (void)hipStreamWaitEvent(st, ev_join, 0);
    }
    return SPRC_OK;
}

extern "C" int sprc_qformer_image(const sprc_qformer_model* m, const float* raw, int32_t B, float* feats, void* feats16,
                                  void* ws, size_t ws_bytes, sprc_stream s) {
    RUN(check_qf(m));
    SPRC_REQUIRE(raw && feats && ws && B > 0, "sprc_qformer_image: bad arguments");
    SPRC_REQUIRE(((uintptr_t)ws % 256) == 0, "sprc_qformer_image: workspace must be 256-byte aligned");
    const int T = 257;
    Bump b(ws, ws_bytes);
    QfBufs q;
    qf_plan(m, B, T, b, q);
    if (!b.ok) {
        set_error("sprc_qformer_image: workspace too small (%zu bytes given)", ws_bytes);
        return SPRC_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)s;
    const int dt = m->dtype, Hd = m->hidden, Lq = m->num_query;
    const int cm = m->x3_image;
    RUN(qf_encode_kv(m, st, q.enc, q.kv, raw, B, T, cm));
    const KvSrc kvs{q.kv, T, nullptr, nullptr, 0, nullptr, (int64_t)m->n_cross * 2 * Hd};
    sprc_qformer_embed_args e;
    memset(&e, 0, sizeof(e));
    e.B = B; e.Lq = Lq; e.Lt = 0; e.hidden = Hd; e.out_dtype = x3_layouts(cm).ln ? SPRC_F16X3 : dt;
    e.query_embeds = m->query_tokens; e.q_bstride = 0;
    e.gamma = m->emb_ln_w; e.beta = m->emb_ln_b; e.eps = m->ln_eps;
    e.y32 = q.h32; e.y16 = q.h16;
    RUN(sprc_qformer_embed(&e, st));
    RUN(qf_stack(m, st, q, B, Lq, &kvs, nullptr, q.h32, q.h16, cm));
    RUN(head(m, st, cm, B * Lq, q.h16, m->vision_proj, q.proj, ID_MAP));
    return sprc_l2norm_rows(q.proj, m->embed_dim, feats, feats16, m->embed_dim, B * Lq, m->embed_dim, dt, st);
}

// kv != nullptr: the K|V projections of the reference images are GIVEN (rows of an sprc_qformer_encode_kv output, query b reads row
// kv_index[b], or b when kv_index is null) and ref_embeds is not read
static int qformer_fuse_impl(const sprc_qformer_model* m, const float* ref_embeds, int32_t enc_tokens,
                             const int64_t* input_ids, const int64_t* attention_mask, int32_t B, float* fusion,
                             void* fusion16, const float* prompt_tokens, float* loss_align, void* ws, size_t ws_bytes, sprc_stream s,
                             const void* kv = nullptr, const int32_t* kv_index = nullptr) {
    RUN(check_qf(m));
    SPRC_REQUIRE((ref_embeds || kv) && input_ids && attention_mask && fusion && ws && B > 0, "sprc_qformer_fuse: bad arguments");
    SPRC_REQUIRE(enc_tokens == 257, "sprc_qformer_fuse: enc_tokens=%d (sprc_qformer_workspace_bytes plans for 257)", enc_tokens);
    SPRC_REQUIRE(((uintptr_t)ws % 256) == 0, "sprc_qformer_fuse: workspace must be 256-byte aligned");
    Bump b(ws, ws_bytes);
    QfBufs q;
    qf_plan(m, B, enc_tokens, b, q);
    if (!b.ok) {
        set_error("sprc_qformer_fuse: workspace too small (%zu bytes given)", ws_bytes);
        return SPRC_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)s;
    const int dt = m->dtype, Hd = m->hidden, Lq = m->num_query, Lt = m->max_txt, S = Lq + Lt;
    const int cm = m->x3_fuse;
    if (kv == nullptr) RUN(qf_encode_kv(m, st, q.enc, q.kv, ref_embeds, B, enc_tokens, cm));
    const KvSrc kvs{kv != nullptr ? kv : q.kv, enc_tokens, kv != nullptr ? kv_index : nullptr, nullptr, 0, nullptr, (int64_t)m->n_cross * 2 * Hd};
    RUN(sprc_qformer_mask(attention_mask, q.mask, B, Lq, Lt, st));
    sprc_qformer_embed_args e;
    memset(&e, 0, sizeof(e));
    e.B = B; e.Lq = Lq; e.Lt = Lt; e.hidden = Hd; e.out_dtype = x3_layouts(cm).ln ? SPRC_F16X3 : dt; e.vocab = m->vocab;
    e.input_ids = input_ids; e.word_emb = m->word_emb; e.pos_emb = m->pos_emb;
    e.gamma = m->emb_ln_w; e.beta = m->emb_ln_b; e.eps = m->ln_eps;
    // pass 1: learned query tokens + text, cross-attention to the reference image (align_prompt.py:332-339)
    e.query_embeds = m->query_tokens; e.q_bstride = 0;
    e.y32 = q.h32; e.y16 = q.h16;
    RUN(sprc_qformer_embed(&e, st));
    RUN(qf_stack(m, st, q, B, S, &kvs, q.mask, q.h32, q.h16, cm, 0, Lq));       // read: the query rows
    if (loss_align != nullptr)                  // training: mse(mean fused query token, mean prompt token)  (align_prompt.py:192-193)
        RUN(sprc_align_mse(q.h32, (int64_t)S * Hd, Lq, Hd, prompt_tokens, B, loss_align, st));
    // pass 2: pass-1 query rows as query_embeds (re-LayerNormed by the embedding LN), no image (:341-346)
    e.query_embeds = q.h32; e.q_bstride = (int64_t)S * Hd;
    e.y32 = q.g32; e.y16 = q.g16;
    RUN(sprc_qformer_embed(&e, st));
    RUN(qf_stack(m, st, q, B, S, nullptr, q.mask, q.g32, q.g16, cm, Lq, 1));     // read: the [CLS] row
    // fusion = normalize(text_proj(pass2[:, 32, :]))  (:348-350): row Lq of every sample
    const sprc_rowmap cls_row = {1, S, Lq};
    RUN(head(m, st, cm, B, q.g16, m->text_proj, q.proj, cls_row));
    return sprc_l2norm_rows(q.proj, m->embed_dim, fusion, fusion16, m->embed_dim, B, m->embed_dim, dt, st);
}

extern "C" int sprc_qformer_fuse_kv(const sprc_qformer_model* m, const void* kv, int32_t enc_tokens, const int32_t* kv_index,
                                    const int64_t* input_ids, const int64_t* attention_mask, int32_t B, float* fusion,
                                    void* fusion16, void* ws, size_t ws_bytes, sprc_stream s) {
    SPRC_REQUIRE(kv != nullptr && ((uintptr_t)kv % 16) == 0, "sprc_qformer_fuse_kv: kv must be a 16-byte aligned sprc_qformer_encode_kv output");
    SPRC_REQUIRE(m && is16(m->dtype), "sprc_qformer_fuse_kv: a 16-bit model (the fp32 engine recomputes its projections)");
    return qformer_fuse_impl(m, nullptr, enc_tokens, input_ids, attention_mask, B, fusion, fusion16, nullptr, nullptr, ws, ws_bytes, s, kv, kv_index);
}

extern "C" int sprc_qformer_fuse(const sprc_qformer_model* m, const float* ref_embeds, int32_t enc_tokens,
                                 const int64_t* input_ids, const int64_t* attention_mask, int32_t B, float* fusion,
                                 void* fusion16, void* ws, size_t ws_bytes, sprc_stream s) {
    return qformer_fuse_impl(m, ref_embeds, enc_tokens, input_ids, attention_mask, B, fusion, fusion16, nullptr, nullptr, ws, ws_bytes, s);
}

// ---- training forward: align_prompt.py:95-200, forward only ---------------------------------------
extern "C" int sprc_qformer_fuse_train(const sprc_qformer_model* m, const float* ref_embeds, int32_t enc_tokens,
                                       const int64_t* input_ids, const int64_t* attention_mask, int32_t B, float* fusion,
                                       void* fusion16, const float* prompt_tokens, float* loss_align, void* ws, size_t ws_bytes,
                                       sprc_stream s) {
    SPRC_REQUIRE(prompt_tokens && loss_align, "sprc_qformer_fuse_train: prompt_tokens / loss_align missing");
    return qformer_fuse_impl(m, ref_embeds, enc_tokens, input_ids, attention_mask, B, fusion, fusion16, prompt_tokens, loss_align, ws,
                             ws_bytes, s);
}

extern "C" int sprc_qformer_text_only(const sprc_qformer_model* m, const float* prompt_tokens, const int64_t* input_ids,
                                      const int64_t* attention_mask, int32_t B, float* feat, void* feat16, void* ws,
                                      size_t ws_bytes, sprc_stream s) {
    RUN(check_qf(m));
    SPRC_REQUIRE(prompt_tokens && input_ids && attention_mask && feat && ws && B > 0, "sprc_qformer_text_only: bad arguments");
    SPRC_REQUIRE(((uintptr_t)ws % 256) == 0, "sprc_qformer_text_only: workspace must be 256-byte aligned");
    Bump b(ws, ws_bytes);
    QfBufs q;
    qf_plan(m, B, 0, b, q, false);
    if (!b.ok) {
        set_error("sprc_qformer_text_only: workspace too small (%zu bytes given)", ws_bytes);
        return SPRC_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)s;
    const int dt = m->dtype, Hd = m->hidden, Lq = m->num_query, Lt = m->max_txt, S = Lq + Lt;
    RUN(sprc_qformer_mask(attention_mask, q.mask, B, Lq, Lt, st));          // cat([ones(32), text mask]) as the reference passes it (:174-176)
    sprc_qformer_embed_args e;
    memset(&e, 0, sizeof(e));
    const int cm = m->x3_fuse;
    e.B = B; e.Lq = Lq; e.Lt = Lt; e.hidden = Hd; e.out_dtype = x3_layouts(cm).ln ? SPRC_F16X3 : dt; e.vocab = m->vocab; e.no_img = 1;
    e.input_ids = input_ids; e.word_emb = m->word_emb; e.pos_emb = m->pos_emb;
    e.gamma = m->emb_ln_w; e.beta = m->emb_ln_b; e.eps = m->ln_eps;
    e.query_embeds = prompt_tokens; e.q_bstride = 0;
    e.y32 = q.h32; e.y16 = q.h16;
    RUN(sprc_qformer_embed(&e, st));
    RUN(qf_stack(m, st, q, B, S, nullptr, q.mask, q.h32, q.h16, cm, 0, 1));
    const sprc_rowmap row0 = {1, S, 0};                                      // last_hidden_state[:, 0, :]  (:177-179)
    RUN(head(m, st, cm, B, q.h16, m->text_proj, q.proj, row0));
    return sprc_l2norm_rows(q.proj, m->embed_dim, feat, feat16, m->embed_dim, B, m->embed_dim, dt, st);
}

extern "C" int sprc_qformer_text(const sprc_qformer_model* m, const int64_t* input_ids, const int64_t* attention_mask, int32_t B,
                                 float* feat, void* feat16, void* ws, size_t ws_bytes, sprc_stream s) {
    RUN(check_qf(m));
    SPRC_REQUIRE(input_ids && attention_mask && feat && ws && B > 0, "sprc_qformer_text: bad arguments");
    SPRC_REQUIRE(((uintptr_t)ws % 256) == 0, "sprc_qformer_text: workspace must be 256-byte aligned");
    Bump b(ws, ws_bytes);
    QfBufs q;
    qf_plan(m, B, 0, b, q, false);
    if (!b.ok) {
        set_error("sprc_qformer_text: workspace too small (%zu bytes given)", ws_bytes);
        return SPRC_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)s;
    const int dt = m->dtype, Hd = m->hidden, Lt = m->max_txt, cm = m->x3_fuse;
    RUN(sprc_qformer_mask(attention_mask, q.mask, B, 0, Lt, st));              // (1 - mask) * -10000 over the Lt text keys (Qformer.py:806-807)
    sprc_qformer_embed_args e;
    memset(&e, 0, sizeof(e));
    e.B = B; e.Lq = 0; e.Lt = Lt; e.hidden = Hd; e.out_dtype = x3_layouts(cm).ln ? SPRC_F16X3 : dt; e.vocab = m->vocab;
    e.input_ids = input_ids; e.word_emb = m->word_emb; e.pos_emb = m->pos_emb;
    e.gamma = m->emb_ln_w; e.beta = m->emb_ln_b; e.eps = m->ln_eps;
    e.y32 = q.h32; e.y16 = q.h16;
    RUN(sprc_qformer_embed(&e, st));
    RUN(qf_stack(m, st, q, B, Lt, nullptr, q.mask, q.h32, q.h16, cm, 0, 1));
    const sprc_rowmap row0 = {1, Lt, 0};                                     // last_hidden_state[:, 0, :]  (rerank.py:388-390)
    RUN(head(m, st, cm, B, q.h16, m->text_proj, q.proj, row0));
    return sprc_l2norm_rows(q.proj, m->embed_dim, feat, feat16, m->embed_dim, B, m->embed_dim, dt, st);
}

// ---- stage-2 rerank: blip2_qformer_cir_rerank.py:399-445 ------------------------------------------
extern "C" size_t sprc_qformer_kv_workspace_bytes(const sprc_qformer_model* m, int32_t B, int32_t tokens) {
    if (!m || B <= 0 || tokens <= 0) return 0;
    return (is16(m->dtype) ? (size_t)B * tokens * m->enc_width * 2 * ((m->x3 & SPRC_X3_CKV) ? 2 : 1) : 0) + 512;
}

extern "C" int sprc_qformer_encode_kv(const sprc_qformer_model* m, const float* raw, int32_t B, int32_t tokens, void* kv,
                                      void* ws, size_t ws_bytes, sprc_stream s) {
    RUN(check_qf(m));
    SPRC_REQUIRE(raw && kv && B > 0 && tokens > 0, "sprc_qformer_encode_kv: bad arguments");
    SPRC_REQUIRE(!is16(m->dtype) || (ws && ((uintptr_t)ws % 256) == 0 && ws_bytes >= sprc_qformer_kv_workspace_bytes(m, B, tokens) - 512),
                 "sprc_qformer_encode_kv: workspace too small or misaligned");
    SPRC_REQUIRE(((uintptr_t)kv % 16) == 0, "sprc_qformer_encode_kv: kv must be 16-byte aligned");
    return qf_encode_kv(m, (hipStream_t)s, ws, kv, raw, B, tokens, m->x3_fuse);
}

extern "C" size_t sprc_qformer_itm_workspace_bytes(const sprc_qformer_model* m, int32_t P) {
    if (!m || P <= 0) return 0;
    Bump b(nullptr, 0);
    QfBufs q;
    return qf_plan(m, P, 0, b, q, false) + 256;
}

extern "C" int sprc_qformer_itm(const sprc_qformer_model* m, const float* itm_w, const float* itm_b, const void* kv_a,
                                int32_t tokens_a, const int32_t* index_a, const void* kv_b, int32_t tokens_b,
                                const int32_t* index_b, const int64_t* input_ids, const int64_t* attention_mask, int32_t P,
                                float* prob, void* ws, size_t ws_bytes, sprc_stream s) {
    RUN(check_qf(m));
    SPRC_REQUIRE(itm_w && itm_b && kv_a && kv_b && input_ids && attention_mask && prob && ws && P > 0, "sprc_qformer_itm: bad arguments");
    SPRC_REQUIRE(tokens_a > 0 && tokens_b > 0, "sprc_qformer_itm: empty key segment");
    SPRC_REQUIRE(((uintptr_t)ws % 256) == 0, "sprc_qformer_itm: workspace must be 256-byte aligned");
    Bump b(ws, ws_bytes);
    QfBufs q;
    qf_plan(m, P, 0, b, q, false);
    if (!b.ok) {
        set_error("sprc_qformer_itm: workspace too small (%zu bytes given)", ws_bytes);
        return SPRC_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)s;
    const int dt = m->dtype, Hd = m->hidden, Lq = m->num_query, Lt = m->max_txt, S = Lq + Lt;
    RUN(sprc_qformer_mask(attention_mask, q.mask, P, Lq, Lt, st));
    sprc_qformer_embed_args e;
    memset(&e, 0, sizeof(e));
    const int cm = m->x3_fuse;
    e.B = P; e.Lq = Lq; e.Lt = Lt; e.hidden = Hd; e.out_dtype = x3_layouts(cm).ln ? SPRC_F16X3 : dt; e.vocab = m->vocab;
    e.input_ids = input_ids; e.word_emb = m->word_emb; e.pos_emb = m->pos_emb;
    e.gamma = m->emb_ln_w; e.beta = m->emb_ln_b; e.eps = m->ln_eps;
    e.query_embeds = m->query_tokens; e.q_bstride = 0;
    e.y32 = q.h32; e.y16 = q.h16;
    RUN(sprc_qformer_embed(&e, st));
    // one Q-Former pass in call shape (ii) over cat(reference, candidate) tokens (:430-437)
    const KvSrc kvs{kv_a, tokens_a, index_a, kv_b, tokens_b, index_b, (int64_t)m->n_cross * 2 * Hd};
    RUN(qf_stack(m, st, q, P, S, &kvs, q.mask, q.h32, q.h16, cm, 0, Lq));      // read: the query rows (itm_head)
    // itm_head on the query rows, mean over them, softmax, P(match)  (:439-445)
    return sprc_itm_head(q.h32, (int64_t)S * Hd, Lq, Hd, itm_w, itm_b, P, prob, st);
}
