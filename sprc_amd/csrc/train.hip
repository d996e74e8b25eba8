// train.hip -- backward kernels of the training step: Blip2QformerCirAlignPrompt.forward
// (lavis/models/blip2_models/blip2_qformer_cir_align_prompt.py:95-200) under blip_fine_tune_2.py:293-304.
// The ViT is frozen there (align_prompt.py:64-69), so what trains is the Q-Former, ln_vision, the two ITC heads, the query /
// prompt tokens and temp.  Everything here is fp32: the products run on the exact-fp32 MFMA GEMM (sprc_gemm) through transposed
// operand copies (dX = dY . W needs W^T K-contiguous, dW = dY^T . X needs dY^T and X^T), the rest are row / element kernels:
//   sprc_transpose_f32, sprc_colsum_f32, sprc_gelu_fwd/bwd, sprc_layernorm_bwd, sprc_attention_bwd, sprc_qformer_embed_rows/bwd,
//   sprc_sim_max_bwd, sprc_contrastive_ce_bwd, sprc_l2norm_bwd, sprc_align_mse_bwd.
// Reductions have a FIXED order (no floating-point atomics) except the two embedding-table scatters, whose colliding rows (the
// padding token, repeated words) are summed by atomicAdd.
#include "common.hpp"

namespace sprc {

constexpr int TR_TILE = 32;

// dst[c, r] = src[r, c]
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, int64_t ld_src, float* __restrict__ dst,
                                                            int64_t ld_dst, int rows, int cols) {
    __shared__ float tile[TR_TILE][TR_TILE + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8
    const int r0 = blockIdx.y * TR_TILE, c0 = blockIdx.x * TR_TILE;
#pragma unroll
    for (int i = 0; i < TR_TILE; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        tile[ty + i][tx] = (r < rows && c < cols) ? src[(int64_t)r * ld_src + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TR_TILE; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (c < cols && r < rows) dst[(int64_t)c * ld_dst + r] = tile[tx][ty + i];
    }
}

// out[n] (+)= sum_m x[m, n].  One workgroup per 32 columns (a 128-B line per row), 8 row lanes x 32 columns, every thread keeps 4 independent
// partial sums over its rows (m = ty, ty + 8, ...), combined in a fixed order: deterministic.  (The first form -- one workgroup per 64 columns,
// 4 waves striding ALL rows one load at a time -- ran 12 workgroups for N = 768: 128 us per call on average, 46 ms of a 257-ms training step,
// profiles/r05_train_step_kernel_stats_before.csv.)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld, int M, int N, float* out, int accumulate) {
    __shared__ float part[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + tx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (n < N) {
        const float* col = x + n;
        int m = ty;
        for (; m + 24 < M; m += 32) {
            a0 += col[(int64_t)m * ld]; a1 += col[(int64_t)(m + 8) * ld]; a2 += col[(int64_t)(m + 16) * ld]; a3 += col[(int64_t)(m + 24) * ld];
        }
        for (; m < M; m += 8) a0 += col[(int64_t)m * ld];
    }
    part[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && n < N) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += part[r][tx];
        out[n] = accumulate ? out[n] + s : s;
    }
}

// dst[c, r] = (16-bit) src[r, c]: the transposed OPERAND COPY of the 16-bit training products (dW = dY^T . X reduces over the rows of dY and X)
template <bool F16>
__global__ __launch_bounds__(256) void transpose_f32_to16_kernel(const float* __restrict__ src, int64_t ld_src, uint16_t* __restrict__ dst,
                                                                 int64_t ld_dst, int rows, int cols) {
    __shared__ float tile[TR_TILE][TR_TILE + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8
    const int r0 = blockIdx.y * TR_TILE, c0 = blockIdx.x * TR_TILE;
#pragma unroll
    for (int i = 0; i < TR_TILE; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        tile[ty + i][tx] = (r < rows && c < cols) ? src[(int64_t)r * ld_src + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TR_TILE; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (c < cols && r < rows) {
            const float v = tile[tx][ty + i];
            dst[(int64_t)c * ld_dst + r] = F16 ? __builtin_bit_cast(uint16_t, (_Float16)v) : f32_to_bf16_bits(v);
        }
    }
}

__device__ __forceinline__ float gelu_grad(float x) {                      // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}
__global__ void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = gelu_erf(x[i]);
}
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dx[i] = dy[i] * gelu_grad(x[i]);
}

// ---- LayerNorm backward: one wave per row (the row in registers), D <= 2048 ----
constexpr int LNB_MAXC = 8, LNB_ROWS = 4, LNB_RPW = 8;      // 4 waves x 8 rows each per block: 32 rows share one partial record
struct LnBwdParams {
    const float* x; int64_t ldx; const float* gamma; const float* dy; int64_t lddy; float eps; int M, D;
    float* dx; int64_t lddx; float* part;      // part: [blocks][2][D] per-block partial sums of (dy * xhat, dy)
};
__global__ __launch_bounds__(64 * LNB_ROWS) void layernorm_bwd_kernel(LnBwdParams p) {
    extern __shared__ float sh[];               // [LNB_ROWS][2][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = p.D >> 2;
    float* mine = sh + (size_t)wave * 2 * p.D;
    float4 pg[LNB_MAXC], pb[LNB_MAXC];          // this wave's running contributions to dgamma / dbeta over its LNB_RPW rows (row order)
#pragma unroll
    for (int c = 0; c < LNB_MAXC; ++c) pg[c] = pb[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < LNB_RPW; ++j) {
        const int row = (blockIdx.x * LNB_ROWS + wave) * LNB_RPW + j;
        if (row >= p.M) break;
        float4 xv[LNB_MAXC], gv[LNB_MAXC];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < LNB_MAXC; ++c) {
            const int i = lane + c * 64;
            xv[c] = i < nch ? reinterpret_cast<const float4*>(p.x + (int64_t)row * p.ldx)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            gv[c] = i < nch ? reinterpret_cast<const float4*>(p.dy + (int64_t)row * p.lddy)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (xv[c].x + xv[c].y) + (xv[c].z + xv[c].w);
        }
        const float mean = wave_sum(s) / (float)p.D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < LNB_MAXC; ++c)
            if (lane + c * 64 < nch) {
                const float a = xv[c].x - mean, b = xv[c].y - mean, cc = xv[c].z - mean, d = xv[c].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        const float rstd = rsqrtf(wave_sum(q) / (float)p.D + p.eps);
        // xhat in xv, g = dy * gamma; s1 = mean(g), s2 = mean(g * xhat)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < LNB_MAXC; ++c) {
            const int i = lane + c * 64;
            if (i < nch) {
                const float4 gm = reinterpret_cast<const float4*>(p.gamma)[i];
                xv[c].x = (xv[c].x - mean) * rstd; xv[c].y = (xv[c].y - mean) * rstd;
                xv[c].z = (xv[c].z - mean) * rstd; xv[c].w = (xv[c].w - mean) * rstd;
                // per-row contributions to dgamma / dbeta
                pg[c].x += gv[c].x * xv[c].x; pg[c].y += gv[c].y * xv[c].y; pg[c].z += gv[c].z * xv[c].z; pg[c].w += gv[c].w * xv[c].w;
                pb[c].x += gv[c].x; pb[c].y += gv[c].y; pb[c].z += gv[c].z; pb[c].w += gv[c].w;
                gv[c].x *= gm.x; gv[c].y *= gm.y; gv[c].z *= gm.z; gv[c].w *= gm.w;
                s1 += (gv[c].x + gv[c].y) + (gv[c].z + gv[c].w);
                s2 += (gv[c].x * xv[c].x + gv[c].y * xv[c].y) + (gv[c].z * xv[c].z + gv[c].w * xv[c].w);
            }
        }
        s1 = wave_sum(s1) / (float)p.D;
        s2 = wave_sum(s2) / (float)p.D;
        if (p.dx != nullptr) {
#pragma unroll
            for (int c = 0; c < LNB_MAXC; ++c) {
                const int i = lane + c * 64;
                if (i < nch)
                    reinterpret_cast<float4*>(p.dx + (int64_t)row * p.lddx)[i] =
                        make_float4(rstd * (gv[c].x - s1 - xv[c].x * s2), rstd * (gv[c].y - s1 - xv[c].y * s2),
                                    rstd * (gv[c].z - s1 - xv[c].z * s2), rstd * (gv[c].w - s1 - xv[c].w * s2));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < LNB_MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nch) {
            reinterpret_cast<float4*>(mine)[i] = pg[c];
            reinterpret_cast<float4*>(mine + p.D)[i] = pb[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * p.D; i += 64 * LNB_ROWS) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < LNB_ROWS; ++w) s += sh[(size_t)w * 2 * p.D + i];
        p.part[(size_t)blockIdx.x * 2 * p.D + i] = s;
    }
}
// dgamma / dbeta (+)= sum over the per-block partials, in block order
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblocks, int D, float* dgamma, float* dbeta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * D) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * 2 * D + i];
    if (i < D) dgamma[i] += s;
    else dbeta[i - D] += s;
}

// ---- attention backward (fp32, head_dim == 64 = one lane per feature) ----
// token t of batch b, head h lives at ptr + (b * T + t) * ld + h * 64 (the forward's layout).  Scratch P / dS [B, H, Tq, Tk].
struct AttnBwdParams {
    int B, H, Tq, Tk;
    const float *q, *k, *v, *dout; int64_t ldq, ldk, ldv, lddo;
    const float* key_mask; float scale;
    float *dq, *dk, *dv; int64_t lddq, lddk, lddv;
    float *P, *dS;
    uint32_t drop_thresh, drop_site; uint64_t drop_seed; float drop_scale;
};
constexpr int AB_MAXK = 9;                      // keys per lane -> Tk <= 576
// one wave per (b, h, query row): P row, dS row, dQ row
__global__ __launch_bounds__(256) void attn_bwd_rows_kernel(AttnBwdParams p) {
    __shared__ float sq[4][64], sdo[4][64], sds[4][64 * AB_MAXK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    const int64_t total = (int64_t)p.B * p.H * p.Tq;
    const bool live = r < total;
    const int64_t rr = live ? r : total - 1;
    const int i = (int)(rr % p.Tq), h = (int)((rr / p.Tq) % p.H), b = (int)(rr / ((int64_t)p.Tq * p.H));
    sq[wave][lane] = p.q[((int64_t)b * p.Tq + i) * p.ldq + h * 64 + lane];
    sdo[wave][lane] = p.dout[((int64_t)b * p.Tq + i) * p.lddo + h * 64 + lane];
    __syncthreads();
    float s[AB_MAXK], dp[AB_MAXK], dmask[AB_MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < AB_MAXK; ++c) {
        const int key = lane + c * 64;
        s[c] = -INFINITY; dp[c] = 0.f;
        if (key < p.Tk) {
            const float* kr = p.k + ((int64_t)b * p.Tk + key) * p.ldk + h * 64;
            const float* vr = p.v + ((int64_t)b * p.Tk + key) * p.ldv + h * 64;
            float a = 0.f, d = 0.f;
            for (int e = 0; e < 64; e += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(kr + e), vv = *reinterpret_cast<const float4*>(vr + e);
                a = fmaf(sq[wave][e], kv.x, a); a = fmaf(sq[wave][e + 1], kv.y, a); a = fmaf(sq[wave][e + 2], kv.z, a); a = fmaf(sq[wave][e + 3], kv.w, a);
                d = fmaf(sdo[wave][e], vv.x, d); d = fmaf(sdo[wave][e + 1], vv.y, d); d = fmaf(sdo[wave][e + 2], vv.z, d); d = fmaf(sdo[wave][e + 3], vv.w, d);
            }
            s[c] = a * p.scale + (p.key_mask ? p.key_mask[(int64_t)b * p.Tk + key] : 0.f);
            dp[c] = d;                                   // dL/d(dropped probability)
            mx = fmaxf(mx, s[c]);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < AB_MAXK; ++c)
        if (lane + c * 64 < p.Tk) { s[c] = expf(s[c] - mx); sum += s[c]; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < AB_MAXK; ++c)
        if (lane + c * 64 < p.Tk) {
            s[c] *= inv;
            // out = (D o P) V: dL/dP = D o (dO V^T); the softmax backward below runs on the UNdropped P
            const float dm = (p.drop_thresh == 0 || drop_keep(p.drop_seed, p.drop_site, (uint64_t)rr * p.Tk + (lane + c * 64), p.drop_thresh)) ? p.drop_scale : 0.f;
            dp[c] *= dm;
            dot += s[c] * dp[c];
            dmask[c] = dm;
        }
    dot = wave_sum(dot);
    float* Prow = p.P + rr * p.Tk;
    float* dSrow = p.dS + rr * p.Tk;
#pragma unroll
    for (int c = 0; c < AB_MAXK; ++c) {
        const int key = lane + c * 64;
        if (key < p.Tk) {
            const float ds = s[c] * (dp[c] - dot);
            sds[wave][key] = ds;
            if (live) { Prow[key] = s[c] * dmask[c]; dSrow[key] = ds; }     // the keys kernel needs D o P (dV = (D o P)^T dO)
        }
    }
    __syncthreads();
    // dq[d] = scale * sum_k dS[k] K[k, d]   (lane = d)
    float acc = 0.f;
    for (int key = 0; key < p.Tk; ++key) acc = fmaf(sds[wave][key], p.k[((int64_t)b * p.Tk + key) * p.ldk + h * 64 + lane], acc);
    if (live) p.dq[((int64_t)b * p.Tq + i) * p.lddq + h * 64 + lane] = acc * p.scale;
}
// one wave per (b, h, key): dK = scale * dS^T Q, dV = P^T dO   (lane = d)
__global__ __launch_bounds__(256) void attn_bwd_keys_kernel(AttnBwdParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= (int64_t)p.B * p.H * p.Tk) return;
    const int key = (int)(r % p.Tk), h = (int)((r / p.Tk) % p.H), b = (int)(r / ((int64_t)p.Tk * p.H));
    const float* Pc = p.P + ((int64_t)b * p.H + h) * p.Tq * p.Tk + key;
    const float* dSc = p.dS + ((int64_t)b * p.H + h) * p.Tq * p.Tk + key;
    float dk = 0.f, dv = 0.f;
    for (int i = 0; i < p.Tq; ++i) {
        dk = fmaf(dSc[(int64_t)i * p.Tk], p.q[((int64_t)b * p.Tq + i) * p.ldq + h * 64 + lane], dk);
        dv = fmaf(Pc[(int64_t)i * p.Tk], p.dout[((int64_t)b * p.Tq + i) * p.lddo + h * 64 + lane], dv);
    }
    p.dk[((int64_t)b * p.Tk + key) * p.lddk + h * 64 + lane] = dk * p.scale;
    p.dv[((int64_t)b * p.Tk + key) * p.lddv + h * 64 + lane] = dv;
}

// ---- Q-Former embeddings without the LayerNorm (pre-LN rows) and their backward ----
__global__ __launch_bounds__(256) void embed_rows_kernel(sprc_qformer_embed_args p, float* __restrict__ pre) {
    const int S = p.Lq + p.Lt, H4 = p.hidden >> 2;
    const int64_t total = (int64_t)p.B * S * H4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % H4);
        const int64_t row = e / H4;
        const int b = (int)(row / S), t = (int)(row % S);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int pos = -1;
        bool word = false;
        int64_t id = 0;
        if (p.no_img) {
            pos = t;
            if (t >= 1 && t <= p.Lq) v = reinterpret_cast<const float4*>(p.query_embeds + (int64_t)b * p.q_bstride + (int64_t)(t - 1) * p.hidden)[c];
            else { word = true; id = p.input_ids[(int64_t)b * p.Lt + (t == 0 ? 0 : t - p.Lq)]; }
        } else if (t < p.Lq) {
            v = reinterpret_cast<const float4*>(p.query_embeds + (int64_t)b * p.q_bstride + (int64_t)t * p.hidden)[c];
        } else {
            pos = t - p.Lq;
            word = true;
            id = p.input_ids[(int64_t)b * p.Lt + pos];
        }
        if (word) {
            id = id < 0 ? 0 : (id >= p.vocab ? p.vocab - 1 : id);
            v = reinterpret_cast<const float4*>(p.word_emb + id * p.hidden)[c];
        }
        if (pos >= 0) {
            const float4 pe = reinterpret_cast<const float4*>(p.pos_emb + (int64_t)pos * p.hidden)[c];
            v.x += pe.x; v.y += pe.y; v.z += pe.z; v.w += pe.w;
        }
        reinterpret_cast<float4*>(pre)[e] = v;
    }
}
// d_pre [B, S, hidden] -> d_query (+= over the batch when q_bstride == 0: atomic), d_word_emb[id] +=, d_pos_emb[pos] += (atomic)
__global__ __launch_bounds__(256) void embed_bwd_kernel(sprc_qformer_embed_args p, const float* __restrict__ dpre, float* dquery,
                                                        int64_t dq_bstride, float* dword, float* dpos) {
    const int S = p.Lq + p.Lt;
    const int64_t total = (int64_t)p.B * S * p.hidden;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int d = (int)(e % p.hidden);
        const int64_t row = e / p.hidden;
        const int b = (int)(row / S), t = (int)(row % S);
        const float g = dpre[e];
        int pos = -1, qrow = -1;
        bool word = false;
        int64_t id = 0;
        if (p.no_img) {
            pos = t;
            if (t >= 1 && t <= p.Lq) qrow = t - 1;
            else { word = true; id = p.input_ids[(int64_t)b * p.Lt + (t == 0 ? 0 : t - p.Lq)]; }
        } else if (t < p.Lq) {
            qrow = t;
        } else {
            pos = t - p.Lq;
            word = true;
            id = p.input_ids[(int64_t)b * p.Lt + pos];
        }
        if (qrow >= 0 && dquery != nullptr) atomicAdd(dquery + (int64_t)b * dq_bstride + (int64_t)qrow * p.hidden + d, g);
        if (word) atomicAdd(dword + (id < 0 ? 0 : (id >= p.vocab ? p.vocab - 1 : id)) * p.hidden + d, g);
        if (pos >= 0) atomicAdd(dpos + (int64_t)pos * p.hidden + d, g);
    }
}

// ---- heads and losses ----
// jstar[b, n] = first argmax_j <fusion[b], feats[n, j]>   (torch.max over the last dim, align_prompt.py:161)
__global__ __launch_bounds__(256) void sim_argmax_kernel(const float* __restrict__ fusion, const float* __restrict__ feats, int B, int N, int J,
                                                         int E, int* __restrict__ jstar) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= (int64_t)B * N) return;
    const int b = (int)(r / N), n = (int)(r % N);
    float best = -INFINITY;
    int bj = 0;
    for (int j = 0; j < J; ++j) {
        float a = 0.f;
        for (int e = lane; e < E; e += 64) a = fmaf(fusion[(int64_t)b * E + e], feats[((int64_t)n * J + j) * E + e], a);
        a = wave_sum(a);
        if (a > best) { best = a; bj = j; }
    }
    if (lane == 0) jstar[r] = bj;
}
__global__ __launch_bounds__(256) void sim_bwd_fusion_kernel(const float* __restrict__ feats, const float* __restrict__ dsim, const int* __restrict__ jstar,
                                                             int B, int N, int J, int E, float* dfusion) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < E; e += 256) {
        float a = 0.f;
        for (int n = 0; n < N; ++n) a = fmaf(dsim[(int64_t)b * N + n], feats[((int64_t)n * J + jstar[(int64_t)b * N + n]) * E + e], a);
        dfusion[(int64_t)b * E + e] += a;
    }
}
__global__ __launch_bounds__(256) void sim_bwd_feats_kernel(const float* __restrict__ fusion, const float* __restrict__ dsim, const int* __restrict__ jstar,
                                                            int B, int N, int J, int E, float* dfeats) {
    const int n = blockIdx.x / J, j = blockIdx.x % J;
    for (int e = threadIdx.x; e < E; e += 256) {
        float a = 0.f;
        for (int b = 0; b < B; ++b)
            if (jstar[(int64_t)b * N + n] == j) a = fmaf(dsim[(int64_t)b * N + n], fusion[(int64_t)b * E + e], a);
        dfeats[((int64_t)n * J + j) * E + e] += a;
    }
}
// loss = mean_b CE(sim[b, :] / temp, b): dsim = g / B * (softmax - onehot) / temp; dtemp += -g / B * sum (softmax - onehot) sim / temp^2
__global__ __launch_bounds__(64) void ce_bwd_kernel(const float* __restrict__ sim, int64_t ld, int B, float temp, float g, float* dsim, float* dtemp) {
    const int lane = threadIdx.x;
    float tacc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* row = sim + (int64_t)b * ld;
        float mx = -INFINITY;
        for (int n = lane; n < B; n += 64) mx = fmaxf(mx, row[n] / temp);
        mx = wave_max(mx);
        float se = 0.f;
        for (int n = lane; n < B; n += 64) se += expf(row[n] / temp - mx);
        se = wave_sum(se);
        for (int n = lane; n < B; n += 64) {
            const float pr = expf(row[n] / temp - mx) / se - (n == b ? 1.0f : 0.0f);
            const float gz = g / (float)B * pr;                     // d loss / d (sim / temp)
            dsim[(int64_t)b * B + n] = gz / temp;
            tacc -= gz * row[n] / (temp * temp);
        }
    }
    tacc = wave_sum(tacc);
    if (lane == 0 && dtemp != nullptr) dtemp[0] += tacc;
}
// y = x / max(||x||, 1e-12): dx = (dy - y <y, dy>) / max(||x||, 1e-12)   (one wave per row, D <= 2048)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy, int64_t lddy,
                                                         float* __restrict__ dx, int64_t lddx, int M, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    const float* xr = x + (int64_t)row * ldx;
    const float* gr = dy + (int64_t)row * lddy;
    float q = 0.f, d = 0.f;
    for (int e = lane; e < D; e += 64) { q = fmaf(xr[e], xr[e], q); d = fmaf(xr[e], gr[e], d); }
    q = wave_sum(q); d = wave_sum(d);
    const float nrm = fmaxf(sqrtf(q), 1e-12f), inv = 1.0f / nrm;
    for (int e = lane; e < D; e += 64) dx[(int64_t)row * lddx + e] = (gr[e] - xr[e] * inv * (d * inv)) * inv;
}
// loss = mse(mean_j h[b, j, :], mean_j prompt[j, :]): dh[b, j, d] += g * 2 / (B D Lq) * (mean_j h - mean_j prompt)[d]
__global__ __launch_bounds__(256) void align_mse_bwd_kernel(const float* __restrict__ h, int64_t sample_stride, int Lq, int D,
                                                            const float* __restrict__ prompt, int B, float g, float* dh, int64_t d_stride) {
    const int b = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) {
        float hm = 0.f, pm = 0.f;
        for (int j = 0; j < Lq; ++j) { hm += h[(int64_t)b * sample_stride + (int64_t)j * D + d]; pm += prompt[(int64_t)j * D + d]; }
        const float gd = g * 2.0f / ((float)B * (float)D * (float)Lq) * ((hm - pm) / (float)Lq);
        for (int j = 0; j < Lq; ++j) dh[(int64_t)b * d_stride + (int64_t)j * D + d] += gd;
    }
}

static int grid1d(int64_t n, int block = 256, int cap = 256 * 16) {
    const int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace sprc

using namespace sprc;

extern "C" int sprc_transpose_f32(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int32_t rows, int32_t cols, sprc_stream s) {
    SPRC_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "sprc_transpose_f32: bad arguments");
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)s, src, ld_src, dst, ld_dst, rows, cols);
    SPRC_CHECK_LAUNCH("sprc_transpose_f32");
    return SPRC_OK;
}

extern "C" int sprc_transpose_f32_to16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int32_t rows, int32_t cols, int32_t dtype,
                                       sprc_stream s) {
    SPRC_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "sprc_transpose_f32_to16: bad arguments");
    SPRC_REQUIRE(is16(dtype), "sprc_transpose_f32_to16: dtype %d is not a 16-bit type", dtype);
    const dim3 grid((cols + TR_TILE - 1) / TR_TILE, (rows + TR_TILE - 1) / TR_TILE);
    if (dtype == SPRC_F16) hipLaunchKernelGGL(transpose_f32_to16_kernel<true>, grid, dim3(256), 0, (hipStream_t)s, src, ld_src, (uint16_t*)dst, ld_dst, rows, cols);
    else hipLaunchKernelGGL(transpose_f32_to16_kernel<false>, grid, dim3(256), 0, (hipStream_t)s, src, ld_src, (uint16_t*)dst, ld_dst, rows, cols);
    SPRC_CHECK_LAUNCH("sprc_transpose_f32_to16");
    return SPRC_OK;
}

extern "C" int sprc_colsum_f32(const float* x, int64_t ld, int32_t M, int32_t N, float* out, int32_t accumulate, sprc_stream s) {
    SPRC_REQUIRE(x && out && M > 0 && N > 0 && ld >= N, "sprc_colsum_f32: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)s, x, ld, M, N, out, accumulate);
    SPRC_CHECK_LAUNCH("sprc_colsum_f32");
    return SPRC_OK;
}

extern "C" int sprc_gelu_fwd(const float* x, float* y, size_t n, sprc_stream s) {
    SPRC_REQUIRE(x && y, "sprc_gelu_fwd: null pointer");
    if (n == 0) return SPRC_OK;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid1d((int64_t)n)), dim3(256), 0, (hipStream_t)s, x, y, n);
    SPRC_CHECK_LAUNCH("sprc_gelu_fwd");
    return SPRC_OK;
}

extern "C" int sprc_gelu_bwd(const float* x, const float* dy, float* dx, size_t n, sprc_stream s) {
    SPRC_REQUIRE(x && dy && dx, "sprc_gelu_bwd: null pointer");
    if (n == 0) return SPRC_OK;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid1d((int64_t)n)), dim3(256), 0, (hipStream_t)s, x, dy, dx, n);
    SPRC_CHECK_LAUNCH("sprc_gelu_bwd");
    return SPRC_OK;
}

extern "C" size_t sprc_layernorm_bwd_workspace_bytes(int32_t M, int32_t D) {
    return (size_t)((M + LNB_ROWS * LNB_RPW - 1) / (LNB_ROWS * LNB_RPW)) * 2 * D * sizeof(float);
}

extern "C" int sprc_layernorm_bwd(const float* x, int64_t ldx, const float* gamma, const float* dy, int64_t lddy, float eps, int32_t M,
                                  int32_t D, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, sprc_stream s) {
    SPRC_REQUIRE(x && gamma && dy && dgamma && dbeta && ws && M > 0, "sprc_layernorm_bwd: bad arguments");
    SPRC_REQUIRE(D > 0 && D % 4 == 0 && D <= 64 * 4 * LNB_MAXC && ldx % 4 == 0 && lddy % 4 == 0 && (!dx || lddx % 4 == 0),
                 "sprc_layernorm_bwd: D=%d unsupported (D %% 4 == 0, D <= 2048, leading dimensions %% 4 == 0)", D);
    SPRC_REQUIRE(ws_bytes >= sprc_layernorm_bwd_workspace_bytes(M, D) && ((uintptr_t)ws % 16) == 0, "sprc_layernorm_bwd: workspace too small");
    const int nblocks = (M + LNB_ROWS * LNB_RPW - 1) / (LNB_ROWS * LNB_RPW);
    LnBwdParams p{x, ldx, gamma, dy, lddy, eps, M, D, dx, lddx, (float*)ws};
    const size_t lds = (size_t)LNB_ROWS * 2 * D * sizeof(float);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblocks), dim3(64 * LNB_ROWS), lds, (hipStream_t)s, p);
    SPRC_CHECK_LAUNCH("sprc_layernorm_bwd");
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * D + 255) / 256), dim3(256), 0, (hipStream_t)s, (const float*)ws, nblocks, D, dgamma, dbeta);
    SPRC_CHECK_LAUNCH("sprc_layernorm_bwd(reduce)");
    return SPRC_OK;
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* x, const float* resid, float* y, size_t n, uint64_t seed, uint32_t site,
                                                      uint32_t thresh, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = drop_keep(seed, site, i, thresh) ? x[i] * scale : 0.f;
        if (resid != nullptr) v += resid[i];
        y[i] = v;
    }
}

extern "C" int sprc_dropout_f32(const float* x, const float* resid, float* y, size_t n, uint64_t seed, uint32_t site, float p, sprc_stream s) {
    SPRC_REQUIRE(x && y, "sprc_dropout_f32: null pointer");
    SPRC_REQUIRE(p >= 0.f && p < 1.f, "sprc_dropout_f32: p = %f outside [0, 1)", (double)p);
    if (n == 0) return SPRC_OK;
    const size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)s, x, resid, y, n, seed, site,
                       p > 0.f ? drop_thresh(p) : 0u, p > 0.f ? 1.0f / (1.0f - p) : 1.0f);
    SPRC_CHECK_LAUNCH("sprc_dropout_f32");
    return SPRC_OK;
}

extern "C" int sprc_attention_bwd(const sprc_attention_bwd_args* a, sprc_stream s) {
    SPRC_REQUIRE(a && a->q && a->k && a->v && a->dout && a->dq && a->dk && a->dv && a->scratch, "sprc_attention_bwd: null pointer");
    SPRC_REQUIRE(a->B > 0 && a->H > 0 && a->Tq > 0 && a->Tk > 0 && a->head_dim == 64, "sprc_attention_bwd: head_dim must be 64 (the Q-Former's)");
    SPRC_REQUIRE(a->Tk <= 64 * AB_MAXK, "sprc_attention_bwd: Tk=%d > %d", a->Tk, 64 * AB_MAXK);
    SPRC_REQUIRE(a->ldk % 4 == 0 && a->ldv % 4 == 0 && ((uintptr_t)a->k % 16) == 0 && ((uintptr_t)a->v % 16) == 0, "sprc_attention_bwd: k / v must be 16-byte aligned rows");
    const size_t need = (size_t)2 * a->B * a->H * a->Tq * a->Tk * sizeof(float);
    SPRC_REQUIRE(a->scratch_bytes >= need, "sprc_attention_bwd: scratch too small (%zu needed)", need);
    float* P = (float*)a->scratch;
    AttnBwdParams p{a->B, a->H, a->Tq, a->Tk, a->q, a->k, a->v, a->dout, a->ldq, a->ldk, a->ldv, a->lddo, a->key_mask, a->scale,
                    a->dq, a->dk, a->dv, a->lddq, a->lddk, a->lddv, P, P + (size_t)a->B * a->H * a->Tq * a->Tk,
                    a->drop_p > 0.f ? drop_thresh(a->drop_p) : 0u, a->drop_site, a->drop_seed, a->drop_p > 0.f ? 1.0f / (1.0f - a->drop_p) : 1.0f};
    SPRC_REQUIRE(a->drop_p >= 0.f && a->drop_p < 1.f, "sprc_attention_bwd: drop_p in [0, 1)");
    const int64_t rows = (int64_t)a->B * a->H * a->Tq, keys = (int64_t)a->B * a->H * a->Tk;
    hipLaunchKernelGGL(attn_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, p);
    SPRC_CHECK_LAUNCH("sprc_attention_bwd(rows)");
    hipLaunchKernelGGL(attn_bwd_keys_kernel, dim3((unsigned)((keys + 3) / 4)), dim3(256), 0, (hipStream_t)s, p);
    SPRC_CHECK_LAUNCH("sprc_attention_bwd(keys)");
    return SPRC_OK;
}

extern "C" int sprc_qformer_embed_rows(const sprc_qformer_embed_args* a, float* pre, sprc_stream s) {
    SPRC_REQUIRE(a && pre && (a->query_embeds || a->Lq == 0), "sprc_qformer_embed_rows: null pointer");
    SPRC_REQUIRE(a->B > 0 && a->Lq >= 0 && a->Lt >= 0 && a->Lq + a->Lt > 0 && a->hidden % 4 == 0, "sprc_qformer_embed_rows: bad shape");
    SPRC_REQUIRE(a->Lt == 0 || (a->input_ids && a->word_emb && a->pos_emb), "sprc_qformer_embed_rows: text tables missing");
    hipLaunchKernelGGL(embed_rows_kernel, dim3(grid1d((int64_t)a->B * (a->Lq + a->Lt) * (a->hidden / 4))), dim3(256), 0, (hipStream_t)s, *a, pre);
    SPRC_CHECK_LAUNCH("sprc_qformer_embed_rows");
    return SPRC_OK;
}

extern "C" int sprc_qformer_embed_bwd(const sprc_qformer_embed_args* a, const float* dpre, float* dquery, int64_t dq_bstride, float* dword,
                                      float* dpos, sprc_stream s) {
    SPRC_REQUIRE(a && dpre, "sprc_qformer_embed_bwd: null pointer");
    SPRC_REQUIRE(a->Lt == 0 || (a->input_ids && dword && dpos), "sprc_qformer_embed_bwd: text gradients need input_ids, dword, dpos");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid1d((int64_t)a->B * (a->Lq + a->Lt) * a->hidden)), dim3(256), 0, (hipStream_t)s, *a, dpre, dquery,
                       dq_bstride, dword, dpos);
    SPRC_CHECK_LAUNCH("sprc_qformer_embed_bwd");
    return SPRC_OK;
}

extern "C" int sprc_sim_max_bwd(const float* fusion, const float* feats, const float* dsim, int32_t B, int32_t N, int32_t J, int32_t E,
                                float* dfusion, float* dfeats, int32_t* jstar, sprc_stream s) {
    SPRC_REQUIRE(fusion && feats && dsim && dfusion && dfeats && jstar && B > 0 && N > 0 && J > 0 && E > 0, "sprc_sim_max_bwd: bad arguments");
    hipLaunchKernelGGL(sim_argmax_kernel, dim3((unsigned)(((int64_t)B * N + 3) / 4)), dim3(256), 0, (hipStream_t)s, fusion, feats, B, N, J, E, jstar);
    SPRC_CHECK_LAUNCH("sprc_sim_max_bwd(argmax)");
    hipLaunchKernelGGL(sim_bwd_fusion_kernel, dim3(B), dim3(256), 0, (hipStream_t)s, feats, dsim, (const int*)jstar, B, N, J, E, dfusion);
    SPRC_CHECK_LAUNCH("sprc_sim_max_bwd(fusion)");
    hipLaunchKernelGGL(sim_bwd_feats_kernel, dim3(N * J), dim3(256), 0, (hipStream_t)s, fusion, dsim, (const int*)jstar, B, N, J, E, dfeats);
    SPRC_CHECK_LAUNCH("sprc_sim_max_bwd(feats)");
    return SPRC_OK;
}

extern "C" int sprc_contrastive_ce_bwd(const float* sim, int64_t ld, int32_t B, float temp, float grad, float* dsim, float* dtemp, sprc_stream s) {
    SPRC_REQUIRE(sim && dsim && B > 0 && temp > 0.f && ld >= B, "sprc_contrastive_ce_bwd: bad arguments");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, sim, ld, B, temp, grad, dsim, dtemp);
    SPRC_CHECK_LAUNCH("sprc_contrastive_ce_bwd");
    return SPRC_OK;
}

extern "C" int sprc_l2norm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, float* dx, int64_t lddx, int32_t M, int32_t D, sprc_stream s) {
    SPRC_REQUIRE(x && dy && dx && M > 0 && D > 0, "sprc_l2norm_bwd: bad arguments");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)s, x, ldx, dy, lddy, dx, lddx, M, D);
    SPRC_CHECK_LAUNCH("sprc_l2norm_bwd");
    return SPRC_OK;
}

extern "C" int sprc_align_mse_bwd(const float* h, int64_t sample_stride, int32_t Lq, int32_t D, const float* prompt, int32_t B, float grad,
                                  float* dh, int64_t d_stride, sprc_stream s) {
    SPRC_REQUIRE(h && prompt && dh && B > 0 && Lq > 0 && D > 0, "sprc_align_mse_bwd: bad arguments");
    hipLaunchKernelGGL(align_mse_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)s, h, sample_stride, Lq, D, prompt, B, grad, dh, d_stride);
    SPRC_CHECK_LAUNCH("sprc_align_mse_bwd");
    return SPRC_OK;
}
