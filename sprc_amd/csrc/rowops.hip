// rowops.hip -- HBM-bound row kernels: LayerNorm, L2-normalise, embeddings, patch extraction, casts.
//
// One 64-lane wave owns one row; the row lives in registers as up to MAXC float4 chunks per lane
// (D <= 2048, D % 4 == 0; the path uses 1408 / 1024 / 768 / 256), loads/stores are 16 B per lane and
// coalesced, statistics are fp32 two-pass (mean, then centred variance) reduced with wave shuffles.
#include "common.hpp"

namespace sprc {

// Cache policy of the row kernels (bit mask, -DSPRC_LN_NT=0: no hints).  2: the fp32 row is LOADED non-temporally -- the residual stream is read once
// here and next by a GEMM epilogue a whole product later (86.76 -> 86.54 ms per bench step on top of the GEMM's non-temporal stores); 1: non-temporal
// stores of the 16-bit copy -- the next GEMM's A operand -- cost 0.15 ms and stay off (profiles/r06_nt_ab.txt).
#ifndef SPRC_LN_NT
#define SPRC_LN_NT 2
#endif
constexpr int MAXC = 8;          // float4 chunks per lane -> D <= 64*4*8 = 2048
constexpr int ROWS_PER_BLOCK = 4;

struct RowRegs {
    float4 v[MAXC];
};

__device__ __forceinline__ void load_row(RowRegs& r, const float* x, int nch, int lane) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
#if SPRC_LN_NT & 2
        if (i < nch) { const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + i); r.v[c] = make_float4(t[0], t[1], t[2], t[3]); }
        else r.v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#else
        r.v[c] = (i < nch) ? reinterpret_cast<const float4*>(x)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    }
}

__device__ __forceinline__ void layernorm_regs(RowRegs& r, int nch, int D, int lane, const float* gamma,
                                               const float* beta, float eps) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) s += (r.v[c].x + r.v[c].y) + (r.v[c].z + r.v[c].w);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (lane + c * 64 < nch) {
            const float a = r.v[c].x - mean, b = r.v[c].y - mean, cc = r.v[c].z - mean, d = r.v[c].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nch) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[i];
            const float4 b = reinterpret_cast<const float4*>(beta)[i];
            r.v[c].x = (r.v[c].x - mean) * rstd * g.x + b.x;
            r.v[c].y = (r.v[c].y - mean) * rstd * g.y + b.y;
            r.v[c].z = (r.v[c].z - mean) * rstd * g.z + b.z;
            r.v[c].w = (r.v[c].w - mean) * rstd * g.w + b.w;
        }
    }
}

// KIND: the operand copy's type -- 0 fp32, 1 bf16, 2 fp16, 3 split row [hi fp16 | lo e4m3 | hi e4m3] (SPRC_F32 / _BF16 / _F16 / _F16X3)
template <int KIND>
__device__ __forceinline__ void store_row(const RowRegs& r, float* y32, void* y16, int nch, int lane) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nch) {
            if (y32) reinterpret_cast<float4*>(y32)[i] = r.v[c];
            if (y16) {
                if constexpr (KIND == 3) {               // split row of logical width D = 4 nch: [D fp16 | D e4m3 | D e4m3]
                    store_split4(reinterpret_cast<char*>(y16), 4 * nch, 4 * i, r.v[c].x, r.v[c].y, r.v[c].z, r.v[c].w);
                } else if constexpr (KIND != 0) {
                    uint2 pk;
                    pk.x = pack16x2<KIND == 2>(r.v[c].x, r.v[c].y);
                    pk.y = pack16x2<KIND == 2>(r.v[c].z, r.v[c].w);
#if SPRC_LN_NT & 1
                    __builtin_nontemporal_store(u32x2{pk.x, pk.y}, reinterpret_cast<u32x2*>(y16) + i);
#else
                    reinterpret_cast<uint2*>(y16)[i] = pk;
#endif
                } else {
                    reinterpret_cast<float4*>(y16)[i] = r.v[c];
                }
            }
        }
    }
}

struct LnParams {
    int M, D;
    const float* x; int64_t ldx; sprc_rowmap xmap;
    const float* gamma; const float* beta; float eps;
    float* y32; int64_t ld32; sprc_rowmap ymap;
    void* y16; int64_t ld16;
    const _Float16* add; int64_t ld_add;      // optional fp16 branch output added to x before the statistics
    float* sum32; int64_t ld_sum;             // optional: x + add (the residual-stream update), may alias x
};

// r += add row (fp16, 8 B per lane and chunk)
__device__ __forceinline__ void add_row_f16(RowRegs& r, const _Float16* a, int nch, int lane) {
    typedef __attribute__((ext_vector_type(4))) _Float16 half4;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nch) {
            const half4 h = reinterpret_cast<const half4*>(a)[i];
            r.v[c].x += (float)h[0]; r.v[c].y += (float)h[1]; r.v[c].z += (float)h[2]; r.v[c].w += (float)h[3];
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void layernorm_kernel(LnParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int nch = p.D >> 2;
    RowRegs r;
    const int64_t xr = map_row(p.xmap, row);
    load_row(r, p.x + xr * p.ldx, nch, lane);
    if (p.add != nullptr) {
        add_row_f16(r, p.add + xr * p.ld_add, nch, lane);
        if (p.sum32 != nullptr) store_row<0>(r, p.sum32 + xr * p.ld_sum, nullptr, nch, lane);
    }
    layernorm_regs(r, nch, p.D, lane, p.gamma, p.beta, p.eps);
    const int64_t yr = map_row(p.ymap, row);
    store_row<KIND>(r, p.y32 ? p.y32 + yr * p.ld32 : nullptr,
                    p.y16 ? (char*)p.y16 + yr * p.ld16 * (KIND ? 2 : 4) : nullptr, nch, lane);
}

// LayerNorm whose operand copy is e4m3fn: y8 = sat(LN(x) * q_scale)  (q_scale = 1 / the consumer GEMM's a_scale)

__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void layernorm_fp8_kernel(LnParams p, float q_scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int nch = p.D >> 2;
    RowRegs r;
    const int64_t xr = map_row(p.xmap, row);
    load_row(r, p.x + xr * p.ldx, nch, lane);
    layernorm_regs(r, nch, p.D, lane, p.gamma, p.beta, p.eps);
    const int64_t yr = map_row(p.ymap, row);
    if (p.y32) store_row<0>(r, p.y32 + yr * p.ld32, nullptr, nch, lane);
    uint32_t* y8 = reinterpret_cast<uint32_t*>((char*)p.y16 + yr * p.ld16);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nch) y8[i] = pack_fp8x4(r.v[c].x * q_scale, r.v[c].y * q_scale, r.v[c].z * q_scale, r.v[c].w * q_scale);
    }
}

// amax[0] = max(amax[0], max |x|) over a bf16 tensor (calibration of the fp8 activation scales; values are >= 0, so the
// integer ordering of the fp32 bit patterns is the numeric one and atomicMax on the bits is exact)
template <bool F16>
__global__ __launch_bounds__(256) void absmax_bf16_kernel(const uint16_t* __restrict__ x, size_t n, float* amax) {
    float m = 0.f;
    if constexpr (F16) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
            m = fmaxf(m, fabsf((float)__builtin_bit_cast(_Float16, x[i])));
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0 && m == m) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));
        return;
    }
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (size_t)gridDim.x * 256 * 8) {
        if (i + 8 <= n) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m = fmaxf(m, fabsf(__uint_as_float(v[e] << 16)));
                m = fmaxf(m, fabsf(__uint_as_float(v[e] & 0xffff0000u)));
            }
        } else {
            for (size_t j = i; j < n; ++j) m = fmaxf(m, fabsf(bf16_bits_to_f32(x[j])));
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m == m) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));
}

template <int KIND>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void qformer_embed_kernel(sprc_qformer_embed_args p) {
    const int lane = threadIdx.x & 63;
    const int S = p.Lq + p.Lt;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= p.B * S) return;
    const int b = row / S, t = row % S;
    const int nch = p.hidden >> 2;
    RowRegs r;
    if (p.no_img) {
        // Qformer.py:88-104 (no_img): rows = [text[0] ; the Lq query rows ; text[1:]], and EVERY row gets its position t
        if (t >= 1 && t <= p.Lq) {
            load_row(r, p.query_embeds + (int64_t)b * p.q_bstride + (int64_t)(t - 1) * p.hidden, nch, lane);
        } else {
            int64_t id = p.input_ids[(int64_t)b * p.Lt + (t == 0 ? 0 : t - p.Lq)];
            id = id < 0 ? 0 : (id >= p.vocab ? p.vocab - 1 : id);
            load_row(r, p.word_emb + id * p.hidden, nch, lane);
        }
        RowRegs pe;
        load_row(pe, p.pos_emb + (int64_t)t * p.hidden, nch, lane);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            r.v[c].x += pe.v[c].x; r.v[c].y += pe.v[c].y; r.v[c].z += pe.v[c].z; r.v[c].w += pe.v[c].w;
        }
    } else if (t < p.Lq) {
        load_row(r, p.query_embeds + (int64_t)b * p.q_bstride + (int64_t)t * p.hidden, nch, lane);
    } else {
        const int pos = t - p.Lq;
        int64_t id = p.input_ids[(int64_t)b * p.Lt + pos];
        id = id < 0 ? 0 : (id >= p.vocab ? p.vocab - 1 : id);          // memory safety; the host validates CPU inputs
        RowRegs pe;
        load_row(r, p.word_emb + id * p.hidden, nch, lane);
        load_row(pe, p.pos_emb + (int64_t)pos * p.hidden, nch, lane);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            r.v[c].x += pe.v[c].x; r.v[c].y += pe.v[c].y; r.v[c].z += pe.v[c].z; r.v[c].w += pe.v[c].w;
        }
    }
    layernorm_regs(r, nch, p.hidden, lane, p.gamma, p.beta, p.eps);
    store_row<KIND>(r, p.y32 ? p.y32 + (int64_t)row * p.hidden : nullptr,
                    p.y16 ? (char*)p.y16 + (int64_t)row * p.hidden * (KIND == 3 ? 4 : KIND ? 2 : 4) : nullptr, nch, lane);
}

template <int KIND>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void l2norm_kernel(const float* x, int64_t ldx, float* y32, void* y16,
                                                                     int64_t ldy, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = D >> 2;
    RowRegs r;
    load_row(r, x + (int64_t)row * ldx, nch, lane);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) q += (r.v[c].x * r.v[c].x + r.v[c].y * r.v[c].y) + (r.v[c].z * r.v[c].z + r.v[c].w * r.v[c].w);
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(q)), 1e-12f);       // F.normalize: x / max(||x||, eps)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { r.v[c].x *= inv; r.v[c].y *= inv; r.v[c].z *= inv; r.v[c].w *= inv; }
    store_row<KIND>(r, y32 ? y32 + (int64_t)row * ldy : nullptr,
                    y16 ? (char*)y16 + (int64_t)row * ldy * (KIND ? 2 : 4) : nullptr, nch, lane);
}

template <bool F16>
__global__ void cast_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        uint2 pk;
        pk.x = pack16x2<F16>(v.x, v.y);
        pk.y = pack16x2<F16>(v.z, v.w);
        reinterpret_cast<uint2*>(dst)[i] = pk;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = (uint16_t)(pack16x2<F16>(src[i], 0.f) & 0xffffu);
}

// rows[(b*G*G + py*G + px), k] = image[b, c, py*P + i, px*P + j],  k = c*P*P + i*P + j  (zero for k >= 3*P*P)
template <int KIND>
__global__ void im2row_kernel(const float* __restrict__ img, void* __restrict__ rows, int B, int S, int P, int kpad) {
    const int G = S / P, PP = P * P, kreal = 3 * PP;
    const int64_t total = (int64_t)B * G * G * kpad;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int k = (int)(e % kpad);
        const int64_t row = e / kpad;
        float v = 0.f;
        if (k < kreal) {
            const int c = k / PP, ij = k % PP, i = ij / P, j = ij % P;
            const int px = (int)(row % G), py = (int)((row / G) % G);
            const int64_t b = row / (G * G);
            v = img[((b * 3 + c) * S + (py * P + i)) * (int64_t)S + (px * P + j)];
        }
        if constexpr (KIND == 3) {              // SPRC_F16X3: split row of logical width kpad (4 kpad bytes)
            store_split1(reinterpret_cast<char*>(rows) + row * 4 * (int64_t)kpad, kpad, k, v);
        } else if constexpr (KIND != 0) reinterpret_cast<uint16_t*>(rows)[e] = (uint16_t)(pack16x2<KIND == 2>(v, 0.f) & 0xffffu);
        else reinterpret_cast<float*>(rows)[e] = v;
    }
}

__global__ void vit_assemble_kernel(const float* __restrict__ patch_out, const float* __restrict__ cls,
                                    const float* __restrict__ pos, float* __restrict__ x, int B, int T, int W4) {
    const int64_t total = (int64_t)B * T * W4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int c = (int)(e % W4);
        const int t = (int)((e / W4) % T);
        const int64_t b = e / ((int64_t)W4 * T);
        const float4 pe = reinterpret_cast<const float4*>(pos)[(int64_t)t * W4 + c];
        const float4 v = (t == 0) ? reinterpret_cast<const float4*>(cls)[c]
                                  : reinterpret_cast<const float4*>(patch_out)[(b * (T - 1) + (t - 1)) * W4 + c];
        reinterpret_cast<float4*>(x)[e] = make_float4(v.x + pe.x, v.y + pe.y, v.z + pe.z, v.w + pe.w);
    }
}

__global__ void qformer_mask_kernel(const int64_t* __restrict__ mask, float* __restrict__ out, int B, int Lq, int Lt) {
    const int S = Lq + Lt;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * S) return;
    const int b = i / S, j = i % S;
    out[i] = (j < Lq) ? 0.f : (1.0f - (float)mask[(int64_t)b * Lt + (j - Lq)]) * -10000.0f;
}

static int grid_for(int64_t n, int block, int cap = 256 * 8) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace sprc

using namespace sprc;

extern "C" int sprc_cast_f32_to_16(const float* src, void* dst, size_t n, int32_t dtype, sprc_stream s) {
    SPRC_REQUIRE(src && dst, "sprc_cast_f32_to_16: null pointer");
    SPRC_REQUIRE(is16(dtype), "sprc_cast_f32_to_16: dtype %d is not a 16-bit type", dtype);
    if (n == 0) return SPRC_OK;
    SPRC_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "sprc_cast_f32_to_16: misaligned");
    const dim3 grid(grid_for((int64_t)(n / 4 + 1), 256)), block(256);
    if (dtype == SPRC_F16) hipLaunchKernelGGL(cast_kernel<true>, grid, block, 0, (hipStream_t)s, src, reinterpret_cast<uint16_t*>(dst), n);
    else hipLaunchKernelGGL(cast_kernel<false>, grid, block, 0, (hipStream_t)s, src, reinterpret_cast<uint16_t*>(dst), n);
    SPRC_CHECK_LAUNCH("sprc_cast_f32_to_16");
    return SPRC_OK;
}

// fp32 [rows, cols] -> split rows [rows, 4 cols bytes] = [hi fp16 | lo e4m3 | hi e4m3]
__global__ void cast_x3_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int64_t rows, int cols4) {
    const int64_t total = rows * cols4, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t row = e / cols4;
        const int c = (int)(e - row * cols4);
        const float4 v = reinterpret_cast<const float4*>(src)[e];
        store_split4(reinterpret_cast<char*>(dst) + row * 16 * cols4, 4 * cols4, 4 * c, v.x, v.y, v.z, v.w);
    }
}

extern "C" int sprc_cast_f32_to_x3(const float* src, void* dst, int64_t rows, int32_t cols, sprc_stream s) {
    SPRC_REQUIRE(src && dst && rows >= 0 && cols > 0 && cols % 4 == 0, "sprc_cast_f32_to_x3: bad arguments (cols %% 4 == 0)");
    if (rows == 0) return SPRC_OK;
    SPRC_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "sprc_cast_f32_to_x3: misaligned");
    hipLaunchKernelGGL(cast_x3_kernel, dim3(grid_for(rows * (cols / 4), 256, 256 * 16)), dim3(256), 0, (hipStream_t)s, src,
                       reinterpret_cast<_Float16*>(dst), rows, cols / 4);
    SPRC_CHECK_LAUNCH("sprc_cast_f32_to_x3");
    return SPRC_OK;
}

extern "C" int sprc_cast_f32_to_bf16(const float* src, uint16_t* dst, size_t n, sprc_stream s) {
    return sprc_cast_f32_to_16(src, dst, n, SPRC_BF16, s);
}

extern "C" int sprc_layernorm(const sprc_layernorm_args* a, sprc_stream s) {
    SPRC_REQUIRE(a && a->x && a->gamma && a->beta, "sprc_layernorm: null pointer");
    SPRC_REQUIRE(a->M > 0, "sprc_layernorm: M=%d", a->M);
    SPRC_REQUIRE(a->D > 0 && a->D % 4 == 0 && a->D <= 64 * 4 * MAXC, "sprc_layernorm: D=%d unsupported (D%%4==0, D<=2048)", a->D);
    SPRC_REQUIRE(a->ldx % 4 == 0 && (!a->y32 || a->ld32 % 4 == 0) && (!a->y16 || a->ld16 % 4 == 0),
                 "sprc_layernorm: leading dimensions must be multiples of 4");
    SPRC_REQUIRE(a->y32 || a->y16, "sprc_layernorm: no output");
    SPRC_REQUIRE(a->add16 == nullptr || (a->ld_add % 4 == 0 && ((uintptr_t)a->add16 % 8) == 0), "sprc_layernorm: add16 must be 8-byte aligned, ld_add % 4 == 0");
    SPRC_REQUIRE(a->sum32 == nullptr || (a->add16 != nullptr && a->ld_sum % 4 == 0), "sprc_layernorm: sum32 needs add16 and ld_sum % 4 == 0");
    LnParams p{a->M, a->D, a->x, a->ldx, a->xmap, a->gamma, a->beta, a->eps, a->y32, a->ld32, a->ymap, a->y16, a->ld16,
               reinterpret_cast<const _Float16*>(a->add16), a->ld_add, a->sum32, a->ld_sum};
    const dim3 grid((a->M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(64 * ROWS_PER_BLOCK);
    ProfScope prof(SPRC_K_ROWOPS, (hipStream_t)s, 8.0 * a->M * (double)a->D,
                   (double)a->M * a->D * (4.0 + (a->y32 ? 4.0 : 0.0) + (a->y16 ? (a->out_dtype == SPRC_F16X3 ? 4.0 : (double)dtype_size(a->out_dtype)) : 0.0) +
                                          (a->add16 ? 2.0 : 0.0) + (a->sum32 ? 4.0 : 0.0)));
    if (a->out_dtype == SPRC_FP8) {
        SPRC_REQUIRE(a->y16 != nullptr && a->y16_scale > 0.f && a->add16 == nullptr && ((uintptr_t)a->y16 % 4) == 0,
                     "sprc_layernorm(fp8): needs y16, y16_scale > 0 and no fused add");
        hipLaunchKernelGGL(layernorm_fp8_kernel, grid, block, 0, (hipStream_t)s, p, a->y16_scale);
    } else if (a->out_dtype == SPRC_BF16) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, (hipStream_t)s, p);
    else if (a->out_dtype == SPRC_F16) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, (hipStream_t)s, p);
    else if (a->out_dtype == SPRC_F16X3) {
        SPRC_REQUIRE(a->y16 == nullptr || (a->ld16 >= 2 * (int64_t)a->D && ((uintptr_t)a->y16 % 8) == 0), "sprc_layernorm(F16X3): ld16 >= 2 D (fp16 units: a split row is 4 D bytes), y16 8-byte aligned");
        hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, (hipStream_t)s, p);
    }
    else hipLaunchKernelGGL(layernorm_kernel<0>, grid, block, 0, (hipStream_t)s, p);
    SPRC_CHECK_LAUNCH("sprc_layernorm");
    return SPRC_OK;
}

// prob[p] = softmax(mean_j (W h[p,j,:] + b))[1] over the first Lq rows of every sample (blip2_qformer_cir_rerank.py:439-445);
// mean_j (W h_j + b) = W (mean_j h_j) + b.  One wave per sample.
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void itm_head_kernel(const float* __restrict__ h, int64_t sample_stride, int Lq,
                                                                       int D, const float* __restrict__ w,
                                                                       const float* __restrict__ bias, int P, float* prob) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (p >= P) return;
    const int nch = D >> 2;
    RowRegs acc, r;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc.v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < Lq; ++j) {
        load_row(r, h + (int64_t)p * sample_stride + (int64_t)j * D, nch, lane);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { acc.v[c].x += r.v[c].x; acc.v[c].y += r.v[c].y; acc.v[c].z += r.v[c].z; acc.v[c].w += r.v[c].w; }
    }
    float l[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        load_row(r, w + (int64_t)k * D, nch, lane);
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) d += (acc.v[c].x * r.v[c].x + acc.v[c].y * r.v[c].y) + (acc.v[c].z * r.v[c].z + acc.v[c].w * r.v[c].w);
        l[k] = wave_sum(d) / (float)Lq + bias[k];
    }
    if (lane == 0) {
        const float mx = fmaxf(l[0], l[1]);
        const float e0 = expf(l[0] - mx), e1 = expf(l[1] - mx);
        prob[p] = e1 / (e0 + e1);
    }
}

// ---- training-forward losses (align_prompt.py:157-193) -------------------------------------------------------------------
// loss = mean_b( logsumexp_n(sim[b,n] / temp) - sim[b,b] / temp ): F.cross_entropy(sim / temp, arange(B)).  One workgroup.
__global__ __launch_bounds__(256) void contrastive_ce_kernel(const float* __restrict__ sim, int64_t ld, int B, float inv_temp, float* loss) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int b = wave; b < B; b += 4) {
        const float* row = sim + (int64_t)b * ld;
        float mx = -INFINITY;
        for (int n = lane; n < B; n += 64) mx = fmaxf(mx, row[n] * inv_temp);
        mx = wave_max(mx);
        float se = 0.f;
        for (int n = lane; n < B; n += 64) se += expf(row[n] * inv_temp - mx);
        se = wave_sum(se);
        acc += (mx + logf(se)) - row[b] * inv_temp;
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)B;
}

// loss = mean over (b, d) of (mean_j h[b, j, d] - mean_j prompt[j, d])^2, j < Lq: F.mse_loss(fusion[:, :32].mean(1), prompt.mean(1))
__global__ __launch_bounds__(256) void align_mse_kernel(const float* __restrict__ h, int64_t sample_stride, int Lq, int D,
                                                        const float* __restrict__ prompt, int B, float* loss) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    RowRegs pm, r;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) pm.v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < Lq; ++j) {
        load_row(r, prompt + (int64_t)j * D, nch, lane);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { pm.v[c].x += r.v[c].x; pm.v[c].y += r.v[c].y; pm.v[c].z += r.v[c].z; pm.v[c].w += r.v[c].w; }
    }
    float acc = 0.f;
    const float inv = 1.0f / (float)Lq;
    for (int b = wave; b < B; b += 4) {
        RowRegs hm;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) hm.v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < Lq; ++j) {
            load_row(r, h + (int64_t)b * sample_stride + (int64_t)j * D, nch, lane);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) { hm.v[c].x += r.v[c].x; hm.v[c].y += r.v[c].y; hm.v[c].z += r.v[c].z; hm.v[c].w += r.v[c].w; }
        }
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const float dx = (hm.v[c].x - pm.v[c].x) * inv, dy = (hm.v[c].y - pm.v[c].y) * inv;
            const float dz = (hm.v[c].z - pm.v[c].z) * inv, dw = (hm.v[c].w - pm.v[c].w) * inv;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        acc += wave_sum(q);
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((part[0] + part[1]) + (part[2] + part[3])) / ((float)B * (float)D);
}

extern "C" int sprc_contrastive_ce(const float* sim, int64_t ld, int32_t B, float temp, float* loss, sprc_stream s) {
    SPRC_REQUIRE(sim && loss && B > 0 && temp > 0.f && ld >= B, "sprc_contrastive_ce: bad arguments");
    hipLaunchKernelGGL(contrastive_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, sim, ld, B, 1.0f / temp, loss);
    SPRC_CHECK_LAUNCH("sprc_contrastive_ce");
    return SPRC_OK;
}

extern "C" int sprc_align_mse(const float* h, int64_t sample_stride, int32_t Lq, int32_t D, const float* prompt, int32_t B, float* loss,
                              sprc_stream s) {
    SPRC_REQUIRE(h && prompt && loss && B > 0 && Lq > 0, "sprc_align_mse: bad arguments");
    SPRC_REQUIRE(D % 4 == 0 && D <= 64 * 4 * MAXC && sample_stride % 4 == 0, "sprc_align_mse: D=%d unsupported", D);
    hipLaunchKernelGGL(align_mse_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, h, sample_stride, Lq, D, prompt, B, loss);
    SPRC_CHECK_LAUNCH("sprc_align_mse");
    return SPRC_OK;
}

extern "C" int sprc_itm_head(const float* h, int64_t sample_stride, int32_t Lq, int32_t D, const float* w, const float* b, int32_t P,
                             float* prob, sprc_stream s) {
    SPRC_REQUIRE(h && w && b && prob && P > 0 && Lq > 0, "sprc_itm_head: bad arguments");
    SPRC_REQUIRE(D % 4 == 0 && D <= 64 * 4 * MAXC && sample_stride % 4 == 0, "sprc_itm_head: D=%d unsupported", D);
    const dim3 grid((P + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(64 * ROWS_PER_BLOCK);
    hipLaunchKernelGGL(itm_head_kernel, grid, block, 0, (hipStream_t)s, h, sample_stride, Lq, D, w, b, P, prob);
    SPRC_CHECK_LAUNCH("sprc_itm_head");
    return SPRC_OK;
}

extern "C" int sprc_qformer_embed(const sprc_qformer_embed_args* a, sprc_stream s) {
    SPRC_REQUIRE(a && a->gamma && a->beta && (a->query_embeds || a->Lq == 0), "sprc_qformer_embed: null pointer");
    SPRC_REQUIRE(a->B > 0 && a->Lq >= 0 && a->Lt >= 0 && a->Lq + a->Lt > 0 && (a->Lq > 0 || !a->no_img), "sprc_qformer_embed: bad shape");
    SPRC_REQUIRE(a->Lt == 0 || (a->input_ids && a->word_emb && a->pos_emb), "sprc_qformer_embed: text tables missing");
    SPRC_REQUIRE(!a->no_img || a->Lt > 0, "sprc_qformer_embed: no_img needs text");
    SPRC_REQUIRE(a->hidden % 4 == 0 && a->hidden <= 64 * 4 * MAXC, "sprc_qformer_embed: hidden=%d unsupported", a->hidden);
    const int rows = a->B * (a->Lq + a->Lt);
    const dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(64 * ROWS_PER_BLOCK);
    if (a->out_dtype == SPRC_BF16) hipLaunchKernelGGL(qformer_embed_kernel<1>, grid, block, 0, (hipStream_t)s, *a);
    else if (a->out_dtype == SPRC_F16) hipLaunchKernelGGL(qformer_embed_kernel<2>, grid, block, 0, (hipStream_t)s, *a);
    else if (a->out_dtype == SPRC_F16X3) hipLaunchKernelGGL(qformer_embed_kernel<3>, grid, block, 0, (hipStream_t)s, *a);
    else hipLaunchKernelGGL(qformer_embed_kernel<0>, grid, block, 0, (hipStream_t)s, *a);
    SPRC_CHECK_LAUNCH("sprc_qformer_embed");
    return SPRC_OK;
}

extern "C" int sprc_l2norm_rows(const float* x, int64_t ldx, float* y32, void* y16, int64_t ldy, int32_t M, int32_t D,
                                int32_t out_dtype, sprc_stream s) {
    SPRC_REQUIRE(x && (y32 || y16), "sprc_l2norm_rows: null pointer");
    SPRC_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXC && ldx % 4 == 0 && ldy % 4 == 0, "sprc_l2norm_rows: bad shape");
    const dim3 grid((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(64 * ROWS_PER_BLOCK);
    if (out_dtype == SPRC_BF16) hipLaunchKernelGGL(l2norm_kernel<1>, grid, block, 0, (hipStream_t)s, x, ldx, y32, y16, ldy, M, D);
    else if (out_dtype == SPRC_F16) hipLaunchKernelGGL(l2norm_kernel<2>, grid, block, 0, (hipStream_t)s, x, ldx, y32, y16, ldy, M, D);
    else hipLaunchKernelGGL(l2norm_kernel<0>, grid, block, 0, (hipStream_t)s, x, ldx, y32, y16, ldy, M, D);
    SPRC_CHECK_LAUNCH("sprc_l2norm_rows");
    return SPRC_OK;
}

extern "C" int sprc_im2row(const float* images, void* rows, int32_t B, int32_t image, int32_t patch, int32_t k_pad,
                           int32_t dtype, sprc_stream s) {
    SPRC_REQUIRE(images && rows, "sprc_im2row: null pointer");
    SPRC_REQUIRE(B > 0 && patch > 0 && image % patch == 0 && k_pad >= 3 * patch * patch, "sprc_im2row: bad shape");
    const int64_t total = (int64_t)B * (image / patch) * (image / patch) * k_pad;
    if (dtype == SPRC_BF16) hipLaunchKernelGGL(im2row_kernel<1>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)s, images, rows, B, image, patch, k_pad);
    else if (dtype == SPRC_F16) hipLaunchKernelGGL(im2row_kernel<2>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)s, images, rows, B, image, patch, k_pad);
    else if (dtype == SPRC_F16X3) hipLaunchKernelGGL(im2row_kernel<3>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)s, images, rows, B, image, patch, k_pad);
    else hipLaunchKernelGGL(im2row_kernel<0>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)s, images, rows, B, image, patch, k_pad);
    SPRC_CHECK_LAUNCH("sprc_im2row");
    return SPRC_OK;
}

extern "C" int sprc_vit_assemble(const float* patch_out, const float* cls, const float* pos, float* x, int32_t B,
                                 int32_t tokens, int32_t width, sprc_stream s) {
    SPRC_REQUIRE(patch_out && cls && pos && x, "sprc_vit_assemble: null pointer");
    SPRC_REQUIRE(B > 0 && tokens > 1 && width % 4 == 0, "sprc_vit_assemble: bad shape");
    const int64_t total = (int64_t)B * tokens * (width / 4);
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)s, patch_out, cls, pos, x, B, tokens, width / 4);
    SPRC_CHECK_LAUNCH("sprc_vit_assemble");
    return SPRC_OK;
}

extern "C" int sprc_qformer_mask(const int64_t* attention_mask, float* out, int32_t B, int32_t Lq, int32_t Lt, sprc_stream s) {
    SPRC_REQUIRE(attention_mask && out && B > 0 && Lq >= 0 && Lt > 0, "sprc_qformer_mask: bad arguments");
    const int n = B * (Lq + Lt);
    hipLaunchKernelGGL(qformer_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, attention_mask, out, B, Lq, Lt);
    SPRC_CHECK_LAUNCH("sprc_qformer_mask");
    return SPRC_OK;
}

extern "C" int sprc_absmax_16(const void* x, size_t n, int32_t dtype, float* amax, sprc_stream s) {
    SPRC_REQUIRE(x && amax && n > 0 && ((uintptr_t)x % 16) == 0 && (dtype == SPRC_BF16 || dtype == SPRC_F16), "sprc_absmax_16: bad arguments");
    const size_t blocks = (n / 8 + 255) / 256;
    const dim3 grid((unsigned)(blocks < 2048 ? (blocks ? blocks : 1) : 2048));
    if (dtype == SPRC_F16)
        hipLaunchKernelGGL(absmax_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)s, reinterpret_cast<const uint16_t*>(x), n, amax);
    else
        hipLaunchKernelGGL(absmax_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)s, reinterpret_cast<const uint16_t*>(x), n, amax);
    SPRC_CHECK_LAUNCH("sprc_absmax_16");
    return SPRC_OK;
}
extern "C" int sprc_absmax_bf16(const void* x, size_t n, float* amax, sprc_stream s) { return sprc_absmax_16(x, n, SPRC_BF16, amax, s); }
