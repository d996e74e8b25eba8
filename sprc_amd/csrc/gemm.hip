// gemm.hip -- C = epilogue(A[M,K] . W[N,K]^T) on gfx950 matrix cores.
//
// Replaces every nn.Linear / F.linear / conv-as-GEMM on the SPRC retrieval path (see sprc.h).
// Both operands are K-contiguous ("B^T" form), so A and W fragments are read the same way.
//
// One kernel template, three tile configurations (workgroup tile BM x BN, K-tile = 128 bytes of K):
//   256 x 256, 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 MFMA 32x32 accumulators   (large GEMMs)
//   256 x 128, 8 waves (4 x 2), wave tile  64 x 64 = 2 x 2                            (N = 1408-class GEMMs)
//   128 x 128, 4 waves (2 x 2), wave tile  64 x 64 = 2 x 2                            (small GEMMs, 3 WGs / CU)
//     bf16: v_mfma_f32_32x32x16_bf16, K-tile = 64 elements;  f32: v_mfma_f32_32x32x2_f32 (exact fp32), 32 elements
// Why big tiles: a CU's vector-memory path delivers ~64 B/clk; a 128x128x64 tile needs 32 KiB per 512
// MFMA-cycles = 64 B/clk at peak MFMA rate (measured: 517-635 TFLOP/s, memory-path bound), a 256x256 tile
// needs half of that.
// Data movement: K-tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction,
// no VGPR round trip), double buffered, next tile issued before the current tile's MFMAs.  The LDS image of
// a wave-instruction is lane-linear (8 rows x 8 16-B slots), so the XOR swizzle (slot ^= (row>>1)&7, which
// makes every ds_read_b128 lane group hit 16 distinct slots of the 256-B bank row) is applied to the
// per-lane SOURCE address and to the fragment reads, never to the destination (guide rule 21).
// Fragment reads are software pipelined one MFMA k-step ahead (registers double buffered).
// Grid: 1-D, XCD-aware remap (block b runs on XCD b%8 -> each XCD gets a contiguous tile range) and an
// 8-row grouped tile order so concurrently resident tiles share A/W panels in the XCD's L2.
#include <stdlib.h>

#include "common.hpp"

namespace sprc {

struct GemmParams {
    int M, N, K;
    const char* A; int64_t lda_b;       // leading dims in BYTES
    const char* W; int64_t ldw_b;
    const float* bias;
    const float* resid; int64_t ldr;
    void* C; int64_t ldc;
    int a_shift, a_stride, a_off;       // row maps (rows_per_group = 1<<shift, shift<0 -> identity)
    int c_shift, c_stride, c_off;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ int64_t map_row_s(int shift, int stride, int off, int r) {
    if (shift < 0) return r;
    return (int64_t)(r >> shift) * stride + (r & ((1 << shift) - 1)) + off;
}

// erf for the bf16 epilogue: Abramowitz-Stegun 7.1.26, |abs err| <= 1.5e-7 (far below bf16 resolution)
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float y = fmaf(1.061405429f, t, -1.453152027f);
    y = fmaf(y, t, 1.421413741f);
    y = fmaf(y, t, -0.284496736f);
    y = fmaf(y, t, 0.254829592f);
    y = 1.0f - y * t * __expf(-ax * ax);
    return copysignf(y, x);
}

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// KT_BYTES: bytes of K per row per K-tile (128 or 64).  STAGES: 1 (single buffer, 2 barriers per tile),
// 2 (double buffer, __syncthreads) or >= 3 (ring: loads STAGES-1 tiles ahead, counted vmcnt + raw s_barrier so
// the prefetched tiles stay in flight across the barrier -- guide T3/T4).
template <typename T, typename OutT, int ACT, bool MAX32, int WM, int WN, int TM, int TN, int KT_BYTES, int STAGES>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 64 * WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int STAGE_BYTES = (BM + BN) * KT_BYTES;
    constexpr int SPR = KT_BYTES / 16, RPB = 256 / KT_BYTES;   // 16-B slots per row; rows per 256-B LDS bank row
    constexpr int KSTEPS = KT_BYTES / 32;                      // MFMA k-steps per K-tile
    constexpr int LA = BM * SPR / NT, LB = BN * SPR / NT;      // 16-B chunks per thread per K-tile
    static_assert(BM * SPR % NT == 0 && BN * SPR % NT == 0, "tile/threads mismatch");
    typedef typename Frag<T>::type frag_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int r32 = lane & 31, half = lane >> 5;

    // ---- tile id: XCD-contiguous remap (bijective for any grid), then grouped (8 m-tiles) order ----
    const int nwg = p.tiles_m * p.tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    constexpr int GROUP_M = 8;
    const int in_group = GROUP_M * p.tiles_n;
    const int group_id = pid / in_group;
    const int first_m = group_id * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int pid_m = first_m + (pid % in_group) % gsz;
    const int pid_n = (pid % in_group) / gsz;
    const int m0 = pid_m * BM, n0 = pid_n * BN;

    // ---- direct-to-LDS staging: lane fills physical slot (lane&7) of row (chunk>>3) with logical slot^f(row) ----
    const char* a_src[LA];
    const char* w_src[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int c = i * NT + tid, row = c / SPR, slot = (c % SPR) ^ ((row / RPB) % SPR);
        const int am = min(m0 + row, p.M - 1);
        a_src[i] = p.A + map_row_s(p.a_shift, p.a_stride, p.a_off, am) * p.lda_b + slot * 16;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int c = i * NT + tid, row = c / SPR, slot = (c % SPR) ^ ((row / RPB) % SPR);
        const int wn = min(n0 + row, p.N - 1);
        w_src[i] = p.W + (int64_t)wn * p.ldw_b + slot * 16;
    }
    auto stage = [&](int buf, int64_t ko) {
        char* dst = smem + buf * STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < LA; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + ko), (lptr_t)(dst + i * NT * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < LB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + ko), (lptr_t)(dst + BM * KT_BYTES + i * NT * 16), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int sw = (r32 / RPB) % SPR;
    const int a_off = (wr * TM * 32 + r32) * KT_BYTES, b_off = BM * KT_BYTES + (wc * TN * 32 + r32) * KT_BYTES;
    auto compute = [&](const char* st) {
        frag_t fa[2][TM], fb[2][TN];
        auto load = [&](int buf, int kk) {
            const int slot = ((kk * 2 + half) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[buf][i] = *reinterpret_cast<const frag_t*>(st + a_off + i * 32 * KT_BYTES + slot);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[buf][i] = *reinterpret_cast<const frag_t*>(st + b_off + i * 32 * KT_BYTES + slot);
        };
        load(0, 0);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            if (kk + 1 < KSTEPS) load((kk + 1) & 1, kk + 1);
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][ni], fa[kk & 1][mi], acc[mi][ni], 0, 0, 0);
            } else {
                // each lane holds 4 consecutive k of its half; MFMA step t pairs k = {8kk+t, 8kk+4+t}
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[kk & 1][ni][t], fa[kk & 1][mi][t], acc[mi][ni], 0, 0, 0);
            }
        }
    };

    const int nt = (int)(((int64_t)p.K * sizeof(T)) / KT_BYTES);
    if constexpr (STAGES == 2) {             // one barrier per K-tile; next tile in flight during the MFMAs
        stage(0, 0);
        for (int t = 0; t < nt; ++t) {
            __syncthreads();                 // tile t landed (vmcnt(0) + barrier); buffer (t+1)&1 is free again
            if (t + 1 < nt) stage((t + 1) & 1, (int64_t)(t + 1) * KT_BYTES);
            compute(smem + (t & 1) * STAGE_BYTES);
        }
    } else if constexpr (STAGES == 1) {      // one buffer: co-resident workgroups hide each other's load phase
        for (int t = 0; t < nt; ++t) {
            stage(0, (int64_t)t * KT_BYTES);
            __syncthreads();
            compute(smem);
            __syncthreads();
        }
    } else {                                 // ring of STAGES buffers, STAGES-1 tiles in flight
        constexpr int LT = LA + LB;          // VMEM ops per thread per K-tile
        static_assert(STAGES <= 4 && LT * (STAGES - 2) < 64, "vmcnt range");
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nt) stage(s, (int64_t)s * KT_BYTES);
        int buf = 0, nbuf = STAGES - 1;
        for (int t = 0; t < nt; ++t) {
            const int ahead = min(STAGES - 2, nt - 1 - t);       // tiles allowed to stay in flight behind tile t
            if (ahead >= 2) wait_vmcnt<2 * LT>();
            else if (ahead == 1) wait_vmcnt<LT>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();    // tile t landed for every wave; everyone finished reading buffer nbuf
            asm volatile("" ::: "memory");
            if (t + STAGES - 1 < nt) stage(nbuf, (int64_t)(t + STAGES - 1) * KT_BYTES);
            compute(smem + buf * STAGE_BYTES);
            buf = (buf + 1 == STAGES) ? 0 : buf + 1;
            nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
        }
    }

    // ---- epilogue.  The MFMAs compute the TRANSPOSED tile (W fragment as the A operand), so the 32x32 D layout
    // puts the C row on the lane and 4 consecutive C columns in consecutive registers:
    //   m = rbase + (lane&31),   n = cbase + 8*(r>>2) + 4*(lane>>5) + (r&3)
    // -> bias / residual / output move as 16-B (fp32) or 8-B (bf16) vectors, one row pointer per lane.
    const bool vec_ok = (p.N % 4 == 0) && (p.ldc % 4 == 0) && (p.resid == nullptr || p.ldr % 4 == 0);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = m0 + (wr * TM + mi) * 32 + r32;
        if constexpr (MAX32) {
            // rows m = query vectors, columns n = gallery tokens: max over the 32 columns of one image
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int cbase = n0 + (wc * TN + ni) * 32;
                float v = acc[mi][ni][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[mi][ni][r]);
                v = fmaxf(v, __shfl_xor(v, 32, 64));
                if (half == 0 && row < p.M && cbase < p.N)
                    reinterpret_cast<float*>(p.C)[(int64_t)row * p.ldc + (cbase >> 5)] = v;
            }
        } else {
            const bool row_ok = row < p.M;
            const int64_t prow = map_row_s(p.c_shift, p.c_stride, p.c_off, row_ok ? row : 0);
            const float* rrow = p.resid ? p.resid + prow * p.ldr : nullptr;
            OutT* crow = reinterpret_cast<OutT*>(p.C) + prow * p.ldc;
            if (vec_ok) {
                f32x4 rv[TN][4];
                if (rrow != nullptr) {                // gather the residual first (C may alias it: in-place stream)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = n0 + (wc * TN + ni) * 32 + 8 * g + 4 * half;
                            rv[ni][g] = (row_ok && col < p.N) ? *reinterpret_cast<const f32x4*>(rrow + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                }
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = n0 + (wc * TN + ni) * 32 + 8 * g + 4 * half;
                        if (!(row_ok && col < p.N)) continue;
                        f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                        if (p.bias != nullptr) v += *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (ACT == SPRC_ACT_GELU) {
                                if constexpr (sizeof(T) == 2) v[e] = 0.5f * v[e] * (1.0f + erf_as(v[e] * 0.70710678118654752440f));
                                else v[e] = gelu_erf(v[e]);
                            }
                            if constexpr (ACT == SPRC_ACT_QUICKGELU) v[e] = quick_gelu(v[e]);
                        }
                        if (rrow != nullptr) v += rv[ni][g];
                        if constexpr (sizeof(OutT) == 2) {
                            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                            const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                            *reinterpret_cast<bf16x4*>(crow + col) = o;
                        } else {
                            *reinterpret_cast<f32x4*>(crow + col) = v;
                        }
                    }
            } else {                                  // unaligned / ragged N: scalar path
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = n0 + (wc * TN + ni) * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                        if (!(row_ok && col < p.N)) continue;
                        float v = acc[mi][ni][r] + (p.bias ? p.bias[col] : 0.f);
                        if constexpr (ACT == SPRC_ACT_GELU) {
                            if constexpr (sizeof(T) == 2) v = 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f));
                            else v = gelu_erf(v);
                        }
                        if constexpr (ACT == SPRC_ACT_QUICKGELU) v = quick_gelu(v);
                        if (rrow != nullptr) v += rrow[col];
                        if constexpr (sizeof(OutT) == 2) crow[col] = (__bf16)v;
                        else crow[col] = v;
                    }
            }
        }
    }
}

static int ilog2_exact(int v) {
    if (v <= 0) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return ((1 << s) == v) ? s : -2;
}

// SPRC_GEMM_TILE: 0 = automatic (default), 1 = 128x128 single buffer, 2 = 128x128 double buffer,
//                 3 = 256x128 double buffer, 4 = 256x256 double buffer, 5 = 256x256 K64-byte tiles 4-stage ring,
//                 6 = 256x128 3-stage ring, 7 = 128x128 4-stage ring
static int tile_override() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SPRC_GEMM_TILE");
        v = e ? atoi(e) : 0;
    }
    return v;
}

template <typename T, typename OutT, int ACT, bool MAX32, int WM, int WN, int TM, int TN, int KT_BYTES, int STAGES>
static int launch_cfg(GemmParams p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, LDS = STAGES * (BM + BN) * KT_BYTES;
    if (((int64_t)p.K * sizeof(T)) % KT_BYTES != 0) {
        set_error("sprc_gemm: K=%d is not a multiple of the K-tile", p.K);
        return SPRC_EINVAL;
    }
    auto kern = gemm_kernel<T, OutT, ACT, MAX32, WM, WN, TM, TN, KT_BYTES, STAGES>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(64 * WM * WN), LDS, st, p);
    SPRC_CHECK_LAUNCH("sprc_gemm");
    return SPRC_OK;
}

template <typename T, typename OutT, int ACT, bool MAX32>
static int launch(const GemmParams& p, hipStream_t st) {
    int cfg = tile_override();
    if (cfg == 0) {
        // measured on MI355X (tools/gemm_bench.py): the 256x256 tile wins once its grid spans >= 4 rounds of the 256
        // CUs (ViT qkv / fc1, Q-Former K|V); below that (N = 1408-class and the small Q-Former GEMMs) two co-resident
        // 128x128 workgroups per CU hide each other's prologue/epilogue better.
        const int64_t t256 = (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256);
        cfg = t256 >= 1024 ? 4 : 2;
    }
    switch (cfg) {
        case 1: return launch_cfg<T, OutT, ACT, MAX32, 2, 2, 2, 2, 128, 1>(p, st);
        case 2: return launch_cfg<T, OutT, ACT, MAX32, 2, 2, 2, 2, 128, 2>(p, st);
        case 3: return launch_cfg<T, OutT, ACT, MAX32, 4, 2, 2, 2, 128, 2>(p, st);
        case 5: return launch_cfg<T, OutT, ACT, MAX32, 2, 4, 4, 2, 64, 4>(p, st);
        case 6: return launch_cfg<T, OutT, ACT, MAX32, 4, 2, 2, 2, 128, 3>(p, st);
        case 7: return launch_cfg<T, OutT, ACT, MAX32, 2, 2, 2, 2, 128, 4>(p, st);
        default: return launch_cfg<T, OutT, ACT, MAX32, 2, 4, 4, 2, 128, 2>(p, st);
    }
}

template <typename T>
static int dispatch(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st) {
    if (a->max32) return launch<T, float, SPRC_ACT_NONE, true>(p, st);
    const bool o16 = a->out_dtype == SPRC_BF16;
    switch (a->act) {
        case SPRC_ACT_NONE:
            return o16 ? launch<T, bf16_t, SPRC_ACT_NONE, false>(p, st) : launch<T, float, SPRC_ACT_NONE, false>(p, st);
        case SPRC_ACT_GELU:
            return o16 ? launch<T, bf16_t, SPRC_ACT_GELU, false>(p, st) : launch<T, float, SPRC_ACT_GELU, false>(p, st);
        case SPRC_ACT_QUICKGELU:
            return o16 ? launch<T, bf16_t, SPRC_ACT_QUICKGELU, false>(p, st)
                       : launch<T, float, SPRC_ACT_QUICKGELU, false>(p, st);
    }
    set_error("sprc_gemm: unknown activation %d", a->act);
    return SPRC_EINVAL;
}

}  // namespace sprc

extern "C" int sprc_gemm(const sprc_gemm_args* a, sprc_stream s) {
    using namespace sprc;
    SPRC_REQUIRE(a != nullptr, "sprc_gemm: null args");
    SPRC_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "sprc_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    SPRC_REQUIRE(a->dtype == SPRC_BF16 || a->dtype == SPRC_F32, "sprc_gemm: bad dtype %d", a->dtype);
    SPRC_REQUIRE(a->out_dtype == SPRC_BF16 || a->out_dtype == SPRC_F32, "sprc_gemm: bad out_dtype %d", a->out_dtype);
    const int es = (int)dtype_size(a->dtype);
    SPRC_REQUIRE(((int64_t)a->K * es) % 128 == 0, "sprc_gemm: K=%d must be a multiple of %d", a->K, 128 / es);
    SPRC_REQUIRE((a->lda * es) % 16 == 0 && (a->ldw * es) % 16 == 0, "sprc_gemm: lda/ldw must be 16-byte multiples");
    SPRC_REQUIRE(a->lda >= a->K && a->ldw >= a->K, "sprc_gemm: leading dimension < K");
    SPRC_REQUIRE(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0, "sprc_gemm: A/W must be 16-byte aligned");
    SPRC_REQUIRE(a->A && a->W && a->C, "sprc_gemm: null operand");
    if (a->max32) {
        SPRC_REQUIRE(a->N % 32 == 0, "sprc_gemm(max32): N=%d must be a multiple of 32", a->N);
        SPRC_REQUIRE(a->amap.rows_per_group == 0 && a->cmap.rows_per_group == 0, "sprc_gemm(max32): no row maps");
    }
    GemmParams p;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = (const char*)a->A; p.lda_b = a->lda * es;
    p.W = (const char*)a->W; p.ldw_b = a->ldw * es;
    p.bias = a->bias; p.resid = a->resid; p.ldr = a->ldr;
    p.C = a->C; p.ldc = a->ldc;
    p.a_shift = ilog2_exact(a->amap.rows_per_group); p.a_stride = a->amap.group_stride; p.a_off = a->amap.group_offset;
    p.c_shift = ilog2_exact(a->cmap.rows_per_group); p.c_stride = a->cmap.group_stride; p.c_off = a->cmap.group_offset;
    if (p.a_shift == -2 || p.c_shift == -2) {
        set_error("sprc_gemm: rows_per_group must be a power of two");
        return SPRC_EUNSUPPORTED;
    }
    p.tiles_m = p.tiles_n = 0;
    hipStream_t st = (hipStream_t)s;
    const double osz = a->max32 ? 4.0 / 32.0 : (double)dtype_size(a->out_dtype);
    ProfScope prof(a->dtype == SPRC_BF16 ? SPRC_K_GEMM_BF16 : SPRC_K_GEMM_F32, st, 2.0 * a->M * (double)a->N * a->K,
                   ((double)a->M * a->K + (double)a->N * a->K) * es + (double)a->M * a->N * (osz + (a->resid ? 4.0 : 0.0)));
    return a->dtype == SPRC_BF16 ? dispatch<bf16_t>(a, p, st) : dispatch<float>(a, p, st);
}

extern "C" int sprc_sim_max(const void* fusion, const void* feats, float* sim, int64_t ld_sim, int32_t nq, int32_t N,
                            int32_t E, int32_t dtype, sprc_stream s) {
    sprc_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.M = nq; g.N = N * 32; g.K = E;
    g.dtype = dtype; g.out_dtype = SPRC_F32; g.act = SPRC_ACT_NONE; g.max32 = 1;
    g.A = fusion; g.lda = E;
    g.W = feats; g.ldw = E;
    g.C = sim; g.ldc = ld_sim;
    return sprc_gemm(&g, s);
}
