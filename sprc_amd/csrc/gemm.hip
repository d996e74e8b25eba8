// gemm.hip -- C = epilogue(A[M,K] . W[N,K]^T) on gfx950 matrix cores.
//
// Replaces every nn.Linear / F.linear / conv-as-GEMM on the SPRC retrieval path (see sprc.h).
// Both operands are K-contiguous ("B^T" form), so A and W fragments are read the same way.
//
// Tile: 128 x 128 x (128 bytes of K) per 256-thread workgroup = 4 waves in a 2x2 grid, each wave a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 fp32 acc regs / lane).
//   bf16: v_mfma_f32_32x32x16_bf16, K-tile = 64 elements, 16 MFMA / wave / K-tile
//   f32 : v_mfma_f32_32x32x2_f32   (exact fp32, parity mode), K-tile = 32 elements
// LDS: 2 stages x (A 16 KiB + W 16 KiB) = 64 KiB -> 2 workgroups / CU.  Rows are 128 B; the 16-B slot
// index is XOR-swizzled with (row>>1)&7 so every ds_read_b128 lane group hits 16 distinct slots of
// the 256-B bank row (guide T2); staging goes HBM -> VGPR -> LDS (ds_write_b128) with the next
// K-tile's global loads issued before the current tile's MFMAs (guide T14).
// Grid: 1-D, XCD-aware remap (block b runs on XCD b%8 -> each XCD gets a contiguous tile range) and
// an 8-row grouped tile order so concurrently resident tiles share A/W panels in the XCD's L2.
#include "common.hpp"

namespace sprc {

constexpr int BM = 128, BN = 128, KT_BYTES = 128, STAGE_BYTES = (BM + BN) * KT_BYTES;

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

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ void mma_tile(const char* sA, const char* sB, int lane, int wr, int wc,
                                         f32x16 (&acc)[2][2]) {
    const int r32 = lane & 31, half = lane >> 5;
    const int sw = (r32 >> 1) & 7;
    const char* pa = sA + (wr * 64 + r32) * KT_BYTES;
    const char* pb = sB + (wc * 64 + r32) * KT_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int slot = ((kk * 2 + half) ^ sw) << 4;
        typename Frag<T>::type a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = *reinterpret_cast<const typename Frag<T>::type*>(pa + i * 32 * KT_BYTES + slot);
            b[i] = *reinterpret_cast<const typename Frag<T>::type*>(pb + i * 32 * KT_BYTES + slot);
        }
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        } else {
            // each lane holds 4 consecutive k of its half; MFMA step t pairs k = {8kk+t, 8kk+4+t}
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
        }
    }
}

template <typename T, typename OutT, int ACT, bool MAX32>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- tile id: XCD-contiguous remap (bijective for any grid), then grouped (8 m-tiles) order ----
    const int nwg = p.tiles_m * p.tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int GROUP_M = 8;
    const int in_group = GROUP_M * p.tiles_n;
    const int group_id = pid / in_group;
    const int first_m = group_id * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int pid_m = first_m + (pid % in_group) % gsz;
    const int pid_n = (pid % in_group) / gsz;
    const int m0 = pid_m * BM, n0 = pid_n * BN;

    // ---- staging assignment: thread -> 4 (row, 16-B slot) chunks of A and of W ----
    const char* a_src[4];
    const char* w_src[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 256 + tid, row = c >> 3, slot = c & 7;
        const int am = min(m0 + row, p.M - 1), wn = min(n0 + row, p.N - 1);
        a_src[i] = p.A + map_row_s(p.a_shift, p.a_stride, p.a_off, am) * p.lda_b + slot * 16;
        w_src[i] = p.W + (int64_t)wn * p.ldw_b + slot * 16;
        lds_off[i] = row * KT_BYTES + ((slot ^ ((row >> 1) & 7)) << 4);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nt = (int)(((int64_t)p.K * sizeof(T)) / KT_BYTES);
    u32x4 ra[4], rw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const u32x4*>(a_src[i]);
        rw[i] = *reinterpret_cast<const u32x4*>(w_src[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<u32x4*>(smem + lds_off[i]) = ra[i];
        *reinterpret_cast<u32x4*>(smem + BM * KT_BYTES + lds_off[i]) = rw[i];
    }
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1) < nt;
        if (more) {
            const int64_t ko = (int64_t)(t + 1) * KT_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const u32x4*>(a_src[i] + ko);
                rw[i] = *reinterpret_cast<const u32x4*>(w_src[i] + ko);
            }
        }
        const char* sA = smem + cur * STAGE_BYTES;
        mma_tile<T>(sA, sA + BM * KT_BYTES, lane, wr, wc, acc);
        if (more) {
            char* dst = smem + (cur ^ 1) * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<u32x4*>(dst + lds_off[i]) = ra[i];
                *reinterpret_cast<u32x4*>(dst + BM * KT_BYTES + lds_off[i]) = rw[i];
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    const int r32 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wc * 64 + ni * 32 + r32;
            const int rbase = m0 + wr * 64 + mi * 32;
            if constexpr (MAX32) {
                float v = acc[mi][ni][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[mi][ni][r]);
                v = fmaxf(v, __shfl_xor(v, 32, 64));
                if (half == 0 && col < p.N && rbase < p.M)
                    reinterpret_cast<float*>(p.C)[(int64_t)col * p.ldc + (rbase >> 5)] = v;
            } else {
                const float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < p.M && col < p.N) {
                        float v = acc[mi][ni][r] + bv;
                        if constexpr (ACT == SPRC_ACT_GELU) v = gelu_erf(v);
                        if constexpr (ACT == SPRC_ACT_QUICKGELU) v = quick_gelu(v);
                        const int64_t prow = map_row_s(p.c_shift, p.c_stride, p.c_off, row);
                        if (p.resid != nullptr) v += p.resid[prow * p.ldr + col];
                        if constexpr (sizeof(OutT) == 2)
                            reinterpret_cast<uint16_t*>(p.C)[prow * p.ldc + col] = f32_to_bf16_bits(v);
                        else
                            reinterpret_cast<float*>(p.C)[prow * p.ldc + col] = v;
                    }
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

template <typename T, typename OutT, int ACT, bool MAX32>
static int launch(const GemmParams& p, hipStream_t st) {
    auto kern = gemm_kernel<T, OutT, ACT, MAX32>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * STAGE_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(256), 2 * STAGE_BYTES, st, p);
    SPRC_CHECK_LAUNCH("sprc_gemm");
    return SPRC_OK;
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
    SPRC_REQUIRE(((int64_t)a->K * es) % KT_BYTES == 0, "sprc_gemm: K=%d must be a multiple of %d", a->K, KT_BYTES / es);
    SPRC_REQUIRE((a->lda * es) % 16 == 0 && (a->ldw * es) % 16 == 0, "sprc_gemm: lda/ldw must be 16-byte multiples");
    SPRC_REQUIRE(a->lda >= a->K && a->ldw >= a->K, "sprc_gemm: leading dimension < K");
    SPRC_REQUIRE(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0, "sprc_gemm: A/W must be 16-byte aligned");
    SPRC_REQUIRE(a->A && a->W && a->C, "sprc_gemm: null operand");
    if (a->max32) {
        SPRC_REQUIRE(a->M % 32 == 0, "sprc_gemm(max32): M=%d must be a multiple of 32", a->M);
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
    p.tiles_m = (a->M + BM - 1) / BM;
    p.tiles_n = (a->N + BN - 1) / BN;
    hipStream_t st = (hipStream_t)s;
    return a->dtype == SPRC_BF16 ? dispatch<bf16_t>(a, p, st) : dispatch<float>(a, p, st);
}

extern "C" int sprc_sim_max(const void* fusion, const void* feats, float* sim, int64_t ld_sim, int32_t nq, int32_t N,
                            int32_t E, int32_t dtype, sprc_stream s) {
    sprc_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.M = N * 32; g.N = nq; g.K = E;
    g.dtype = dtype; g.out_dtype = SPRC_F32; g.act = SPRC_ACT_NONE; g.max32 = 1;
    g.A = feats; g.lda = E;
    g.W = fusion; g.ldw = E;
    g.C = sim; g.ldc = ld_sim;
    return sprc_gemm(&g, s);
}
