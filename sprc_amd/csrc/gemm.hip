// gemm.hip -- C entry points of the GEMM (sprc_gemm, sprc_gemm_pair, sprc_sim_max) and the bf16-operand instantiations.
// Kernels and launchers: gemm_impl.hpp.
#include "gemm_impl.hpp"

namespace sprc {
int gemm_dispatch_bf16(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st) { return dispatch<bf16_t>(a, p, st); }
}  // namespace sprc

static int gemm_impl(const sprc_gemm_args* a, const sprc_gemm_args* b, sprc_stream s) {
    using namespace sprc;
    SPRC_REQUIRE(a != nullptr, "sprc_gemm: null args");
    SPRC_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "sprc_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    SPRC_REQUIRE(a->dtype == SPRC_BF16 || a->dtype == SPRC_F16 || a->dtype == SPRC_F32 || a->dtype == SPRC_FP8, "sprc_gemm: bad dtype %d", a->dtype);
    SPRC_REQUIRE(a->out_dtype == SPRC_BF16 || a->out_dtype == SPRC_F32 || a->out_dtype == SPRC_F16 || a->out_dtype == SPRC_FP8 ||
                     a->out_dtype == SPRC_F16X3, "sprc_gemm: bad out_dtype %d", a->out_dtype);
    SPRC_REQUIRE(a->out_dtype != SPRC_F16X3 || (a->dtype == SPRC_F16 && !a->resid && !a->max32 && a->act != SPRC_ACT_QUICKGELU &&
                                                a->N % 4 == 0 && a->ldc >= 2 * (int64_t)a->N),
                 "sprc_gemm: SPRC_F16X3 output takes fp16 operands, N %% 4 == 0, ldc >= 2 N (fp16 units), no residual / max32 / QuickGELU");
    SPRC_REQUIRE(a->k8 >= 0 && (a->k8 == 0 || (a->dtype == SPRC_F16 && !a->max32 && a->K % 128 == 0 && a->k8 % 128 == 0 && a->k8 >= 512 &&
                                               a->lda >= a->K + a->k8 / 2 && a->ldw >= a->K + a->k8 / 2)),
                 "sprc_gemm(k8): a split-precision product takes fp16 operands, K %% 128 == 0, k8 %% 128 == 0, k8 >= 512, lda / ldw >= K + k8 / 2, no max32");
    SPRC_REQUIRE(a->dtype != SPRC_FP8 || (a->w_scale != nullptr && a->a_scale > 0.f && !a->max32 && b == nullptr),
                 "sprc_gemm(fp8): needs w_scale, a_scale > 0; no max32 / paired launch");
    SPRC_REQUIRE(a->out_dtype != SPRC_FP8 || a->out_scale > 0.f, "sprc_gemm: SPRC_FP8 output needs out_scale > 0");
    SPRC_REQUIRE(a->out_dtype != SPRC_F16 || a->dtype == SPRC_F16 ||
                     ((a->dtype == SPRC_BF16 || a->dtype == SPRC_FP8) && a->act == SPRC_ACT_NONE && !a->resid && !a->max32),
                 "sprc_gemm: SPRC_F16 output takes fp16 operands, or bf16 / fp8 operands and a plain (bias-only) epilogue");
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
    if (!fits_u32(p)) {
        set_error("sprc_gemm: a 256-row tile of A or W spans >= 4 GiB, or a row stride >= 16 MiB (lda=%lld ldw=%lld)", (long long)a->lda, (long long)a->ldw);
        return SPRC_EUNSUPPORTED;
    }
    p.tiles_m = p.tiles_n = 0;
    p.ksplit = 1; p.split_stride = 0;
    p.dual = 0; p.nwg0 = 0; p.W2 = nullptr; p.bias2 = nullptr; p.a_off2 = p.c_off2 = 0;
    if (b != nullptr) {                 // second product of a paired launch: same shapes, operands A / C / resid and row-map geometry
        SPRC_REQUIRE(!a->max32 && !b->max32, "sprc_gemm_pair: no max32 epilogue");
        SPRC_REQUIRE(b->M == a->M && b->N == a->N && b->K == a->K && b->dtype == a->dtype && b->out_dtype == a->out_dtype &&
                         b->act == a->act && b->A == a->A && b->lda == a->lda && b->ldw == a->ldw && b->C == a->C && b->k8 == a->k8 &&
                         b->ldc == a->ldc && b->resid == a->resid && b->ldr == a->ldr &&
                         b->amap.rows_per_group == a->amap.rows_per_group && b->amap.group_stride == a->amap.group_stride &&
                         b->cmap.rows_per_group == a->cmap.rows_per_group && b->cmap.group_stride == a->cmap.group_stride,
                     "sprc_gemm_pair: the two products may differ in W, bias and the row-map offsets only");
        SPRC_REQUIRE(b->W != nullptr && ((uintptr_t)b->W % 16) == 0, "sprc_gemm_pair: bad second W");
        SPRC_REQUIRE((a->bias == nullptr) == (b->bias == nullptr), "sprc_gemm_pair: bias on both products or on neither");
        p.dual = 1;
        p.W2 = (const char*)b->W; p.bias2 = b->bias;
        p.a_off2 = b->amap.group_offset; p.c_off2 = b->cmap.group_offset;
    }
    p.scratch = reinterpret_cast<float*>(a->scratch); p.scratch_elems = (int64_t)(a->scratch_bytes / 4);
    p.w_scale = a->dtype == SPRC_FP8 ? a->w_scale : nullptr; p.a_scale = a->a_scale; p.out_scale = a->out_scale;
    static const int dbg = env_int("SPRC_GEMM_DEBUG", 0) | (env_int("SPRC_GEMM_DEAD", 1) ? 0 : 128);     // bit 128: dead waves multiply like live ones (A/B)
    p.debug = dbg;
    p.duo_sleep = 0; p.duo_ctr = nullptr;
    static const int epi_wide = env_int("SPRC_EPI_WIDE", 1);       // 0: 4 columns per lane in every epilogue (A/B switch, gemm_epilogue)
    p.epi_wide = epi_wide;
    p.order = -1;                       // per-kernel default (launch_*), SPRC_GEMM_ORDER overrides
    p.k8 = a->k8;
    hipStream_t st = (hipStream_t)s;
    const double osz = a->max32 ? 4.0 / 32.0 : a->out_dtype == SPRC_F16X3 ? 4.0 : (double)dtype_size(a->out_dtype);     // a split row: 4 bytes per column
    const double np = b != nullptr ? 2.0 : 1.0;
    // algorithmic work: the product the caller MEANS (k_alg: a split-precision launch reduces over K = 3 k_alg, the patch embedding
    // over zero padding); executed flops are recorded next to it
    SPRC_REQUIRE(a->k_alg >= 0 && a->k_alg <= a->K, "sprc_gemm: k_alg=%d outside [0, K=%d]", a->k_alg, a->K);
    const double ka = a->k_alg > 0 ? (double)a->k_alg : (double)a->K;
    ProfScope prof(a->dtype == SPRC_F32 ? SPRC_K_GEMM_F32 : SPRC_K_GEMM_BF16, st, np * 2.0 * a->M * (double)a->N * ka,
                   np * (((double)a->M * ka + (double)a->N * ka) * es + (double)a->M * a->N * (osz + (a->resid ? 4.0 : 0.0))),
                   np * 2.0 * a->M * (double)a->N * ((double)a->K + a->k8));     // executed MACs: fp16 ones + e4m3 correction ones (half the time each)
    if (a->k8 > 0) return gemm_dispatch_f16e(a, p, st);
    if (a->dtype == SPRC_FP8) return gemm_dispatch_fp8(a, p, st);
    if (a->dtype == SPRC_F16) return gemm_dispatch_f16(a, p, st);
    return a->dtype == SPRC_BF16 ? gemm_dispatch_bf16(a, p, st) : gemm_dispatch_f32(a, p, st);
}

extern "C" int sprc_gemm(const sprc_gemm_args* a, sprc_stream s) { return gemm_impl(a, nullptr, s); }

extern "C" int sprc_gemm_pair(const sprc_gemm_args* a, const sprc_gemm_args* b, sprc_stream s) {
    SPRC_REQUIRE(a != nullptr && b != nullptr, "sprc_gemm_pair: null args");
    return gemm_impl(a, b, s);
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
