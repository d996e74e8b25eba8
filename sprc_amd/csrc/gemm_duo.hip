// gemm_duo.hip -- instantiations and launcher of the duo GEMM kernel (gemm_duo.hpp: 256 x 128 tiles, two free-running workgroups per CU).
// An A/B variant (SPRC_GEMM_DUO=1), OFF by default: measured 24 % behind the anti-phase kernel (negative results of round 5).
// Only the epilogues the ViT uses are instantiated (16-bit output with / without GELU, fp32 output); anything else falls back.
#include <mutex>

#include "gemm_impl.hpp"

namespace sprc {
#include "gemm_duo.hpp"

static int* duo_counters(hipStream_t st) {  // per device, zeroed once ON THE LAUNCH STREAM (ordered before the first launch that counts on them); the
    static int* ctr[MAX_DEVICES] = {nullptr};   // kernels only ever increment them.  One allocation per device for the life of the process.
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    const int dev = current_device();
    if (ctr[dev] == nullptr) {
        int* c = nullptr;
        if (hipMalloc(&c, DUO_CTRS * sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemsetAsync(c, 0, DUO_CTRS * sizeof(int), st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipFree(c); return nullptr; }
        ctr[dev] = c;
    }
    return ctr[dev];
}
template <typename T, typename OutT, int ACT>
static int launch_duo(GemmParams p, hipStream_t st) {
    auto kern = gemm_duo_kernel<T, OutT, ACT>;
    static bool attr_set[MAX_DEVICES] = {false};
    constexpr int LDS_REQ = (SPRC_DUO_ABL & 4) ? 100 * 1024 : DUO_LDS;
    optin_lds(kern, LDS_REQ, attr_set);
    p.tiles_m = (p.M + DUO_BM - 1) / DUO_BM;
    p.tiles_n = (p.N + DUO_BN - 1) / DUO_BN;
    static const int order = env_int("SPRC_DUO_ORDER", 8);      // W-resident groups of 8 n-tiles (the same 1024 W rows as the 256-wide groups of 4)
    p.order = order;
    p.nwg0 = p.tiles_m * p.tiles_n;
    const int total = p.nwg0 * (p.dual ? 2 : 1);
    const int slots = ((SPRC_DUO_ABL & 4) ? 1 : 2) * num_cus(st);
    // stagger: (W + E) / 2 with W = the K loop of a tile alone on the matrix pipe (16 MFMAs x 32 cycles per K-tile) and E ~ the epilogue
    static const int sleep_env = env_int("SPRC_DUO_SLEEP", -1);
    const int nt = (int)((int64_t)p.K * 2 / DUO_KTB);
    p.duo_sleep = sleep_env >= 0 ? sleep_env : (total > num_cus(st) ? nt * 256 + 4000 : 0);
    p.duo_ctr = duo_counters(st);
    if (p.duo_ctr == nullptr) p.duo_sleep = 0;
    hipLaunchKernelGGL(kern, dim3(total < slots ? total : slots), dim3(256), LDS_REQ, st, p);
    SPRC_CHECK_LAUNCH("sprc_gemm(duo)");
    return SPRC_OK;
}


template <typename T>
static int duo_dispatch(int out_kind, int act, const GemmParams& p, hipStream_t st) {
    typedef T O16;
    if (out_kind == 0) {
        switch (act) {
            case SPRC_ACT_NONE: return launch_duo<T, O16, SPRC_ACT_NONE>(p, st);
            case SPRC_ACT_GELU: return launch_duo<T, O16, SPRC_ACT_GELU>(p, st);
        }
    } else if (out_kind == 1) {
        switch (act) {
            case SPRC_ACT_NONE: return launch_duo<T, float, SPRC_ACT_NONE>(p, st);
        }
    }
    return SPRC_EUNSUPPORTED;
}

int gemm_duo_launch(bool f16, int out_kind, int act, const GemmParams& p, hipStream_t st) {
#ifdef SPRC_DUO_FAST
    if (!f16) return SPRC_EUNSUPPORTED;
#else
    if (!f16) return duo_dispatch<bf16_t>(out_kind, act, p, st);
#endif
    return duo_dispatch<f16_t>(out_kind, act, p, st);
}
}  // namespace sprc
