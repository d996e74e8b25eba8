// rank.hip -- ranking stage (R7): wavefront bitonic top-k and exact rank counting.
//
// The ranking contract (oracle.rank_stable): order gallery items by the key
//   ( fl32(1 - sim) , index )  ascending
// i.e. a STABLE argsort of the fp32 distance the reference computes (validate_blip.py:253-254).
// Keys are built as 64-bit integers (order-preserving image of the fp32 distance in the high word,
// the index in the low word), so every comparison is an integer comparison and the result is
// bit-exact and independent of how the row is split across waves, workgroups or GPUs.
//
// sprc_topk: one 256-thread workgroup per query row.  Each of the 4 waves streams 64-element
// chunks of the row (coalesced 256-B reads), skips a chunk when none of its keys beats the wave's
// current 64th-best (one ballot), otherwise bitonic-sorts the chunk across lanes (21 compare-
// exchange stages on shuffles) and bitonic-merges it into its sorted running top-64 held one key
// per lane.  The 4 per-wave lists are merged through LDS.  HBM traffic = the sim row once.
// sprc_rank_of: position of listed gallery items in that order by counting smaller keys (no sort).
#include "common.hpp"

namespace sprc {

__device__ __forceinline__ uint32_t dist_bits(float sim) {
    const float d = 1.0f - sim;                      // the reference's `1 - pred_sim`, fp32
    if (d != d) return 0xffffffffu;                  // NaN sorts last (torch.argsort convention)
    uint32_t u = __float_as_uint(d + 0.0f);          // -0.0 -> +0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Item {
    uint64_t key;      // (dist_bits << 32) | index
    uint32_t pos;      // position inside the row (to fetch the original sim at the end)
};

__device__ __forceinline__ Item shfl_item(const Item& a, int src) {
    Item r;
    const uint32_t lo = __shfl((uint32_t)a.key, src, 64);
    const uint32_t hi = __shfl((uint32_t)(a.key >> 32), src, 64);
    r.key = ((uint64_t)hi << 32) | lo;
    r.pos = __shfl(a.pos, src, 64);
    return r;
}

// compare-exchange with lane^j; keep_min: this lane keeps the smaller key
__device__ __forceinline__ void cmpxchg(Item& a, int lane, int j, bool keep_min) {
    const Item o = shfl_item(a, lane ^ j);
    const bool take = keep_min ? (o.key < a.key) : (o.key > a.key);
    if (take) a = o;
}

__device__ __forceinline__ void bitonic_sort64(Item& a, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool asc = (lane & k) == 0;                // k == 64 -> all ascending
            const bool lower = (lane & j) == 0;
            cmpxchg(a, lane, j, asc == lower);
        }
    }
}

// `a` holds a bitonic sequence across lanes -> ascending
__device__ __forceinline__ void bitonic_merge64(Item& a, int lane) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) cmpxchg(a, lane, j, (lane & j) == 0);
}

// top (sorted asc) <- 64 smallest of top U other (other sorted asc)
__device__ __forceinline__ void merge_keep_low(Item& top, const Item& other, int lane) {
    const Item rev = shfl_item(other, 63 - lane);
    if (rev.key < top.key) top = rev;
    bitonic_merge64(top, lane);
}

__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ sim, int64_t ld, const int32_t* __restrict__ gidx,
                                                   int idx_base, int N, int k, float* __restrict__ out_sim,
                                                   int32_t* __restrict__ out_idx) {
    __shared__ uint64_t s_key[4][64];
    __shared__ uint32_t s_pos[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = blockIdx.x;
    const float* row = sim + q * ld;
    const int32_t* grow = gidx ? gidx + q * (int64_t)N : nullptr;

    Item top;
    top.key = ~0ull;
    top.pos = 0xffffffffu;
    const int nchunks = (N + 63) >> 6;
    for (int c = wave; c < nchunks; c += 4) {
        const int n = c * 64 + lane;
        Item it;
        it.key = ~0ull;
        it.pos = 0xffffffffu;
        if (n < N) {
            const uint32_t id = grow ? (uint32_t)grow[n] : (uint32_t)(n + idx_base);
            it.key = ((uint64_t)dist_bits(row[n]) << 32) | id;
            it.pos = (uint32_t)n;
        }
        const uint64_t worst = shfl_item(top, 63).key;
        if (!__any(it.key < worst)) continue;                // nothing in this chunk enters the top-64
        bitonic_sort64(it, lane);
        merge_keep_low(top, it, lane);
    }
    s_key[wave][lane] = top.key;
    s_pos[wave][lane] = top.pos;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            Item o;
            o.key = s_key[w][lane];
            o.pos = s_pos[w][lane];
            merge_keep_low(top, o, lane);
        }
        if (lane < k) {
            const bool ok = top.pos != 0xffffffffu;
            out_idx[q * k + lane] = ok ? (int32_t)(uint32_t)top.key : -1;
            out_sim[q * k + lane] = ok ? row[top.pos] : -INFINITY;
        }
    }
}

constexpr int RANK_L = 16;
__global__ __launch_bounds__(256) void rank_of_kernel(const float* __restrict__ sim, int64_t ld, const int32_t* __restrict__ listed,
                                                      int N, int L, int32_t* __restrict__ rank) {
    __shared__ uint64_t s_t[RANK_L];
    __shared__ int s_cnt[RANK_L];
    const int64_t q = blockIdx.x;
    const float* row = sim + q * ld;
    for (int l0 = 0; l0 < L; l0 += RANK_L) {
        const int nl = min(RANK_L, L - l0);
        if ((int)threadIdx.x < nl) {
            const int t = listed[q * L + l0 + threadIdx.x];
            s_t[threadIdx.x] = (t >= 0 && t < N) ? (((uint64_t)dist_bits(row[t]) << 32) | (uint32_t)t) : 0ull;
            s_cnt[threadIdx.x] = 0;
        }
        __syncthreads();
        int cnt[RANK_L];
#pragma unroll
        for (int l = 0; l < RANK_L; ++l) cnt[l] = 0;
        for (int n = threadIdx.x; n < N; n += 256) {
            const uint64_t key = ((uint64_t)dist_bits(row[n]) << 32) | (uint32_t)n;
#pragma unroll
            for (int l = 0; l < RANK_L; ++l)
                if (l < nl) cnt[l] += (key < s_t[l]) ? 1 : 0;
        }
#pragma unroll
        for (int l = 0; l < RANK_L; ++l) {
            if (l < nl) {
                int v = cnt[l];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt[l], v);
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < nl) {
            const int t = listed[q * L + l0 + threadIdx.x];
            rank[q * L + l0 + threadIdx.x] = (t >= 0 && t < N) ? s_cnt[threadIdx.x] : -1;
        }
        __syncthreads();
    }
}

}  // namespace sprc

using namespace sprc;

extern "C" int sprc_topk(const float* sim, int64_t ld, const int32_t* gidx, int32_t idx_base, int32_t nq, int32_t N,
                         int32_t k, float* out_sim, int32_t* out_idx, sprc_stream s) {
    SPRC_REQUIRE(sim && out_sim && out_idx, "sprc_topk: null pointer");
    SPRC_REQUIRE(nq > 0 && N > 0 && ld >= N, "sprc_topk: bad shape nq=%d N=%d ld=%lld", nq, N, (long long)ld);
    SPRC_REQUIRE(k >= 1 && k <= 64, "sprc_topk: k=%d must be in [1,64]", k);
    ProfScope prof(SPRC_K_RANK, (hipStream_t)s, 0.0, (double)nq * N * (gidx ? 8.0 : 4.0) + (double)nq * k * 8.0);
    hipLaunchKernelGGL(topk_kernel, dim3(nq), dim3(256), 0, (hipStream_t)s, sim, ld, gidx, idx_base, N, k, out_sim, out_idx);
    SPRC_CHECK_LAUNCH("sprc_topk");
    return SPRC_OK;
}

extern "C" int sprc_rank_of(const float* sim, int64_t ld, const int32_t* listed, int32_t nq, int32_t N, int32_t L,
                            int32_t* rank, sprc_stream s) {
    SPRC_REQUIRE(sim && listed && rank, "sprc_rank_of: null pointer");
    SPRC_REQUIRE(nq > 0 && N > 0 && L > 0 && ld >= N, "sprc_rank_of: bad shape");
    ProfScope prof(SPRC_K_RANK, (hipStream_t)s, 0.0, (double)nq * N * 4.0);
    hipLaunchKernelGGL(rank_of_kernel, dim3(nq), dim3(256), 0, (hipStream_t)s, sim, ld, listed, N, L, rank);
    SPRC_CHECK_LAUNCH("sprc_rank_of");
    return SPRC_OK;
}
