// attention.hip -- out = softmax(scale * Q K^T + key_mask) V  per (batch, head), no S x S matrix in HBM.
//
// bf16 kernel (MFMA, flash-style), one workgroup per (batch, head):
//   * K  [Tk_pad][DHP] and V^T [DHP][Tk_pad] of the head are staged ONCE in LDS (Tk <= 257 on this
//     path: 55 KiB + 55 KiB for ViT-g), padded row strides (DHP*2+16 B, Tk_pad*2+8 B) make the
//     ds_read_b128 / ds_read_b64 fragment reads bank-conflict free.
//   * every wave owns 32-row query tiles.  Per 32-key tile it computes the TRANSPOSED scores
//     S^T = K . Q^T with v_mfma_f32_32x32x16_bf16, so a lane holds 16 keys of ONE query column:
//     row max/sum are in-register plus one cross-half shuffle (guide: "swapped QK^T").
//   * O^T = V^T . P^T keeps that query-per-lane layout, so the online-softmax rescale is a per-lane
//     scalar and P never leaves registers: the MFMA k-index is an arbitrary permutation of the
//     keys as long as P and V use the same one, which removes the permlane/LDS round trip.
//   * head_dim 88 (EVA ViT-g) is zero-padded to 96 = 6 MFMA k-steps; Tk is padded to a multiple of
//     32 with -inf scores.
// f32 kernel: exact-fp32 VALU restatement for parity mode (one wave per query row).
#include <type_traits>

#include "common.hpp"
#ifndef SPRC_ATTN_NT
#define SPRC_ATTN_NT 0
#endif

namespace sprc {

constexpr float LOG2E = 1.4426950408889634f;

// sum of the two bf16 values packed in a dword, as floats.  The softmax denominator is accumulated from the ROUNDED
// probabilities that enter the P.V MFMAs: numerator and denominator then carry the same rounding, and the error of the
// weighted average scales with |v - mean(v)| instead of |v| (first order: sum_k p_k eps_k (v_k - vbar) / sum_k p_k).
template <bool F16>
__device__ __forceinline__ float rounded_pair_sum(uint32_t pk) {
    if constexpr (F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
        const f16x2_t v = __builtin_bit_cast(f16x2_t, pk);
        return (float)v[0] + (float)v[1];
    } else {
        return __uint_as_float(pk << 16) + __uint_as_float(pk & 0xffff0000u);
    }
}
// one MFMA k-step on 16-bit fragments of the engine's operand type (F16: IEEE half, else bf16): same layout, same rate
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct AttnParams {
    int B, H, Tq, Tk, dh;
    const char* q; int64_t ldq;     // leading dims in ELEMENTS
    const char* k; int64_t ldk;
    const char* v; int64_t ldv;
    char* out; int64_t ldo;
    const float* key_mask;
    float scale;
    // optional second key/value segment (stage-2 rerank: keys = cat(reference tokens, candidate tokens)): keys [0, Tk1) come
    // from (k, v) at batch row idx1[b] (b when null), keys [Tk1, Tk) from (k2, v2) at batch row idx2[b]; Tk1 == Tk: one segment
    int Tk1;
    const char* k2; int64_t ldk2;
    const char* v2; int64_t ldv2;
    const int32_t* idx1; const int32_t* idx2;
    int x3;                         // > 0 (fp16, resident kernel): `out` rows in the SPRC_F16X3 layout, x3 = logical row width H * dh
    uint32_t drop_thresh, drop_site; uint64_t drop_seed; float drop_scale;    // fp32 kernel, training: probability dropout (thresh 0: none)
};

// byte address of head h of key/value token t of batch b (t in the concatenated key axis)
__device__ __forceinline__ const char* kv_token(const AttnParams& p, const char* seg1, int64_t ld1, const char* seg2, int64_t ld2,
                                                int b, int h, int t) {
    if (t < p.Tk1) {
        const int64_t row = (int64_t)(p.idx1 ? p.idx1[b] : b) * p.Tk1 + t;
        return seg1 + (row * ld1 + (int64_t)h * p.dh) * 2;
    }
    const int64_t row = (int64_t)(p.idx2 ? p.idx2[b] : b) * (p.Tk - p.Tk1) + (t - p.Tk1);
    return seg2 + (row * ld2 + (int64_t)h * p.dh) * 2;
}

// ------------------------------------------------------------------------------------------------
template <int DHP, int NW, bool F16>
__global__ __launch_bounds__(64 * NW) void attn_bf16_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = DHP / 16;              // MFMA k-steps of QK^T
    constexpr int DT = DHP / 32;              // 32-wide tiles of the head dim in O^T
    constexpr int KROW = DHP * 2 + 16;        // bytes per K row in LDS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, half = lane >> 5;
    int b, h;                                 // all heads of an image on one XCD (block i runs on XCD i % 8): see attn_stream_kernel
    if ((p.B & 7) == 0) { const int x = blockIdx.x & 7, i = blockIdx.x >> 3; h = i % p.H; b = (i / p.H) * 8 + x; }
    else { b = blockIdx.x / p.H; h = blockIdx.x % p.H; }
    const int Tkp = (p.Tk + 31) & ~31;
    const int VROW = Tkp * 2 + 8;             // bytes per V^T row in LDS
    char* sK = smem;
    char* sV = sK + (size_t)Tkp * KROW;
    float* sM = reinterpret_cast<float*>(sV + (size_t)DHP * VROW);

    const int dh = p.dh;
    const int nqt = (p.Tq + 31) >> 5;
    // Q^T fragment (B operand) of query tile qt: lane (q = r32, half) holds Q[q][ks*16 + half*8 .. +8].  The wave's FIRST tile is
    // fetched here, before the K / V staging, so its memory round trip runs under the staging instead of after the barrier
    // (the Q-Former launches are a few tiles per workgroup: that round trip was a fifth of their time).
    auto load_q = [&](int qt, u32x4 (&qf)[KS]) {
        const int qrow = min(qt * 32 + r32, p.Tq - 1);
        const char* qptr = p.q + (((int64_t)b * p.Tq + qrow) * p.ldq + (int64_t)h * dh) * 2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 16 + half * 8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(qptr + (d0 < dh ? d0 : 0) * 2);
            qf[ks] = d0 < dh ? v : u32x4{0u, 0u, 0u, 0u};
        }
    };
    u32x4 qf[KS];
    load_q(min(wave, nqt - 1), qf);

    // ---- stage K (row-major, zero padded) and V transposed (VT[d][key], two keys per 32-bit write).  Global loads are
    // issued in batches of UNR independent requests per thread before any LDS write, so a batch costs one memory
    // latency instead of UNR (PMC before: 51 % of wave-cycles in s_waitcnt). ----
    constexpr int NTH = 64 * NW, UNR = 8, CPR = DHP / 8;          // 16-B chunks per row
    const int nK = Tkp * CPR;
    for (int it0 = tid; it0 < nK; it0 += NTH * UNR) {
        u32x4 val[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {             // branch-free: clamped address, the value masked afterwards (predicated loads
            const int it = min(it0 + u * NTH, nK - 1), t = it / CPR, c = it % CPR;     // make hipcc drain the queue where they meet)
            const bool ok = t < p.Tk && c * 8 < dh;
            const u32x4 v = *reinterpret_cast<const u32x4*>(kv_token(p, p.k, p.ldk, p.k2, p.ldk2, b, h, min(t, p.Tk - 1)) + (ok ? c : 0) * 16);
            val[u] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int it = it0 + u * NTH, t = it / CPR, c = it % CPR;
            if (it < nK) *reinterpret_cast<u32x4*>(sK + t * KROW + c * 16) = val[u];
        }
    }
    const int nV = (Tkp / 2) * CPR;
    for (int it0 = tid; it0 < nV; it0 += NTH * (UNR / 2)) {
        u32x4 va[UNR / 2], vb[UNR / 2];
#pragma unroll
        for (int u = 0; u < UNR / 2; ++u) {
            const int it = min(it0 + u * NTH, nV - 1), kp = it % (Tkp / 2), c = it / (Tkp / 2), t0 = kp * 2;
            const bool okc = c * 8 < dh;
            const int cc = okc ? c : 0;
            const u32x4 a = *reinterpret_cast<const u32x4*>(kv_token(p, p.v, p.ldv, p.v2, p.ldv2, b, h, min(t0, p.Tk - 1)) + cc * 16);
            const u32x4 bq = *reinterpret_cast<const u32x4*>(kv_token(p, p.v, p.ldv, p.v2, p.ldv2, b, h, min(t0 + 1, p.Tk - 1)) + cc * 16);
            va[u] = (okc && t0 < p.Tk) ? a : u32x4{0u, 0u, 0u, 0u};
            vb[u] = (okc && t0 + 1 < p.Tk) ? bq : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < UNR / 2; ++u) {
            const int it = it0 + u * NTH, kp = it % (Tkp / 2), c = it / (Tkp / 2), t0 = kp * 2;
            if (it < nV) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t lo = (va[u][e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                    const uint32_t hi = (vb[u][e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                    *reinterpret_cast<uint32_t*>(sV + (c * 8 + e) * VROW + t0 * 2) = lo | (hi << 16);
                }
            }
        }
    }
    // ---- additive mask in the log2 domain; -inf on padded keys ----
    for (int t = tid; t < Tkp; t += 64 * NW) {
        float m = -INFINITY;
        if (t < p.Tk) m = p.key_mask ? p.key_mask[(int64_t)b * p.Tk + t] * LOG2E : 0.f;
        sM[t] = m;
    }
    __syncthreads();

    const float sc = p.scale * LOG2E;
    const int nkt = Tkp >> 5;
    const bool plain_tail = (p.Tk & 31) == 0;
    for (int qt = wave; qt < nqt; qt += NW) {
        if (qt != wave) load_q(qt, qf);
        f32x16 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;

        for (int kt = 0; kt < nkt; ++kt) {
            // S^T tile: rows = keys, col = query
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const char* krow = sK + (kt * 32 + r32) * KROW + half * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(krow + ks * 32);
                s = mfma16<F16>(kf, qf[ks], s);
            }
            float m_new, psum = 0.f;
            if (p.key_mask == nullptr && (plain_tail || kt + 1 < nkt)) {
                // no mask and no padded key in this tile: p = exp2(s*sc - m) with ONE fma per score
                float mx = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                m_new = fmaxf(m_run, mx * sc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -m_new));
                    psum += s[r];
                }
            } else {
                // scale + additive mask; key of reg r: kt*32 + (r&3) + 8*(r>>2) + 4*half
                float mx = -INFINITY;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(sM + kt * 32 + 8 * g + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s[g * 4 + e] = fmaf(s[g * 4 + e], sc, mk[e]);
                        mx = fmaxf(mx, s[g * 4 + e]);
                    }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                m_new = fmaxf(m_run, mx);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
                    psum += s[r];
                }
            }
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            if (__any(alpha != 1.0f)) {          // the running max moved for some query of this wave (rare after the first tiles)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
            // P^T fragments (B operand): k-slot j uses this lane's regs 8j..8j+7
            u32x4 pf[2];
            psum = 0.f;                          // re-summed from the rounded probabilities (see rounded_pair_sum)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pf[j][e] = pack16x2<F16>(s[8 * j + 2 * e], s[8 * j + 2 * e + 1]);   // v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32
                    psum += rounded_pair_sum<F16>(pf[j][e]);
                }
            }
            l_run = l_run * alpha + psum;
            // O^T += V^T . P^T ; A operand lane (d = r32, half): keys {16j+4half+0..3, 16j+8+4half+0..3}
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const char* vrow = sV + (dt * 32 + r32) * VROW + (kt * 32 + 4 * half) * 2;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow + (16 * j) * 2);
                    const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + (16 * j + 8) * 2);
                    const u32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
                    o[dt] = mfma16<F16>(vv, pf[j], o[dt]);
                }
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        const int qo = qt * 32 + r32;
        if (qo < p.Tq) {
            char* optr = p.out + (((int64_t)b * p.Tq + qo) * p.ldo + (int64_t)h * dh) * 2;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = dt * 32 + 8 * g + 4 * half;
                    if (d0 < dh) {
                        if constexpr (F16) {
                            if (p.x3 > 0) {             // split-precision output row (SPRC_F16X3): [hi fp16 | lo e4m3 | hi e4m3], logical width p.x3
                                store_split4(optr - (int64_t)h * dh * 2, p.x3, h * dh + d0, o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv,
                                             o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
                                continue;
                            }
                        }
                        uint2 pk;
                        pk.x = pack16x2<F16>(o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv);
                        pk.y = pack16x2<F16>(o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
                        *reinterpret_cast<uint2*>(optr + d0 * 2) = pk;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming bf16 kernel for long query axes (the ViT blocks: Tq = Tk = 257).  The resident-K/V kernel above needs
// 110 KB of LDS for a ViT-g head, i.e. ONE workgroup per CU: its 60 us of staging per layer never overlapped its 144 us
// of compute.  Here a workgroup is 3 waves = 96 queries of one (image, head) and walks the keys in 32-key tiles that are
// double-buffered in LDS (2 x 13.5 KB): global loads of tile t+1 are issued before the MFMAs of tile t and written to
// the other buffer afterwards (register staging, one barrier per tile), so five workgroups fit on a CU (LDS 27 KB,
// <= 128 VGPRs) and one workgroup's loads, softmax VALU and stores run under its neighbours' MFMAs.
// The three workgroups of an (image, head) re-stream the same K/V: the block index is remapped so that they run
// back to back on ONE XCD (block b runs on XCD b % 8) and the re-reads hit that XCD's L2.
// Same math as above: S^T = K.Q^T (a lane owns one query column), online softmax in the exp2 domain, O^T = V^T.P^T with P
// kept in registers; V is transposed on its way into LDS (two keys per 32-bit write).
// VALU diet (PMC on the first version: 20 VALU instructions per MFMA, the waves 50 % in s_waitcnt): (1) the padded-key mask
// only exists in the code of the LAST tile; (2) with head_dim 88 in a 96-row V^T tile, row 88 holds ones, so the softmax
// denominator falls out of the P.V MFMAs (no 16 adds per tile, and it is rescaled with O); (3) the running max only moves
// when a tile's max exceeds it by more than 2^RESCALE_LOG2 (guide T13): P then stays <= 256 -- exact in fp32 accumulation,
// same relative precision in bf16 -- and the 48-multiply rescale of O runs in the first tile and rarely after;
// (4) global addresses of a thread's pieces advance by a constant per tile.
constexpr float RESCALE_LOG2 = 8.0f;

// Bounds-checked 16-B loads through a raw buffer descriptor (non-template helpers: this hipcc drops the host stub of a kernel
// TEMPLATE whose dependent code calls the buffer builtins directly).  A lane whose offset lies outside [0, bytes - 16] reads
// zeros: padded keys, the zero columns of the 96-wide tile and idle lanes need NO branch -- every load of a tile is
// straight-line code, so the compiler can count them (`s_waitcnt vmcnt(N)`) instead of draining the queue at each branch
// (which had serialised the double-buffered prefetch: one exposed ~2-us round trip per key tile).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t attn_rsrc(const char* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 attn_load16(__amdgpu_buffer_rsrc_t rs, uint32_t voffset) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffset, 0, 0));
}
constexpr uint32_t ATTN_OOB = 0x80000000u;   // beyond every descriptor's range (an image's K/V rows span < 2 GiB)

template <int DHP, bool ONES, int NW, bool F16, bool ABLATE = false>
__global__ __launch_bounds__(64 * NW, 3) void attn_stream_kernel(AttnParams p, int nqb, int xcd_map, int debug_arg) {
    // the production instantiation folds every ablation branch away: a run-time branch around the loads or the tile math makes
    // the compiler merge its wait counters at the join and drain the whole load queue there
    const int debug = ABLATE ? debug_arg : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 64 * NW, KS = DHP / 16, DT = DHP / 32, CPR = DHP / 8;
    constexpr int KROW = DHP * 2 + 16;        // bytes per K row in LDS (conflict-free ds_read_b128 across 32 rows)
    constexpr int VROW = 32 * 2 + 8;          // bytes per V^T row of one 32-key tile
    constexpr int KBYTES = 32 * KROW, BUF = KBYTES + DHP * VROW;
    constexpr int KPT = (32 * CPR + NT - 1) / NT;              // 16-B K pieces per thread per tile
    static_assert(16 * CPR <= NT, "one (key pair, chunk) item of V per thread");
    static_assert(NW * 32 * (DHP * 2 + 16) <= 160 * 1024, "output strips fit in LDS");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, half = lane >> 5;
    // Block b runs on XCD b % 8.  A token row of the packed qkv tensor holds the K (V) slices of ALL heads back to back
    // (176 B each for ViT-g: not a multiple of the 128-B line), so neighbouring heads share cache lines: all heads and
    // query blocks of one image are placed on ONE XCD, consecutively, and every fetched line is used by that XCD's L2
    // (heads spread over the XCDs fetched 457 MB per launch against 300 MB of distinct bytes).
    int b, h, qb;
    if (xcd_map == 2) { const int x = blockIdx.x & 7, i = blockIdx.x >> 3; qb = i % nqb; h = (i / nqb) % p.H; b = (i / (nqb * p.H)) * 8 + x; }
    else if (xcd_map == 1) { const int x = blockIdx.x & 7, i = blockIdx.x >> 3; qb = i % nqb; const int bh = (i / nqb) * 8 + x; b = bh / p.H; h = bh % p.H; }
    else { const int bh = blockIdx.x / nqb; qb = blockIdx.x % nqb; b = bh / p.H; h = bh % p.H; }
    const int dh = p.dh;
    const int nkt = (p.Tk + 31) >> 5;
    const bool ragged = (p.Tk & 31) != 0;

    // ---- this thread's share of a tile: KPT 16-B pieces of K, one (key pair, 8-wide chunk) item of V ----
    // descriptors cover this (image, head): rows [0, Tk) of head_dim*2 valid bytes each (the last row bounds the range)
    const __amdgpu_buffer_rsrc_t rs_k = attn_rsrc(p.k + ((int64_t)b * p.Tk * p.ldk + (int64_t)h * dh) * 2,
                                                  (uint32_t)((int64_t)(p.Tk - 1) * p.ldk * 2 + dh * 2));
    const __amdgpu_buffer_rsrc_t rs_v = attn_rsrc(p.v + ((int64_t)b * p.Tk * p.ldv + (int64_t)h * dh) * 2,
                                                  (uint32_t)((int64_t)(p.Tk - 1) * p.ldv * 2 + dh * 2));
    uint32_t k_off[KPT];
    int k_row[KPT], k_lds[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int idx = tid + u * NT, c = idx % CPR;
        k_row[u] = idx / CPR;
        k_lds[u] = k_row[u] * KROW + c * 16;
        // columns >= head_dim stay zero (pad of the 96-wide tile); rows >= 32 belong to no piece
        k_off[u] = (k_row[u] < 32 && c * 8 < dh) ? (uint32_t)(k_row[u] * p.ldk * 2 + c * 16) : ATTN_OOB;
    }
    // consecutive lanes take consecutive 16-B chunks of ONE row (like K): with lanes running over the key pairs instead, a
    // wave instruction touched 16 different rows (16-32 cache lines for 1 KB)
    const int v_kp = tid / CPR, v_c = tid % CPR;
    const bool v_item = v_kp < 16;
    const bool v_ones = ONES && v_item && v_c * 8 == dh;         // this thread owns V^T rows dh .. dh+7: row dh = ones
    // V^T rows are 18 dwords apart, so the eight rows of chunk c, c+4 and c+8 start on the same banks and the 32-bit
    // transposing writes of a wave were 3-way conflicted (SQ_LDS_BANK_CONFLICT 45 % of the LDS cycles).  Key-pair column kp of
    // rows 32..63 / 64..95 is stored at kp ^ 8 / kp ^ 4: at most 2-way (free for ds_write_b32); the fragment reads apply the
    // same constant per 32-row tile (it permutes their four 8-byte pieces, at no cost).
    const int v_col = (v_kp ^ (((v_c >> 2) & 1) * 8) ^ (((v_c >> 3) & 1) * 4)) * 4;
    uint32_t v_off = (v_item && v_c * 8 < dh) ? (uint32_t)(v_kp * 2 * p.ldv * 2 + v_c * 16) : ATTN_OOB;
    const uint32_t kstep = 32 * p.ldk * 2, vstep = 32 * p.ldv * 2, vnext = p.ldv * 2;
    // Two staging register sets: the loads of tile t+2 are issued while tile t is computed and tile t+1 waits in the other
    // set (ablation: with one set -- loads issued one tile ahead -- the kernel ran 175 us, 108 us with the loop's loads
    // removed, 148 us with the MATH removed: a chain of nine exposed ~2-us memory round trips per workgroup).
    struct Stage { u32x4 kreg[KPT], va, vb; };
    Stage stg[2];
    auto fetch = [&](Stage& g) {                                 // the next tile in key order: branch-free, rows >= Tk read zeros
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
            g.kreg[u] = attn_load16(rs_k, k_off[u]);
            k_off[u] += kstep;
        }
        g.va = attn_load16(rs_v, v_off);
        g.vb = attn_load16(rs_v, v_off + vnext);
        v_off += vstep;
    };
    // issued unconditionally, also past the last tile (those rows read zeros at no memory cost): a branch around the loads
    // makes the compiler drain the whole queue (`s_waitcnt vmcnt(0)`) where the paths join
    auto fetch_any = [&](Stage& g, int) { fetch(g); };
    auto commit = [&](Stage& g, int buf) {
        u32x4 va = g.va, vb = g.vb;
        char* sK = smem + buf * BUF;
        char* sV = sK + KBYTES;
#pragma unroll
        for (int u = 0; u < KPT; ++u)
            if (k_row[u] < 32) *reinterpret_cast<u32x4*>(sK + k_lds[u]) = g.kreg[u];
        if (v_item) {
            if (v_ones) va[0] = vb[0] = F16 ? 0x3c00u : 0x3f80u;  // 1.0 (fp16 / bf16) in row dh for both keys of the pair
#pragma unroll
            for (int e = 0; e < 8; e += 2) {                     // rows (e, e+1) of the chunk: low / high halves of dword e/2
                const uint32_t a = va[e >> 1], c = vb[e >> 1];
                *reinterpret_cast<uint32_t*>(sV + (v_c * 8 + e) * VROW + v_col) = (a & 0xffffu) | (c << 16);
                *reinterpret_cast<uint32_t*>(sV + (v_c * 8 + e + 1) * VROW + v_col) = (a >> 16) | (c & 0xffff0000u);
            }
        }
    };
    using std::false_type;
    using std::true_type;
    fetch_any(stg[0], 0);
    fetch_any(stg[1], 1);

    // Q^T fragments (B operand): lane (q = r32, half) holds Q[q][ks*16 + half*8 .. +8]
    const int qt = qb * NW + wave;
    const int qrow = min(qt * 32 + r32, p.Tq - 1);
    const char* qptr = p.q + (((int64_t)b * p.Tq + qrow) * p.ldq + (int64_t)h * dh) * 2;
    u32x4 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = ks * 16 + half * 8;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (d0 < dh && !(debug & 32)) val = *reinterpret_cast<const u32x4*>(qptr + d0 * 2);
        qf[ks] = val;
    }
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.scale * LOG2E;

    // one 32-key tile out of buffer kt & 1
    auto tile = [&](int kt, auto tail_) {
        constexpr bool TAIL = decltype(tail_)::value;
        const char* sK = smem + (kt & 1) * BUF;
        const char* sV = sK + KBYTES;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const char* krow = sK + r32 * KROW + half * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // (issuing all KS fragment reads ahead of the first MFMA through inline asm changed nothing: 124-127 us either way --
            // with three waves per SIMD the other waves cover the read latency)
            const u32x4 kf = *reinterpret_cast<const u32x4*>(krow + ks * 32);
            s = mfma16<F16>(kf, qf[ks], s);
        }
        if constexpr (TAIL) {                   // padded keys: key of reg r = kt*32 + (r&3) + 8*(r>>2) + 4*half
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= p.Tk) s[r] = -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc;            // every tile holds >= 1 real key: finite
        if (__any(mx - m_run > RESCALE_LOG2)) {                  // always in the first tile (m_run = -inf), rarely afterwards
            const float m_new = (mx - m_run > RESCALE_LOG2) ? mx : m_run;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // 1 for the queries that keep their max
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -m_run));
        u32x4 pf[2];
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf[j][e] = pack16x2<F16>(s[8 * j + 2 * e], s[8 * j + 2 * e + 1]);
                if constexpr (!ONES) psum += rounded_pair_sum<F16>(pf[j][e]);       // the denominator sums what the MFMA multiplies
            }
        }
        if constexpr (!ONES) l_run += psum;
        u32x4 vf[DT][2];                        // V^T fragments: issued together, ahead of the softmax arithmetic's tail
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const char* vrow = sV + (dt * 32 + r32) * VROW + (4 * half) * 2;
            const int x = ((dt & 1) * 8) ^ ((dt >> 1) * 4);          // column swizzle of this 32-row tile, in key pairs (4 B)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow + ((8 * j) ^ x) * 4);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + ((8 * j + 4) ^ x) * 4);
                vf[dt][j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                o[dt] = mfma16<F16>(vf[dt][j], pf[j], o[dt]);
    };

    commit(stg[0], 0);
    __syncthreads();
    const int n_main = ragged ? nkt - 1 : nkt;                   // tiles without padded keys
    // iteration kt: tile kt is in LDS buffer kt & 1, tile kt+1 in flight / landed in register set (kt+1) & 1; set kt & 1 is
    // free (committed at the end of iteration kt-1) and takes the loads of tile kt+2.  Unrolled by two: static set indices.
    auto step = [&](int kt, auto par_) {
        constexpr int PAR = decltype(par_)::value;
        if (!(debug & 1)) fetch_any(stg[PAR], kt + 2);
        if (!(debug & 4)) tile(kt, false_type{});
        if (kt + 1 < nkt && !(debug & 2)) commit(stg[PAR ^ 1], (kt + 1) & 1);   // buffer last read in tile kt-1: every wave is past its barrier
        if (!(debug & 8)) __syncthreads();
    };
    for (int kt = 0; kt < n_main; kt += 2) {
        step(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < n_main) step(kt + 1, std::integral_constant<int, 1>{});
    }
    if (ragged) tile(nkt - 1, true_type{});
    float l_tot;
    if constexpr (ONES) {
        // row dh = 88 of O^T (tile dt = 2, d_local = 24 -> reg 12 of the half-0 lanes) is sum_k P[q][k] . 1
        static_assert(DHP == 96, "ones row: head_dim 88 in a 96-row tile");
        l_tot = __shfl(o[2][12], r32, 64);
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.0f / l_tot;
    // Output through LDS: in registers a lane owns one query ROW, so storing straight out is 12 x 8 B per lane with every
    // wave instruction touching 32 different rows -- store-issue bound, 22 us of the launch (ablation).  Each wave
    // transposes its 32 x head_dim tile through a private strip of the (now idle) K/V buffers and writes whole rows as
    // 16-B pieces: 5.5 instructions per lane, 176 contiguous bytes per row.
    constexpr int ORS = DHP * 2 + 16;         // strip row stride: 16-B aligned, 2-way bank conflicts at most
    // (the launch sizes LDS for max(K/V buffers, NW output strips))
    __syncthreads();                          // every wave is done reading K/V tiles
    char* so = smem + wave * (32 * ORS);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = dt * 32 + 8 * g + 4 * half;
            if (d0 < dh) {
                uint2 pk;
                pk.x = pack16x2<F16>(o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv);
                pk.y = pack16x2<F16>(o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
                *reinterpret_cast<uint2*>(so + r32 * ORS + d0 * 2) = pk;
            }
        }
    __builtin_amdgcn_wave_barrier();          // the strip is private to this wave: LDS operations of one wave complete in order
    const int cpr = dh >> 3;
    char* obase = p.out + (((int64_t)b * p.Tq + qt * 32) * p.ldo + (int64_t)h * dh) * 2;
    if (!(debug & 16)) {
        for (int idx = lane; idx < 32 * cpr; idx += 64) {
            const int row = idx / cpr, c = idx - row * cpr;
            if (qt * 32 + row < p.Tq)
#if SPRC_ATTN_NT & 1
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(so + row * ORS + c * 16), reinterpret_cast<u32x4*>(obase + (int64_t)row * p.ldo * 2 + c * 16));
#else
                *reinterpret_cast<u32x4*>(obase + (int64_t)row * p.ldo * 2 + c * 16) = *reinterpret_cast<const u32x4*>(so + row * ORS + c * 16);
#endif
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming kernel, second form (round 3): the K / V tiles go global -> LDS by DMA (`buffer_load_dwordx4 ... lds`) and V is
// read back TRANSPOSED by the LDS unit (`ds_read_b64_tr_b16`).  Ablations of the first form (tools/attn_abl.sh, 135 us on the box
// of that run): without the tile math 113 us, without loads + commit 100 us -- the staging path (two register sets of 16-B loads,
// the commit with its eight transposing 4-B LDS writes and 16 VALU per thread and tile) weighed as much as the arithmetic.  Here
//   * a tile is 32 key rows of K (13 slots of 16 B: 208-B pitch, conflict-free for the fragments' ds_read_b128) and 32 rows of V
//     (12 slots: 192-B pitch = 48 dwords, which spreads the four rows x two 16-column groups that a 32-lane half of a transposing
//     read touches over all 64 banks).  A DMA instruction writes 64 consecutive slots, lane by lane, from any 64 global addresses:
//     slot (key, chunk) <- chunk of that key's row; padding slots of K take the zeros of an out-of-range offset; slots past the K
//     region and the V slots at d >= head_dim are masked off (EXEC) -- V's are filled once, and hold the ONES column (V[k][dh] = 1:
//     the softmax denominator falls out of the P.V MFMAs as before);
//   * no staging registers, no commit: a ring of NBUF tiles, per key tile ONE counted `s_waitcnt vmcnt` + ONE barrier, then the
//     DMA of tile kt + NBUF - 1 is issued into the buffer every wave has just left;
//   * V^T fragments (A operand of O^T += V^T P^T): lane i of a 16-lane group hands the address of 8-byte piece i of a
//     [4 keys][16 d] block (row i / 4, columns 4 (i % 4) ..) and receives column i (tools/tr_probe.hip checks this on the hardware).
// The arithmetic of a tile (S^T = K Q^T, lazy rescale, exp2, O^T += V^T P^T) is the first form's.
template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void attn_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, uint32_t voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, 0, 0, (SPRC_ATTN_NT & 2) ? 2 : 0);
}
__device__ __forceinline__ u32x2 attn_tr_read(const char* lds_src) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)lds_src));
}

// the value lane (l ^ 32) holds: v_permlane32_swap (gfx950) exchanges the two 32-lane halves in the VALU -- __shfl_xor(x, 32) is a
// ds_bpermute, an LDS round trip in the middle of every tile's dependency chain
__device__ __forceinline__ float attn_other_half(float x) {
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

template <int DHP, bool ONES, int NW, bool F16, int NBUF>
__global__ __launch_bounds__(64 * NW, NBUF == 2 ? 4 : 3) void attn_dma_kernel(AttnParams p, int nqb, int xcd_map) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = DHP / 16, DT = DHP / 32, CPR = DHP / 8;
    constexpr int KSL = CPR + 1, VSL = 12;                     // 16-B slots per K / V row
    constexpr int KROW = KSL * 16, VROW = VSL * 16;
    constexpr int KQ = (32 * KSL + 63) / 64, VQ = 32 * VSL / 64;          // DMA wave-instructions per tile
    constexpr int KBYTES = KQ * 1024, VBYTES = 32 * VROW, BUF = KBYTES + VBYTES;   // K region: whole instructions (the last one's tail is padding)
    constexpr int NQ = KQ + VQ, NI = (NQ + NW - 1) / NW;                 // ... and per wave (instruction q = n NW + wave: K pieces first, then V)
    constexpr int D = NBUF - 1;                                          // tiles in flight ahead of the one being multiplied
    static_assert(CPR <= VSL && (32 * VSL) % 64 == 0 && NBUF >= 2 && NBUF <= 3, "tile geometry");
    static_assert(NW * 32 * (DHP * 2 + 16) <= NBUF * BUF, "output strips fit in the tile ring");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, half = lane >> 5;
    int b, h, qb;
    if (xcd_map == 2) { const int x = blockIdx.x & 7, i = blockIdx.x >> 3; qb = i % nqb; h = (i / nqb) % p.H; b = (i / (nqb * p.H)) * 8 + x; }
    else if (xcd_map == 1) { const int x = blockIdx.x & 7, i = blockIdx.x >> 3; qb = i % nqb; const int bh = (i / nqb) * 8 + x; b = bh / p.H; h = bh % p.H; }
    else { const int bh = blockIdx.x / nqb; qb = blockIdx.x % nqb; b = bh / p.H; h = bh % p.H; }
    const int dh = p.dh;
    const int nkt = (p.Tk + 31) >> 5;
    const bool ragged = (p.Tk & 31) != 0;

    // ---- this lane's share of a tile's DMA: instruction q = n * NW + wave writes slots [64 q, 64 q + 64) ----
    const int bk = p.idx1 ? p.idx1[b] : b;    // batch row of this sample's keys / values (kv_index: the optional reference-K|V reuse, one segment)
    const __amdgpu_buffer_rsrc_t rs_k = attn_rsrc(p.k + ((int64_t)bk * p.Tk * p.ldk + (int64_t)h * dh) * 2,
                                                  (uint32_t)((int64_t)(p.Tk - 1) * p.ldk * 2 + dh * 2));
    const __amdgpu_buffer_rsrc_t rs_v = attn_rsrc(p.v + ((int64_t)bk * p.Tk * p.ldv + (int64_t)h * dh) * 2,
                                                  (uint32_t)((int64_t)(p.Tk - 1) * p.ldv * 2 + dh * 2));
    uint32_t d_off[NI];
    bool d_ok[NI];
    const bool d_last = (NI - 1) * NW + wave < NQ;                       // does this wave issue its NI-th instruction? (wave-uniform)
    const uint32_t kstep = 32 * p.ldk * 2, vstep = 32 * p.ldv * 2;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int q = n * NW + wave;
        if (q < KQ) {
            const int slot = q * 64 + lane, key = slot / KSL, c = slot - key * KSL;
            d_ok[n] = true;
            // padding (chunks past head_dim, slots past the 32 rows): zeros through the descriptor's range check
            d_off[n] = (slot < 32 * KSL && c * 8 < dh) ? (uint32_t)(key * p.ldk * 2 + c * 16) : ATTN_OOB;
        } else {
            const int slot = (q - KQ) * 64 + lane, key = slot / VSL, c = slot - key * VSL;
            d_ok[n] = c * 8 < dh;
            d_off[n] = (uint32_t)(key * p.ldv * 2 + c * 16);
            // the V slots no DMA writes, in every buffer of the ring: zeros, and the ONES column at d = head_dim
            if (q < NQ && c * 8 >= dh) {
                const u32x4 fill = {(ONES && c * 8 == dh) ? (F16 ? 0x3c00u : 0x3f80u) : 0u, 0u, 0u, 0u};
#pragma unroll
                for (int u = 0; u < NBUF; ++u) *reinterpret_cast<u32x4*>(smem + u * BUF + KBYTES + slot * 16) = fill;
            }
        }
    }
    // the next tile in key order into ring buffer `buf`: issued unconditionally, also past the last tile (rows >= Tk are out of the
    // descriptor's range: zeros, no memory traffic), so that the wait below is one constant per wave
    auto fetch = [&](int buf) {
        char* dst = smem + buf * BUF;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int q = n * NW + wave;
            if (n + 1 < NI || NQ % NW == 0 || d_last) {
                if (q < KQ) {
                    attn_dma16(rs_k, dst + q * 1024, d_off[n]);
                    d_off[n] += kstep;
                } else {
                    if (d_ok[n]) attn_dma16(rs_v, dst + q * 1024, d_off[n]);     // EXEC-masked: the fill of the other slots stays
                    d_off[n] += vstep;
                }
            }
        }
    };
    auto wait_tile = [&]() {                   // all but this wave's newest (D - 1) tiles' DMAs have landed
        if constexpr (D == 1) attn_wait_vmcnt<0>();
        else if (NQ % NW == 0 || d_last) attn_wait_vmcnt<(D - 1) * NI>();
        else attn_wait_vmcnt<(D - 1) * (NI - 1)>();
    };
    const bool small = ragged && p.Tk - (nkt - 1) * 32 <= 4;     // the last tile holds at most 4 real keys
    const int n_main = small ? nkt - 1 : nkt;
    const float sc = p.scale * LOG2E;
    // V^T fragment addressing: group g = lane >> 4 = 2 half + sub; the lane hands piece i = lane & 15 of the block
    // [keys k0 + 4 half .. +4][d = 32 dt + 16 sub .. +16]: row i >> 2, columns 4 (i & 3)
    const int vi = lane & 15, vsub = (lane >> 4) & 1;
    const int v_lane = (4 * half + (vi >> 2)) * VROW + (16 * vsub + 4 * (vi & 3)) * 2;
    const int k_lane = r32 * KROW + half * 16;
    // Q^T fragments first (B operand): lane (q = r32, half) holds Q[q][ks*16 + half*8 .. +8]; the DMAs queue up behind them
    const int qt = qb * NW + wave;
    const int qrow = min(qt * 32 + r32, p.Tq - 1);
    const char* qptr = p.q + (((int64_t)b * p.Tq + qrow) * p.ldq + (int64_t)h * dh) * 2;
    u32x4 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = ks * 16 + half * 8;
        u32x4 val = {0u, 0u, 0u, 0u};
#if SPRC_ATTN_NT & 4
        if (d0 < dh) val = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(qptr + d0 * 2));
#else
        if (d0 < dh) val = *reinterpret_cast<const u32x4*>(qptr + d0 * 2);
#endif
        qf[ks] = val;
    }

#pragma unroll
    for (int u = 0; u < D; ++u) fetch(u);

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // tail: the tile holds padded keys (a wave-uniform run-time flag: a second instantiation of this body for the last tile, next
    // to tail_small, took the allocator from 119 registers to 53 spilled ones)
    auto tile = [&](int kt, const char* sK, bool tail) {
        const char* sV = sK + KBYTES;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const char* krow = sK + k_lane;
        // (two accumulation chains over even / odd k-steps, 16 more additions: 125.9 vs 123.8 us -- the chain is not what the tile waits for)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 kf = *reinterpret_cast<const u32x4*>(krow + ks * 32);
            s = mfma16<F16>(kf, qf[ks], s);
        }
        if (tail) {                             // padded keys: key of reg r = kt*32 + (r&3) + 8*(r>>2) + 4*half
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= p.Tk) s[r] = -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, attn_other_half(mx)) * sc;               // every tile holds >= 1 real key: finite
        if (__any(mx - m_run > RESCALE_LOG2)) {                  // always in the first tile (m_run = -inf), rarely afterwards
            const float m_new = (mx - m_run > RESCALE_LOG2) ? mx : m_run;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -m_run));
        u32x4 pf[2];
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf[j][e] = pack16x2<F16>(s[8 * j + 2 * e], s[8 * j + 2 * e + 1]);
                if constexpr (!ONES) psum += rounded_pair_sum<F16>(pf[j][e]);
            }
        }
        if constexpr (!ONES) l_run += psum;
        // V^T fragments (the compiler moves the reads up into the exponentials)
        u32x4 vf[DT][2];
        const char* vb = sV + v_lane;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x2 lo = attn_tr_read(vb + (16 * j) * VROW + dt * 64);
                const u32x2 hi = attn_tr_read(vb + (16 * j + 8) * VROW + dt * 64);
                vf[dt][j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                o[dt] = mfma16<F16>(vf[dt][j], pf[j], o[dt]);
    };

    // A last tile with at most 4 real keys (the ViT's 257 tokens: ONE).  They are keys 0..3 of the tile = registers 0..3 of the half-0
    // lanes, so the softmax arithmetic runs on 4 values instead of 16 and P.V on the first 16-key step only, whose upper eight keys
    // carry zero probabilities (their V fragment is not even read): ~45 % of a full tile, and the ninth tile was 11 % of the key loop.
    auto tail_small = [&](int kt, const char* sK) {
        const char* sV = sK + KBYTES;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const char* krow = sK + k_lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 kf = *reinterpret_cast<const u32x4*>(krow + ks * 32);
            s = mfma16<F16>(kf, qf[ks], s);
        }
        const int tk = p.Tk - kt * 32;
        float sv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = (half == 0 && r < tk) ? s[r] : -INFINITY;
        float mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        mx = fmaxf(mx, attn_other_half(mx)) * sc;               // key 0 is real: finite
        if (__any(mx - m_run > RESCALE_LOG2)) {
            const float m_new = (mx - m_run > RESCALE_LOG2) ? mx : m_run;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = __builtin_amdgcn_exp2f(fmaf(sv[r], sc, -m_run));       // exp2(-inf) = 0 for the padded keys
        const u32x4 pf = {pack16x2<F16>(sv[0], sv[1]), pack16x2<F16>(sv[2], sv[3]), 0u, 0u};
        if constexpr (!ONES) l_run += rounded_pair_sum<F16>(pf[0]) + rounded_pair_sum<F16>(pf[1]);
        const char* vb = sV + v_lane;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const u32x2 lo = attn_tr_read(vb + dt * 64);
            o[dt] = mfma16<F16>(u32x4{lo[0], lo[1], 0u, 0u}, pf, o[dt]);
        }
    };

    // iteration kt: tiles kt .. kt + D - 1 are in flight or landed; wait for tile kt (this wave's share: everything but the DMAs of
    // its newest D - 1 tiles), barrier (every wave's share is visible, and every wave has left tile kt - 1's buffer), refill that
    // buffer with tile kt + D, multiply tile kt
    int buf = 0, nxt = D % NBUF;
    for (int kt = 0; kt < n_main; ++kt) {
        wait_tile();
        __syncthreads();
        fetch(nxt);
        tile(kt, smem + buf * BUF, ragged && kt == nkt - 1);
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }
    if (small) {                              // (nothing left to fetch)
        attn_wait_vmcnt<0>();
        __syncthreads();
        tail_small(nkt - 1, smem + buf * BUF);
    }
    float l_tot;
    if constexpr (ONES) {
        static_assert(DHP == 96, "ones row: head_dim 88 in a 96-row tile");
        l_tot = __shfl(o[2][12], r32, 64);       // row dh = 88 of O^T: tile dt = 2, d_local = 24 -> reg 12 of the half-0 lanes
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.0f / l_tot;
    constexpr int ORS = DHP * 2 + 16;
    attn_wait_vmcnt<0>();                     // the DMAs issued past the last tile still write into the ring
    if constexpr (F16) {
        if (p.x3 > 0) {                       // split-precision output rows (SPRC_F16X3, logical width p.x3 = H * dh): [hi fp16 | lo e4m3 | hi e4m3], straight
            const int qo = qt * 32 + r32;     // from the registers (a lane owns one query row) -- the Q-Former's 32-query cross-attention launches
            if (qo < p.Tq) {
                char* orow = p.out + ((int64_t)b * p.Tq + qo) * p.ldo * 2;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int d0 = dt * 32 + 8 * g + 4 * half;
                        if (d0 < dh)
                            store_split4(orow, p.x3, h * dh + d0, o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv, o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
                    }
            }
            return;
        }
    }
    __syncthreads();                          // every wave is done reading K/V tiles
    char* so = smem + wave * (32 * ORS);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = dt * 32 + 8 * g + 4 * half;
            if (d0 < dh) {
                uint2 pk;
                pk.x = pack16x2<F16>(o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv);
                pk.y = pack16x2<F16>(o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
                *reinterpret_cast<uint2*>(so + r32 * ORS + d0 * 2) = pk;
            }
        }
    __builtin_amdgcn_wave_barrier();
    char* obase = p.out + (((int64_t)b * p.Tq + qt * 32) * p.ldo + (int64_t)h * dh) * 2;
    auto store_rows = [&](int cpr) {          // 16-B chunks per output row
        for (int idx = lane; idx < 32 * cpr; idx += 64) {
            const int row = idx / cpr, c = idx - row * cpr;
            if (qt * 32 + row < p.Tq)
#if SPRC_ATTN_NT & 1
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(so + row * ORS + c * 16), reinterpret_cast<u32x4*>(obase + (int64_t)row * p.ldo * 2 + c * 16));
#else
                *reinterpret_cast<u32x4*>(obase + (int64_t)row * p.ldo * 2 + c * 16) = *reinterpret_cast<const u32x4*>(so + row * ORS + c * 16);
#endif
        }
    };
    if (ONES) store_rows(11);                 // head_dim 88 (compile-time divisor: the run-time division cost ~25 VALU per piece)
    else if (dh == DHP) store_rows(DHP / 8);
    else store_rows(dh >> 3);
}

// ------------------------------------------------------------------------------------------------
// exact fp32: one wave per query row; lanes = keys for QK^T, lanes = head dims for PV.
constexpr int F32_MAXK = 9;     // keys per lane -> Tk <= 576 (the rerank's 514 = 2 x 257 encoder tokens)
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi_raw = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
    const bool valid = qi_raw < p.Tq;
    const int qi = valid ? qi_raw : p.Tq - 1;
    const int dh = p.dh;
    float* qs = reinterpret_cast<float*>(smem) + wave * (dh + p.Tk);
    float* ps = qs + dh;
    const float* q = reinterpret_cast<const float*>(p.q) + ((int64_t)b * p.Tq + qi) * p.ldq + (int64_t)h * dh;
    // fp32 rows of the (possibly two-segment) key axis
    auto kv_row = [&](const char* seg1, int64_t ld1, const char* seg2, int64_t ld2, int t) -> const float* {
        if (t < p.Tk1) return reinterpret_cast<const float*>(seg1) + ((int64_t)(p.idx1 ? p.idx1[b] : b) * p.Tk1 + t) * ld1 + (int64_t)h * dh;
        return reinterpret_cast<const float*>(seg2) + ((int64_t)(p.idx2 ? p.idx2[b] : b) * (p.Tk - p.Tk1) + (t - p.Tk1)) * ld2 + (int64_t)h * dh;
    };
    for (int d = lane; d < dh; d += 64) qs[d] = q[d];
    __syncthreads();
    float s[F32_MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < F32_MAXK; ++i) {
        const int key = lane + i * 64;
        s[i] = -INFINITY;
        if (key < p.Tk) {
            const float* kr = kv_row(p.k, p.ldk, p.k2, p.ldk2, key);
            float acc = 0.f;
            for (int d = 0; d < dh; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(kr + d);
                acc = fmaf(qs[d], kv.x, acc); acc = fmaf(qs[d + 1], kv.y, acc);
                acc = fmaf(qs[d + 2], kv.z, acc); acc = fmaf(qs[d + 3], kv.w, acc);
            }
            s[i] = acc * p.scale + (p.key_mask ? p.key_mask[(int64_t)b * p.Tk + key] : 0.f);
            mx = fmaxf(mx, s[i]);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < F32_MAXK; ++i) {
        const int key = lane + i * 64;
        if (key < p.Tk) {
            const float e = expf(s[i] - mx);
            sum += e;                                   // the softmax is normalised BEFORE the dropout (Qformer.py:260-264)
            ps[key] = (p.drop_thresh == 0 || drop_keep(p.drop_seed, p.drop_site, (((uint64_t)b * p.H + h) * p.Tq + qi) * p.Tk + key, p.drop_thresh))
                          ? e * p.drop_scale : 0.f;
        }
    }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    float* o = reinterpret_cast<float*>(p.out) + ((int64_t)b * p.Tq + qi) * p.ldo + (int64_t)h * dh;
    for (int d = lane; d < dh; d += 64) {
        float acc = 0.f;
        for (int key = 0; key < p.Tk; ++key) acc = fmaf(ps[key], kv_row(p.v, p.ldv, p.v2, p.ldv2, key)[d], acc);
        if (valid) o[d] = acc * inv;
    }
}

template <int DHP, int NW, bool F16>
static int launch_bf16(const AttnParams& p, hipStream_t st) {
    const int Tkp = (p.Tk + 31) & ~31;
    const size_t lds = (size_t)Tkp * (DHP * 2 + 16) + (size_t)DHP * (Tkp * 2 + 8) + (size_t)Tkp * 4;
    if (lds > 160 * 1024) {
        set_error("sprc_attention: Tk=%d needs %zu bytes of LDS (max 163840)", p.Tk, lds);
        return SPRC_EUNSUPPORTED;
    }
    auto kern = attn_bf16_kernel<DHP, NW, F16>;
    static size_t attr[64] = {0};               // hipFuncSetAttribute applies to the current device: one high-water mark per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev >= 0 && dev < 64 ? dev : 0;
    if (lds > attr[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr[dev] = lds;
    }
    hipLaunchKernelGGL(kern, dim3(p.B * p.H), dim3(64 * NW), lds, st, p);
    SPRC_CHECK_LAUNCH("sprc_attention(bf16)");
    return SPRC_OK;
}

template <int DHP, bool ONES, int NW, bool F16>
static int launch_stream(const AttnParams& p, hipStream_t st) {
    constexpr int BUF = 32 * (DHP * 2 + 16) + DHP * 72;
    constexpr int LDS = 2 * BUF > NW * 32 * (DHP * 2 + 16) ? 2 * BUF : NW * 32 * (DHP * 2 + 16);
    const int nqb = ((p.Tq + 31) / 32 + NW - 1) / NW;
    const int bh = p.B * p.H;
    // SPRC_ATTN_DEBUG (timing ablations, WRONG results): 1 no global loads in the loop, 2 no LDS commit, 4 no tile math, 8 no barrier
    static const int debug = [] { const char* e = getenv("SPRC_ATTN_DEBUG"); return e ? atoi(e) : 0; }();
    static const int xcd = [] { const char* e = getenv("SPRC_ATTN_XCD"); return e ? atoi(e) : 1; }();
    const int xmap = !xcd ? 0 : (xcd == 3 && (bh % 8) == 0) ? 1 : (p.B % 8) == 0 ? 2 : (bh % 8) == 0 ? 1 : 0;
    if constexpr (!F16) {
        if (debug != 0) {
            hipLaunchKernelGGL((attn_stream_kernel<DHP, ONES, NW, false, true>), dim3(bh * nqb), dim3(64 * NW), LDS, st, p, nqb, xmap, debug);
            SPRC_CHECK_LAUNCH("sprc_attention(bf16, streaming, ablation)");
            return SPRC_OK;
        }
    }
    hipLaunchKernelGGL((attn_stream_kernel<DHP, ONES, NW, F16, false>), dim3(bh * nqb), dim3(64 * NW), LDS, st, p, nqb, xmap, 0);
    SPRC_CHECK_LAUNCH("sprc_attention(bf16, streaming)");
    return SPRC_OK;
}

template <int DHP, bool ONES, int NW, bool F16, int NBUF>
static int launch_dma(const AttnParams& p, hipStream_t st) {

    constexpr int BUF = (32 * (DHP / 8 + 1) + 63) / 64 * 1024 + 32 * 12 * 16;       // K region in whole DMA instructions + V region
    constexpr int LDS = NBUF * BUF;
    const int nqb = ((p.Tq + 31) / 32 + NW - 1) / NW;
    const int bh = p.B * p.H;
    static const int xcd = [] { const char* e = getenv("SPRC_ATTN_XCD"); return e ? atoi(e) : 1; }();
    const int xmap = !xcd ? 0 : (xcd == 3 && (bh % 8) == 0) ? 1 : (p.B % 8) == 0 ? 2 : (bh % 8) == 0 ? 1 : 0;
    hipLaunchKernelGGL((attn_dma_kernel<DHP, ONES, NW, F16, NBUF>), dim3(bh * nqb), dim3(64 * NW), LDS, st, p, nqb, xmap);
    SPRC_CHECK_LAUNCH("sprc_attention(16-bit, streaming, DMA)");
    return SPRC_OK;
}

}  // namespace sprc

namespace sprc {
static bool Tk_all_le64(const AttnParams& p) { return p.Tk <= 64; }
// 16-bit engines (F16: fp16 operands, else bf16): kernel choice by shape
template <bool F16>
static int attention16(const sprc_attention_args* a, const AttnParams& p, bool two, hipStream_t st) {
    SPRC_REQUIRE(a->head_dim % 8 == 0 && a->head_dim <= 96, "sprc_attention(16-bit): head_dim=%d unsupported", a->head_dim);
    SPRC_REQUIRE(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldv % 8 == 0 && a->ldo % 4 == 0,
                 "sprc_attention(16-bit): leading dims must be multiples of 8");
    SPRC_REQUIRE(((uintptr_t)a->q % 16) == 0 && ((uintptr_t)a->k % 16) == 0 && ((uintptr_t)a->v % 16) == 0 &&
                     ((uintptr_t)a->out % 8) == 0, "sprc_attention(16-bit): misaligned pointer");
    SPRC_REQUIRE(!two || (a->ldk2 % 8 == 0 && a->ldv2 % 8 == 0 && ((uintptr_t)a->k2 % 16) == 0 && ((uintptr_t)a->v2 % 16) == 0),
                 "sprc_attention(16-bit): second key segment misaligned");
    const bool small = a->Tq <= 128;
    // long query axes without a key mask (the ViT blocks): streaming kernel, four workgroups of three waves per CU (153 VGPRs;
    // 5- and 9-wave workgroups measured 268 / 170 us against 125: fewer co-resident workgroups).
    // (SPRC_ATTN_STREAM=0 keeps the resident-K/V kernel for A/B runs)
    static const int stream = [] { const char* e = getenv("SPRC_ATTN_STREAM"); return e ? atoi(e) : 1; }();
    // SPRC_ATTN_DMA: 0 = the first streaming form (register staging), 2 / 3 = the DMA form with a ring of that many tiles
    static const int dma = [] { const char* e = getenv("SPRC_ATTN_DMA"); return e ? atoi(e) : 2; }();
    // ONE 32-query tile over a long key axis (the Q-Former's cross-attention: 32 query tokens x 257 encoder tokens, 12 heads x 64; Qformer.py:175-281):
    // the resident kernel stages all of K and V^T through registers into LDS with four waves and then multiplies on ONE of them (108.9 us per
    // launch of 233 x 12 heads against a ~37-us memory floor).  The streaming DMA kernel with ONE-wave workgroups keeps a ring of 32-key tiles
    // per workgroup (11.3 KB each), 7 / 4 workgroups per CU (ring of 2 / 3).  SPRC_ATTN_CROSS: 0 = the resident kernel (A/B), 2 (default) / 3 = ring depth.
    // Measured (profiles/r06_cross_attn_ab.txt; launches alone, K|V rows of 9216 elements as in the model): 233 x 12 heads 88.3 -> 39.1 us (ring of 2;
    // 45.5 with a ring of 3: fewer workgroups per CU), 128 x 12 heads 47.5 -> 21.3; the bench step 88.0 -> 87.65 ms, same box, alternating.
    static const int cross = [] { const char* e = getenv("SPRC_ATTN_CROSS"); return e ? atoi(e) : 2; }();
    // (kv_index on ONE segment -- sprc_qformer_fuse_kv -- takes the same kernel: its scores stay bit-identical to the per-query projection's)
    if (cross && a->Tq <= 32 && p.Tk >= 128 && a->head_dim <= 64 && a->key_mask == nullptr && !two &&
        (a->out_x3 ? F16 : (a->ldo % 8 == 0 && ((uintptr_t)a->out % 16) == 0))) {
        return cross == 2 ? launch_dma<64, false, 1, F16, 2>(p, st) : launch_dma<64, false, 1, F16, 3>(p, st);
    }
    if (stream && dma && !small && a->key_mask == nullptr && !two && a->kv_index == nullptr && a->ldo % 8 == 0 && ((uintptr_t)a->out % 16) == 0) {
        static const int nw4 = [] { const char* e = getenv("SPRC_ATTN_NW"); return e ? atoi(e) == 4 : 0; }();
        if (dma == 2 && nw4) {
            if (a->head_dim <= 64) return launch_dma<64, false, 4, F16, 2>(p, st);
            if (a->head_dim == 88) return launch_dma<96, true, 4, F16, 2>(p, st);
        }
        if (dma == 2) {
            if (a->head_dim <= 64) return launch_dma<64, false, 3, F16, 2>(p, st);
            if (a->head_dim == 88) return launch_dma<96, true, 3, F16, 2>(p, st);
            return launch_dma<96, false, 3, F16, 2>(p, st);
        }
        if (a->head_dim <= 64) return launch_dma<64, false, 3, F16, 3>(p, st);
        if (a->head_dim == 88) return launch_dma<96, true, 3, F16, 3>(p, st);
        return launch_dma<96, false, 3, F16, 3>(p, st);
    }
    if (stream && !small && a->key_mask == nullptr && !two && a->kv_index == nullptr && a->ldo % 8 == 0 && ((uintptr_t)a->out % 16) == 0) {
        if (a->head_dim <= 64) return launch_stream<64, false, 3, F16>(p, st);
        if (a->head_dim == 88) return launch_stream<96, true, 3, F16>(p, st);     // denominator from the ones row of the padded V^T tile
        return launch_stream<96, false, 3, F16>(p, st);
    }
    // 257 tokens = 9 query tiles: nine waves (one tile each, the K / V staging shared by nine) beat eight waves of which
    // one carries two tiles: 209 -> 195 us per ViT-g layer
    const int nqt = (a->Tq + 31) / 32;
    if (nqt == 9) {
        if (a->head_dim <= 64) return launch_bf16<64, 9, F16>(p, st);
        return launch_bf16<96, 9, F16>(p, st);
    }
    // small query axes (the Q-Former: 32 or 64 query rows = one or two 32-row tiles): SPRC_ATTN_SMALL_NW waves per workgroup
    static const int small_nw = [] { const char* e = getenv("SPRC_ATTN_SMALL_NW"); return e ? atoi(e) : 4; }();
    // measured (tools/qf_attn_bench.py, 233 x 12 heads): self-attention over 64 keys 40.9 us with four waves per workgroup, 28.5 with
    // two (one per query tile: twice the workgroups per CU); cross-attention over 257 keys 84.5 / 106.8 / 142.5 us with 4 / 2 / 1
    // waves (there the extra waves carry the K / V staging) -> two waves for short key axes, four otherwise
    if (a->head_dim <= 64 && small && (small_nw == 2 || (small_nw == 4 && Tk_all_le64(p)))) return launch_bf16<64, 2, F16>(p, st);
    if (a->head_dim <= 64 && small && small_nw == 1) return launch_bf16<64, 1, F16>(p, st);
    if (a->head_dim <= 64) return small ? launch_bf16<64, 4, F16>(p, st) : launch_bf16<64, 8, F16>(p, st);
    return small ? launch_bf16<96, 4, F16>(p, st) : launch_bf16<96, 8, F16>(p, st);
}
}  // namespace sprc

extern "C" int sprc_attention(const sprc_attention_args* a, sprc_stream s) {
    using namespace sprc;
    SPRC_REQUIRE(a && a->q && a->k && a->v && a->out, "sprc_attention: null pointer");
    SPRC_REQUIRE(a->B > 0 && a->H > 0 && a->Tq > 0 && a->Tk > 0 && a->head_dim > 0, "sprc_attention: bad shape");
    const bool two = a->k2 != nullptr;
    SPRC_REQUIRE(!two || (a->v2 && a->Tk2 > 0 && !a->key_mask), "sprc_attention: a second key segment needs k2, v2, Tk2 > 0 and no key mask");
    SPRC_REQUIRE(two || (!a->kv2_index && !a->v2), "sprc_attention: kv2_index / v2 come with a second key segment (k2)");
    SPRC_REQUIRE(!a->kv_index || two || a->Tq <= 128 || a->dtype == SPRC_F32,
                 "sprc_attention: kv_index on a single key segment is served by the resident kernels (Tq <= 128) and the fp32 one");
    const int Tk_all = a->Tk + (two ? a->Tk2 : 0);
    AttnParams p{a->B, a->H, a->Tq, Tk_all, a->head_dim, (const char*)a->q, a->ldq, (const char*)a->k, a->ldk,
                 (const char*)a->v, a->ldv, (char*)a->out, a->ldo, a->key_mask, a->scale,
                 a->Tk, (const char*)a->k2, a->ldk2, (const char*)a->v2, a->ldv2, a->kv_index, a->kv2_index,
                 a->out_x3 ? a->H * a->head_dim : 0,
                 a->drop_p > 0.f ? drop_thresh(a->drop_p) : 0u, a->drop_site, a->drop_seed, a->drop_p > 0.f ? 1.0f / (1.0f - a->drop_p) : 1.0f};
    SPRC_REQUIRE(a->drop_p >= 0.f && a->drop_p < 1.f && (a->drop_p == 0.f || (a->dtype == SPRC_F32 && !two)),
                 "sprc_attention: drop_p in [0, 1); probability dropout is a mode of the fp32 (training) kernel, one key segment");
    SPRC_REQUIRE(!a->out_x3 || (a->dtype == SPRC_F16 && a->Tq <= 128 && a->ldo >= 2 * (int64_t)a->H * a->head_dim && (a->H * a->head_dim) % 4 == 0),
                 "sprc_attention: out_x3 needs dtype SPRC_F16, Tq <= 128 (resident kernel), ldo >= 2 H head_dim (a split row is 4 bytes per column)");
    hipStream_t st = (hipStream_t)s;
    const double bh = (double)a->B * a->H, esz = (double)dtype_size(a->dtype);
    ProfScope prof(SPRC_K_ATTN, st, 4.0 * bh * a->Tq * (double)Tk_all * a->head_dim,
                   bh * a->head_dim * esz * (2.0 * a->Tq + 2.0 * Tk_all));
    if (a->dtype == SPRC_BF16) return attention16<false>(a, p, two, st);
    if (a->dtype == SPRC_F16) return attention16<true>(a, p, two, st);
    SPRC_REQUIRE(a->dtype == SPRC_F32, "sprc_attention: bad dtype %d", a->dtype);
    SPRC_REQUIRE(Tk_all <= 64 * F32_MAXK, "sprc_attention(f32): Tk=%d > %d", Tk_all, 64 * F32_MAXK);
    SPRC_REQUIRE(a->head_dim % 4 == 0 && a->ldk % 4 == 0 && (!two || a->ldk2 % 4 == 0), "sprc_attention(f32): head_dim/ldk must be multiples of 4");
    const size_t lds = 4 * (size_t)(a->head_dim + Tk_all) * sizeof(float);
    hipLaunchKernelGGL(attn_f32_kernel, dim3((a->Tq + 3) / 4, a->H, a->B), dim3(256), lds, st, p);
    SPRC_CHECK_LAUNCH("sprc_attention(f32)");
    return SPRC_OK;
}
