// gemm_impl.hpp -- C = epilogue(A[M,K] . W[N,K]^T) on gfx950 matrix cores: kernels, launchers and the per-operand-type
// dispatcher.  Included by gemm.hip (C entry points + bf16 operands), gemm_f32.hip and gemm_fp8.hip: one translation unit per
// operand type, so the three sets of template instantiations compile in parallel (one TU took six minutes).
//
// Replaces every nn.Linear / F.linear / conv-as-GEMM on the SPRC retrieval path (see sprc.h).
// Both operands are K-contiguous ("B^T" form), so A and W fragments are read the same way.
//
// Two kernels (workgroup tile BM x BN, K-tile = 128 bytes of K):
//   gemm_anti_kernel  256 x 256, 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 MFMA 32x32 accumulators, 1 WG / CU, bf16:
//                     two wave groups in anti-phase (one in a 16-MFMA cluster while the other reads fragments and
//                     stages), piece-scheduled staging with counted vmcnt -- see the comment above the kernel
//   gemm_kernel       128 x 128, 4 waves (2 x 2), wave tile 64 x 64, 2 WGs / CU (small grids, remainder rows, split-K,
//                     and every fp32 GEMM; also instantiated as a lock-step 256 x 256 for the fp32 engine)
//     bf16: v_mfma_f32_32x32x16_bf16, K-tile = 64 elements;  f32: v_mfma_f32_32x32x2_f32 (exact fp32), 32 elements
//   MIX instantiations of both (fp16 operands, gemm_f16e.hip): split-precision products -- fp16 K-tiles, then e4m3 K-tiles on the
//     MX-scaled MFMA into the same accumulators (sprc.h: SPRC_F16X3, sprc_gemm_args.k8)
// The dispatcher (launch<>) chooses between them -- and a "256x256 on the first M & ~255 rows + 128x128 on the rest"
// split -- with a round-count cost model.
// Data movement: K-tiles go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds through an SRSRC based at the tile's
// first row: 1 KiB per wave-instruction, no VGPR round trip, no per-load address arithmetic), double buffered.  The LDS
// image of a wave-instruction is lane-linear (8 rows x 8 16-B slots), so the XOR swizzle (slot ^= (row>>1)&7: every
// ds_read_b128 lane group hits 16 distinct slots of the 256-B bank row, SQ_LDS_BANK_CONFLICT = 0) is applied to the
// per-lane SOURCE offset and to the fragment reads, never to the destination (guide rule 21).
// The 128x128 bf16 main loop is hand software-pipelined (inline-asm ds_read_b128 + counted lgkmcnt): fragment reads of
// k-step kk+1 and a quarter of the next K-tile's loads are issued before the MFMAs of k-step kk.
// Tile order: XCD-contiguous remap of the block index (block b runs on XCD b%8) + grouped order, so tiles resident on
// one XCD share A/W panels in its L2.
// Epilogue: MFMAs compute the TRANSPOSED tile, so a lane owns one C row and 4 consecutive columns per register
// quad: bias / residual / output are 16-B (fp32) or 8-B (bf16) vectors.
#pragma once
#include <stdlib.h>

#include <type_traits>

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
    float* scratch; int64_t scratch_elems;   // optional caller scratch (split-K partials)
    int order;                          // tile order (tile_origin): 0 = 8-m grouped, n > 0 = groups of n n-tiles sweeping m
    int dual;                           // 1: the grid holds TWO products of identical shape; workgroups >= nwg0 take the second set
    int nwg0;
    const char* W2; const float* bias2; int a_off2, c_off2;      // what differs in the second product: weight, bias, row-map offsets
    int ksplit;                         // > 1: split-K launch of the 128x128 kernel (grid = tiles x ksplit), raw fp32 partials
    int64_t split_stride;               // elements between the partial planes of consecutive K splits
    const float* w_scale; float a_scale;     // fp8 operands: acc * (a_scale * w_scale[n]) before the bias (null: no scaling)
    float out_scale;                    // fp8 output: value * out_scale before the conversion (1 / the consumer's dequantisation scale)
    int debug;                          // SPRC_GEMM_DEBUG on the 256x256 kernel: 64 s_memtime stamp build (tools/gemm_stamp.py)
    int k8;                             // MIX kernels (split-precision products): e4m3 reduction elements that FOLLOW the K fp16 elements in
                                        // every row of A and W (K-tiles of 128 bytes either way); their partial sum enters scaled by 2^-18
    int duo_sleep;                      // duo kernel (gemm_duo.hpp): cycles the second workgroup of a CU sleeps before its first tile (0: no stagger)
    int* duo_ctr;                       // duo kernel: per-CU arrival counters (DUO_CTRS ints, only ever incremented)
    int epi_wide;                       // 1: 8 columns per lane in the epilogues of outputs narrower than fp32 (16-B stores); 0: 4 (A/B switch)
};

// MX block scales of the split-precision products' e4m3 segments: E8M0 118 = 2^-9 on BOTH operands -> 2^-18 on the product, the
// factor the producers put in ([x_lo 2^12 | x] . [W 2^6 | W_lo 2^18], sprc.h: SPRC_F16X3).  One constant for every e4m3 K-tile.
constexpr int MX_UNIT_SCALE = 0x7f7f7f7f, MX_SPLIT_SCALE = 0x76767676;

__device__ __forceinline__ int64_t map_row_s(int shift, int stride, int off, int r) {
    if (shift < 0) return r;
    return (int64_t)(r >> shift) * stride + (r & ((1 << shift) - 1)) + off;
}

// GELU of the bf16 epilogue: x * Phi(x) with Phi(x) ~ 1 / (1 + 2^(x (k0 + k1 t + k2 t^2))), t = min(x^2, 80) -- a logistic
// approximation of the normal CDF with a fitted odd quintic exponent: |gelu error| <= 2.6e-5 for every finite x (checked
// in fp32 on [-40, 40]), an order below the bf16 rounding of the stored value.  9 VALU ops per value.  The epilogue runs
// with the matrix pipe idle and is VALU-issue bound (4 cycles per wave64 op, two waves per SIMD), so it costs what it
// counts: on the 750-us ViT fc1 GEMM the erf form (A&S 7.1.26, ~17 ops) took 150 us, a degree-8 Horner polynomial (13 ops,
// packed or scalar, pinned constants or literals -- all the same) 106-117 us, x*sigmoid(1.702x) (5 ops) 40 us.
__device__ __forceinline__ float gelu_fast(float x) {
    const float t = fminf(x * x, 80.0f);
    float q = fmaf(t, 1.014263136e-03f, -1.067757234e-01f);
    q = fmaf(q, t, -2.301121235e+00f);
    const float e = __builtin_amdgcn_exp2f(x * q);          // +inf for very negative x: rcp(inf) = 0
    return x * __builtin_amdgcn_rcpf(e + 1.0f);
}

struct fp8_t { uint8_t bits; };        // OCP e4m3fn (gfx950), operand / output tag type
struct f16x3_t { uint16_t bits; };     // OUTPUT tag: split-precision rows SPRC_F16X3 ([hi fp16 | lo e4m3 | hi e4m3], see sprc.h)

// saturating fp32 -> 2 x e4m3fn (v_cvt_pk_fp8_f32: RNE; inputs clamped to +-448 first, the format has no infinity)
__device__ __forceinline__ uint32_t pack_fp8x2(float a, float b, uint32_t old, bool hi) {
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f);
    b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    return hi ? (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, true) : (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, false);
}
// Cache policy of the epilogue's global accesses (bit mask; -DSPRC_EPI_NT=0 is the A/B build without any hint).  1: the 16-byte stores of outputs
// narrower than fp32 are NON-TEMPORAL (`global_store_dwordx4 ... nt`): at the end of a round 256 workgroups write 33 MB of qkv / fc1 output at
// once -- more than the L2s hold -- and with the default write-allocate policy that burst evicts the A / W panels every CU re-reads in the next
// round.  Same box, round-robin (profiles/r06_nt_ab.txt): 87.61 -> 86.76 ms per bench step, GEMM class 78.2 -> 77.2 ms.  2: fp32 stores as well
// (86.72: neutral -- the LayerNorm reads x right behind them), 4: non-temporal residual loads (neutral).  NT on the staging LOADS of either operand costs
// 3-4 ms (SPRC_LD_NT), NT attention loads 1.5 ms, NT stores of the attention output or of the LayerNorm's 16-bit copy 0.2-0.3 ms: those are re-read.
#ifndef SPRC_EPI_NT
#define SPRC_EPI_NT 1
#endif
// 4 consecutive outputs of one row: 16 B (fp32) or 8 B (bf16 / fp16)
// n_split / col: logical row width N and this quad's first column (SPRC_F16X3 outputs only: the e4m3 segments sit behind the N fp16 values)
template <typename OutT>
__device__ __forceinline__ void store_out4(OutT* dst, const f32x4& v, int n_split = 0, int col = 0) {
    if constexpr (std::is_same<OutT, float>::value) {
#if SPRC_EPI_NT & 2
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
#else
        *reinterpret_cast<f32x4*>(dst) = v;
#endif
    } else if constexpr (std::is_same<OutT, f16x3_t>::value) {
        store_split4(reinterpret_cast<char*>(dst) - 2 * col, n_split, col, v[0], v[1], v[2], v[3]);
    } else if constexpr (std::is_same<OutT, f16_t>::value) {
        typedef __attribute__((ext_vector_type(4))) _Float16 half4;
        *reinterpret_cast<half4*>(dst) = half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    } else if constexpr (std::is_same<OutT, fp8_t>::value) {
        *reinterpret_cast<uint32_t*>(dst) = pack_fp8x2(v[2], v[3], pack_fp8x2(v[0], v[1], 0u, false), true);
    } else {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        *reinterpret_cast<bf16x4*>(dst) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
}

// 8 consecutive outputs of one row (outputs narrower than fp32): ONE 16-B store for 16-bit types, 8 B for fp8, 16 + 8 + 8 B for split rows
template <typename OutT>
__device__ __forceinline__ void store_out8(OutT* dst, const f32x4& a, const f32x4& b, int n_split = 0, int col = 0) {
    if constexpr (std::is_same<OutT, f16x3_t>::value) {
        store_split8(reinterpret_cast<char*>(dst) - 2 * col, n_split, col, a, b);
    } else if constexpr (std::is_same<OutT, f16_t>::value) {
#if SPRC_EPI_NT & 1
        __builtin_nontemporal_store(u32x4{pack_f16x2(a[0], a[1]), pack_f16x2(a[2], a[3]), pack_f16x2(b[0], b[1]), pack_f16x2(b[2], b[3])}, reinterpret_cast<u32x4*>(dst));
#else
        *reinterpret_cast<u32x4*>(dst) = u32x4{pack_f16x2(a[0], a[1]), pack_f16x2(a[2], a[3]), pack_f16x2(b[0], b[1]), pack_f16x2(b[2], b[3])};
#endif
    } else if constexpr (std::is_same<OutT, fp8_t>::value) {
#if SPRC_EPI_NT & 1
        __builtin_nontemporal_store(u32x2{pack_fp8x2(a[2], a[3], pack_fp8x2(a[0], a[1], 0u, false), true),
                                          pack_fp8x2(b[2], b[3], pack_fp8x2(b[0], b[1], 0u, false), true)}, reinterpret_cast<u32x2*>(dst));
#else
        *reinterpret_cast<u32x2*>(dst) = u32x2{pack_fp8x2(a[2], a[3], pack_fp8x2(a[0], a[1], 0u, false), true),
                                               pack_fp8x2(b[2], b[3], pack_fp8x2(b[0], b[1], 0u, false), true)};
#endif
    } else {
        static_assert(std::is_same<OutT, bf16_t>::value, "store_out8: outputs narrower than fp32");
#if SPRC_EPI_NT & 1
        __builtin_nontemporal_store(u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])}, reinterpret_cast<u32x4*>(dst));
#else
        *reinterpret_cast<u32x4*>(dst) = u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
#endif
    }
}

// one output element (the ragged-N scalar path)
template <typename OutT>
__device__ __forceinline__ void store_out1(OutT* dst, float v, int n_split = 0, int col = 0) {
    if constexpr (std::is_same<OutT, fp8_t>::value) dst->bits = (uint8_t)(pack_fp8x2(v, 0.f, 0u, false) & 0xffu);
    else if constexpr (std::is_same<OutT, f16x3_t>::value) store_split1(reinterpret_cast<char*>(dst) - 2 * col, n_split, col, v);
    else *dst = (OutT)v;
}

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<f16_t> { typedef f16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };
template <> struct Frag<fp8_t> { typedef u32x4 type; };       // unused: the fp8 main loop is the hand-pipelined one

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Raw-buffer (SRSRC) 16-B-per-lane load straight into LDS.  Kept in NON-template helpers: this hipcc silently drops the
// host stub of a kernel TEMPLATE whose dependent code calls the buffer builtins directly.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xffffffff, 0x00020000);
}
#ifndef SPRC_LD_NT
#define SPRC_LD_NT 0
#endif
template <int AUX = 0>
__device__ __forceinline__ void buffer_load_lds16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, uint32_t voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds_dst, 16, voffset, soffset, 0, AUX);     // AUX 2 = nt (gfx940+ cache policy bits: 1 sc0, 2 nt, 16 sc1)
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ds_read_b128 the compiler cannot sink: issued where written, completion tracked by OUR lgkmcnt (guide 5.7).
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);       // keep the MFMAs below the wait (guide rule 18)
}

// One K-tile (4 MFMA k-steps) of a wave's TM x TN accumulator block, hand software-pipelined.
// a_base / b_base: LDS byte address of this lane's first A / W row in the current stage; c0 = swizzled slot of
// k-step 0 ((half ^ f(row)) << 4); k-step kk reads slot c0 ^ (kk << 5).
// fp8 (FP8 = true): a 16-B fragment holds 16 k-values = TWO e4m3 MFMA k-steps (low / high 8 bytes); which bytes form a
// k-step is a free permutation of k as long as both operands use the same one, so the LDS layout and the reads are the bf16 ones.
// T = operand type of the 16-B fragments: bf16_t, f16_t (v_mfma_f32_32x32x16_f16: same rate, same fragment layout, 11-bit
// significands -- the reference's own GPU precision, blip2.py:36-44) or fp8_t.
template <typename T>
__device__ __forceinline__ f32x16 mfma_frag(const u32x4& b, const u32x4& a, f32x16 c) {
    if constexpr (std::is_same<T, fp8_t>::value) {
        const long b0 = (long)(((uint64_t)b[1] << 32) | b[0]), b1 = (long)(((uint64_t)b[3] << 32) | b[2]);
        const long a0 = (long)(((uint64_t)a[1] << 32) | a[0]), a1 = (long)(((uint64_t)a[3] << 32) | a[2]);
        c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(b0, a0, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(b1, a1, c, 0, 0, 0);
    } else if constexpr (std::is_same<T, f16_t>::value) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
    }
}

// MX-scaled e4m3 MFMA with unit block scales (E8M0 127 = 2^0): v_mfma_scale_f32_32x32x64_f8f6f4 covers K = 64 per instruction
// at TWICE the rate of the non-scaled fp8 / bf16 instructions (5 PF dense).  A lane supplies 32 bytes = two 16-B fragments
// of the bf16 LDS layout; with all scales 1 the k-order inside the 64 is a free permutation shared by both operands.
// SPRC_FP8_MX 0 falls back to two non-scaled 32x32x16 steps per fragment (bf16 rate).
#ifndef SPRC_ANTI_LDPOS
#define SPRC_ANTI_LDPOS 1          // the in-cluster load goes out after MFMA number LDPOS (0..15)
#endif
#ifndef SPRC_FP8_MX
#define SPRC_FP8_MX 1
#endif
typedef __attribute__((ext_vector_type(8))) int i32x8;
// volatile asm: as a pure intrinsic hipcc SINKS the MFMAs of a whole K-tile pair to the loop latch (legal, and fatal for the
// interval structure); the statement stays where it is written.  Callers keep >= 8 other MFMAs between two uses of one
// accumulator (no hazard nops are inserted for inline asm).
// SPRC_MX_FP6_TIMING (variant builds only, tools/r06_ab_fp6.sh; WRONG results): the correction segments' MFMAs issued in the fp6 (e2m3) format on
// the first 24 of the 32 operand bytes -- half the passes of the e4m3 form -- to MEASURE what fp6 correction segments would buy before any
// producer writes them.
#ifndef SPRC_MX_FP6_TIMING
#define SPRC_MX_FP6_TIMING 0
#endif
__device__ __forceinline__ f32x16 mfma_mx8(const i32x8& bb, const i32x8& aa, f32x16 c, int sc = MX_UNIT_SCALE) {
    // sc: the E8M0 block scale of BOTH operands in every byte (0x7f = 2^0: unit scales; MX_SPLIT_SCALE for the split-precision segments)
    if constexpr (SPRC_MX_FP6_TIMING != 0) {
        typedef __attribute__((ext_vector_type(6))) int i32x6;
        const i32x6 b6 = {bb[0], bb[1], bb[2], bb[3], bb[4], bb[5]}, a6 = {aa[0], aa[1], aa[2], aa[3], aa[4], aa[5]};
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(c) : "v"(b6), "v"(a6), "v"(sc));
        return c;
    }
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(bb), "v"(aa), "v"(sc));
    return c;
}
// A 32-B MX operand = the 16-B fragments of two k-steps, read by compiler-tracked LDS loads so that the register allocator
// defines the two halves of the 8-register tuple in place (inline-asm ds_reads into 4-register values cost a copy per
// operand: 48 VGPRs in the 256 x 256 kernel).  The caller fences the region with sched_barrier(0) on both sides.
__device__ __forceinline__ i32x8 lds_pair(uint32_t addr_lo, uint32_t addr_hi) {
    typedef const __attribute__((address_space(3))) u32x4* lp;
    const u32x4 lo = *(lp)(addr_lo), hi = *(lp)(addr_hi);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}
template <int SC = MX_UNIT_SCALE>
__device__ __forceinline__ f32x16 mfma_mx(const u32x4& b0, const u32x4& b1, const u32x4& a0, const u32x4& a1, f32x16 c) {
    const i32x8 bb = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
    const i32x8 aa = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    constexpr int FMT = SPRC_MX_FP6_TIMING != 0 ? 2 /* e2m3: timing variant, see mfma_mx8 */ : 0 /* e4m3 */;
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bb, aa, c, FMT, FMT, 0, SC, 0, SC);
}

// One K-tile of the 128x128 kernel on MX fp8: two K = 64 steps, each from a PAIR of 16-B fragments per operand row-tile; the
// second pair is read and the next K-tile's loads are issued before the MFMAs of the first.
template <int TM, int TN, int KT_BYTES, int SC = MX_UNIT_SCALE, typename Issue>
__device__ __forceinline__ void pipe_ktile_mx(uint32_t a_base, uint32_t b_base, uint32_t c0, f32x16 (&acc)[TM][TN], Issue&& issue) {
    static_assert(KT_BYTES == 128, "two K = 64 steps per K-tile");
    u32x4 fa[2][2][TM], fb[2][2][TN];           // [pair][fragment in pair][row-tile]
    auto read_pair = [&](auto p_) {
        constexpr int pr = decltype(p_)::value;
        static_for<0, 2>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            const uint32_t cn = c0 ^ ((2 * pr + k) << 5), an = a_base + cn, bn = b_base + cn;
            static_for<0, TM>([&](auto i) { fa[pr][k][decltype(i)::value] = lds_read128<decltype(i)::value * 32 * KT_BYTES>(an); });
            static_for<0, TN>([&](auto i) { fb[pr][k][decltype(i)::value] = lds_read128<decltype(i)::value * 32 * KT_BYTES>(bn); });
        });
    };
    read_pair(std::integral_constant<int, 0>{});
    static_for<0, 2>([&](auto p_) {
        constexpr int pr = decltype(p_)::value;
        if constexpr (pr == 0) read_pair(std::integral_constant<int, 1>{});
        issue(std::integral_constant<int, 2 * pr>{});
        issue(std::integral_constant<int, 2 * pr + 1>{});
        if constexpr (pr == 0) wait_lgkmcnt<2 * (TM + TN)>();
        else wait_lgkmcnt<0>();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = mfma_mx<SC>(fb[pr][0][ni], fb[pr][1][ni], fa[pr][0][mi], fa[pr][1][mi], acc[mi][ni]);
        __builtin_amdgcn_s_setprio(0);
    });
}

template <int TM, int TN, int KT_BYTES, typename T, typename Issue>
__device__ __forceinline__ void pipe_ktile_bf16(uint32_t a_base, uint32_t b_base, uint32_t c0, f32x16 (&acc)[TM][TN],
                                                Issue&& issue) {
    static_assert(KT_BYTES == 128, "4 k-steps per K-tile");
    u32x4 fa[2][TM], fb[2][TN];
    {
        const uint32_t an = a_base + c0, bn = b_base + c0;
        static_for<0, TM>([&](auto i) { fa[0][i] = lds_read128<decltype(i)::value * 32 * KT_BYTES>(an); });
        static_for<0, TN>([&](auto i) { fb[0][i] = lds_read128<decltype(i)::value * 32 * KT_BYTES>(bn); });
    }
    static_for<0, 4>([&](auto kk_) {
        constexpr int kk = decltype(kk_)::value, cur = kk & 1, nxt = cur ^ 1;
        if constexpr (kk < 3) {
            const uint32_t cn = c0 ^ ((kk + 1) << 5), an = a_base + cn, bn = b_base + cn;
            static_for<0, TM>([&](auto i) { fa[nxt][i] = lds_read128<decltype(i)::value * 32 * KT_BYTES>(an); });
            static_for<0, TN>([&](auto i) { fb[nxt][i] = lds_read128<decltype(i)::value * 32 * KT_BYTES>(bn); });
        }
        issue(kk_);
        if constexpr (kk < 3) wait_lgkmcnt<TM + TN>();
        else wait_lgkmcnt<0>();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = mfma_frag<T>(fb[cur][ni], fa[cur][mi], acc[mi][ni]);
        __builtin_amdgcn_s_setprio(0);
    });
}

// Tile order.  Block b runs on XCD b%8 (observed dispatch); the remap gives every XCD a CONTIGUOUS range of `pid`s, and
// the pid -> (m,n) map walks a group of GN n-tiles over ALL m-tiles (m fastest inside the group).  The ~32 workgroups
// resident on one XCD then form an 8(m) x 4(n) patch (12 operand panels per 32 tiles), and successive rounds on that XCD
// keep the SAME 4 W panels (2.9 MB, L2-resident) while the A panels stream through once.  (mode 0: the earlier
// 8-m-tile grouped order, where both operands change every round.)
__device__ __forceinline__ void tile_origin(int vb, int nwg, int tiles_m, int tiles_n, int BM, int BN, int mode, int& m0, int& n0) {
    const int xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
    const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    if ((mode & 255) == 0) {
        const int GROUP_M = (mode >> 8) ? (mode >> 8) : 4;     // SPRC_GEMM_ORDER = 0 | GROUP_M << 8 forces another group height (A/B)
        const int in_group = GROUP_M * tiles_n;
        const int first_m = (pid / in_group) * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        m0 = (first_m + (pid % in_group) % gsz) * BM;
        n0 = ((pid % in_group) / gsz) * BN;
    } else {
        const int GN = mode;                      // n-tiles per group
        const int in_group = GN * tiles_m;
        // (this runs before the first load of every tile: one float division instead of three integer ones -- all values
        // are < 2^20, the quotient is exact after one correction step -- and shifts for the divisors that are 32 and 8
        // everywhere but in the last group / last block)
        int grp = (int)((float)pid * __builtin_amdgcn_rcpf((float)in_group));
        int rem = pid - grp * in_group;
        if (rem < 0) { --grp; rem += in_group; }
        if (rem >= in_group) { ++grp; rem -= in_group; }
        const int first_n = grp * GN;
        const int gsz = min(tiles_n - first_n, GN);
        // inside a group: blocks of 8 m-tiles x gsz n-tiles, m fastest
        int blk, inb;
        if (gsz == 4) { blk = rem >> 5; inb = rem & 31; }
        else { blk = rem / (8 * gsz); inb = rem % (8 * gsz); }
        const int mrem = min(tiles_m - blk * 8, 8);
        int im, in;
        if (mrem == 8) { im = inb & 7; in = inb >> 3; }
        else { im = inb % mrem; in = inb / mrem; }
        m0 = (blk * 8 + im) * BM;
        n0 = (first_n + in) * BN;
    }
}

// The kernel's parameter block through a LAUNDERED kernarg pointer: loads through it are scheduled where they are written (the compiler
// cannot merge them with the entry-time loads of the by-value parameter), so a phase can fetch its own parameters when it starts.
typedef const __attribute__((address_space(4))) GemmParams* kparams_t;
__device__ __forceinline__ kparams_t kernarg_params() {
    kparams_t kp = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}

// ---- epilogue (shared).  Transposed 32x32 D layout: m = rbase + (lane&31),  n = cbase + 8*(r>>2) + 4*(lane>>5) + (r&3) ----
// Each wave transposes its accumulators through a private LDS strip (32 rows x (32 TN + 4) floats: the padding makes both
// the ds_write_b128 of the fragment layout and the row-major ds_read_b128 conflict-free) so that global memory sees FULL
// cache lines: one wave instruction covers 64/(8 TN) whole rows x 128 TN bytes.  With the fragment layout written straight
// out, an instruction touched 32 rows x 32 B and a 256 x 256 fp32 tile with residual took ~55k cycles (7 B/clk/CU,
// store-issue bound: s_memtime, tools/gemm_stamp.py) -- more than a third of a K = 1408 tile.
// The caller guarantees every wave is done reading operand tiles from LDS (a barrier after the K loop).
constexpr int EPI_STRIP_BYTES(int TN) { return 32 * (32 * TN + 4) * 4; }

template <typename T, typename OutT, int ACT, bool MAX32, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[TM][TN], int cm0, int cn0, int wr, int wc,
                                              int r32, int half, bool vec_ok, char* smem, int wave) {
    if constexpr (!MAX32) {
        if (vec_ok) {
            // W = columns per lane.  4: one 16-B strip read, 16-B (fp32) / 8-B (16-bit) / 4-B (fp8) stores.  8 (outputs narrower than
            // fp32, N % 8 == 0): two strip reads, ONE 16-B store per lane for 16-bit outputs -- the store tail of a 16-bit epilogue is
            // store-ISSUE-bound (guide T21: half the instructions at equal bytes halve it), and a wave instruction then covers 8 rows x
            // 128 B = whole cache lines.  SPRC_EPI_WIDE=0 keeps W = 4 (A/B switch).
            auto vec_path = [&](auto w_) {
            constexpr int W = decltype(w_)::value, NV = W / 4;
            constexpr int CH = 32 * TN / W, RPI = 64 / CH, ITERS = 32 / RPI, LD = (32 * TN + 4) * 4;   // chunks per row, rows per instruction
            const int lane = half * 32 + r32, ch = lane % CH, rsub = lane / CH;
            char* strip = smem + wave * EPI_STRIP_BYTES(TN);
            const int col = cn0 + wc * TN * 32 + ch * W;
            const bool col_ok = col < p.N;
            f32x4 bv[NV], sv[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) { bv[j] = f32x4{0.f, 0.f, 0.f, 0.f}; sv[j] = f32x4{1.f, 1.f, 1.f, 1.f}; }
            if (p.bias != nullptr && col_ok) {
#pragma unroll
                for (int j = 0; j < NV; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + col + 4 * j);
            }
            const bool scaled = sizeof(T) == 1 && p.w_scale != nullptr;     // fp8 operands: per-output-channel dequantisation (a compile-time
                                                                            // false elsewhere: the run-time flag cost a v_mul + v_cndmask per element)
            if (scaled && col_ok) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    sv[j] = *reinterpret_cast<const f32x4*>(p.w_scale + col + 4 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sv[j][e] *= p.a_scale;
                }
            }
            // Residual rows are fetched TWO 32-row blocks ahead of their use (the fragments' registers are free here): with the
            // loads of a block issued only when the block was reached, every block paid a full memory round trip -- 23 k cycles
            // for a tile alone on the chip, 45-50 k with 256 tiles in their epilogues at once (s_memtime, tools/epi_probe.sh).
            // Loads are branch-free (clamped addresses; "is there a residual" selects one of two straight-line bodies): hipcc
            // drains the memory queue wherever predicated loads meet.
            // (C may alias the residual: every element is read by the lane that writes it, before it writes it.)
            auto block_row = [&](int mi, int it) { return cm0 + (wr * TM + mi) * 32 + it * RPI + rsub; };
            const int colc = col_ok ? col : 0;
            // Row addressing without branches or 64-bit multiplies (the .s of the first form: per store a uniform branch on "is there a row
            // map", two v_mul_lo_u32 + a v_mad_u64_u32 -- quarter-rate -- and 64-bit adds): the row map in its branch-free form (identity =
            // shift 31, stride 0), rows taken RELATIVE to the tile's first mapped row, so that a row's offset is one 24-bit multiply
            // (launchers check the ranges: epi_fits_u32) added to a tile-uniform base pointer.
            const int nz = ~(p.c_shift >> 31), sh = p.c_shift & 31, msk = (int)((1u << sh) - 1u), cstr = p.c_stride & nz, coff = p.c_off & nz;
            auto mrow = [&](int r) { return __mul24(r >> sh, cstr) + (r & msk) + coff; };
            const int row0m = mrow(cm0);
            OutT* const cbase = reinterpret_cast<OutT*>(p.C) + (int64_t)row0m * p.ldc;
            const float* const rbase = p.resid + (int64_t)row0m * p.ldr;
            const uint32_t ldc32 = (uint32_t)p.ldc, ldr32 = (uint32_t)p.ldr;
            // The bias / scale vectors are retired HERE, by a wait on the straight-line path.  Left to hipcc, the wait for these (predicated) loads lands
            // inside the first row's bounds-checked block; the path around that block still carries them as pending, so the wait is repeated in
            // EVERY row's block -- as `s_waitcnt vmcnt(1)` / `vmcnt(0)`, which in hardware waits for the PREVIOUS row's global store to complete:
            // 16 serial store round trips per wave and tile (the 11 k cycles of the 16-bit epilogue; seen in the .s of every 16-bit-output kernel).
            __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0); lgkmcnt / expcnt untouched
            auto body = [&](auto res_) {
                constexpr bool RES = decltype(res_)::value;
                f32x4 rv[2][ITERS][NV];
                auto load_resid = [&](int mi, f32x4 (&dst)[ITERS][NV]) {
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const uint32_t ro = __umul24((uint32_t)(mrow(min(block_row(mi, it), p.M - 1)) - row0m), ldr32) + (uint32_t)colc;
#pragma unroll
#if SPRC_EPI_NT & 4
                        for (int j = 0; j < NV; ++j) dst[it][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rbase + ro + 4 * j));
#else
                        for (int j = 0; j < NV; ++j) dst[it][j] = *reinterpret_cast<const f32x4*>(rbase + ro + 4 * j);
#endif
                    }
                };
                if constexpr (RES) {
                    load_resid(0, rv[0]);
                    if constexpr (TM > 1) load_resid(1, rv[1]);
                }
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<f32x4*>(strip + r32 * LD + (ni * 32 + 8 * g + 4 * half) * 4) =
                                f32x4{acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        f32x4 v[NV];
#pragma unroll
                        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const f32x4*>(strip + (it * RPI + rsub) * LD + ch * (W * 4) + j * 16);
#pragma unroll
                        for (int j = 0; j < NV; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {       // element-wise on purpose: vector adds become v_pk_add_f32 (slower)
                            if (scaled) v[j][e] *= sv[j][e];
                            v[j][e] += bv[j][e];
                            if constexpr (ACT == SPRC_ACT_GELU) {
                                if constexpr (sizeof(T) <= 2) v[j][e] = gelu_fast(v[j][e]);
                                else v[j][e] = gelu_erf(v[j][e]);
                            }
                            if constexpr (ACT == SPRC_ACT_QUICKGELU) v[j][e] = sizeof(T) <= 2 ? quick_gelu_fast(v[j][e]) : quick_gelu(v[j][e]);
                            if constexpr (RES) v[j][e] += rv[mi & 1][it][j][e];
                            if constexpr (std::is_same<OutT, fp8_t>::value) v[j][e] *= p.out_scale;
                        }
                        const int row = block_row(mi, it);
                        if (!(row < p.M && col_ok)) continue;
                        OutT* const dst = cbase + (__umul24((uint32_t)(mrow(row) - row0m), ldc32) + (uint32_t)col);
                        if constexpr (W == 8) store_out8<OutT>(dst, v[0], v[1], p.N, col);
                        else store_out4<OutT>(dst, v[0], p.N, col);
                    }
                    if constexpr (RES) {
                        if (mi + 2 < TM) load_resid(mi + 2, rv[mi & 1]);
                    }
                }
            };
            if (p.resid != nullptr) body(std::true_type{});
            else body(std::false_type{});
            };
            if constexpr (sizeof(OutT) < 4) {
                const bool wide_ok = p.epi_wide && (p.N % 8 == 0) && (p.ldc % 8 == 0) && ((uintptr_t)p.C % 16 == 0);
                if (wide_ok) { vec_path(std::integral_constant<int, 8>{}); return; }
            }
            vec_path(std::integral_constant<int, 4>{});
            return;
        }
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = cm0 + (wr * TM + mi) * 32 + r32;
        if constexpr (MAX32) {
            // rows m = query vectors, columns n = gallery tokens: max over the 32 columns of one image
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int cbase = cn0 + (wc * TN + ni) * 32;
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
            {                                         // unaligned / ragged N: scalar path (the vector path returned above)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = cn0 + (wc * TN + ni) * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                        if (!(row_ok && col < p.N)) continue;
                        float v = acc[mi][ni][r];
                        if (p.w_scale != nullptr) v *= p.a_scale * p.w_scale[col];
                        v += (p.bias ? p.bias[col] : 0.f);
                        if constexpr (ACT == SPRC_ACT_GELU) {
                            if constexpr (sizeof(T) <= 2) v = gelu_fast(v);
                            else v = gelu_erf(v);
                        }
                        if constexpr (ACT == SPRC_ACT_QUICKGELU) v = sizeof(T) <= 2 ? quick_gelu_fast(v) : quick_gelu(v);
                        if (rrow != nullptr) v += rrow[col];
                        if constexpr (std::is_same<OutT, fp8_t>::value) v *= p.out_scale;
                        store_out1<OutT>(crow + col, v, p.N, col);
                    }
            }
        }
    }
}

// NS = LDS stages.  2: K-tile t+1 is fetched while t is multiplied, one vmcnt(0) + barrier per K-tile -- right when several
// workgroups share a CU and cover each other's waits.  4: a ring three K-tiles deep with counted waits, for the launches that
// cannot even give every CU a workgroup (remainder rows, the small Q-Former products): there a K-tile cost a full memory
// round trip (~1.5 k cycles for 4 MFMAs; the 128-row ViT remainders 20 us per launch).
// MIX (fp16 operands only): the split-precision product -- the K fp16 elements of every operand row are followed by p.k8 e4m3 elements
// (sprc.h: SPRC_F16X3); their K-tiles run on the MX-scaled MFMA into the same accumulators.
template <typename T, typename OutT, int ACT, bool MAX32, int WM, int WN, int TM, int TN, int NS = 2, bool MIX = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(NS == 2 || (NS == 4 && sizeof(T) <= 2), "stages: 2, or a ring of 4 for 1- and 2-byte operands");
    static_assert(!MIX || std::is_same<T, f16_t>::value, "MIX: fp16 main segment");
    constexpr int KT_BYTES = 128;
    constexpr int NT = 64 * WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int STAGE_BYTES = (BM + BN) * KT_BYTES;
    constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;          // 16-B chunks per thread per K-tile
    constexpr int LQ = (LA + LB) / 4;                          // global->LDS loads issued per MFMA k-step
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0 && (LA + LB) % 4 == 0, "tile/threads mismatch");
    typedef typename Frag<T>::type frag_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int r32 = lane & 31, half = lane >> 5;
    const int nwg = p.tiles_m * p.tiles_n;
    int nt = (int)(((int64_t)p.K * sizeof(T)) / KT_BYTES);
    int nt16 = nt;                                              // MIX: K-tiles [0, nt16) are fp16, [nt16, nt) e4m3
    if constexpr (MIX) nt += p.k8 / KT_BYTES;
    int vb = blockIdx.x, ks = 0;
    if (p.dual && vb >= p.nwg0) {                               // second product of a paired launch
        vb -= p.nwg0;
        p.W = p.W2; p.bias = p.bias2; p.a_off = p.a_off2; p.c_off = p.c_off2;
    }
    int64_t kbase = 0;                                          // byte offset of this workgroup's first K-tile
    if (p.ksplit > 1) {                                         // split-K: workgroup (tile vb, split ks) reduces K-tiles [t0, t0 + nt)
        ks = vb / nwg;
        vb -= ks * nwg;
        const int per = (nt + p.ksplit - 1) / p.ksplit, t0 = min(ks * per, nt);
        nt = min(per, nt - t0);
        nt16 = max(0, min(nt16 - t0, nt));
        kbase = (int64_t)t0 * KT_BYTES;
    }

    // ---- direct-to-LDS staging: lane fills physical slot (chunk&7) of row (chunk>>3) with logical slot^f(row) ----
    // SRSRC buffer loads (workgroup-uniform base = first row of the tile, 32-bit lane offsets, K-tile offset in an SGPR):
    // no per-load address VALU, and they issue 2-3x faster than global_load_lds with 64-bit lane addresses.
    int m0, n0;
    tile_origin(vb, nwg, p.tiles_m, p.tiles_n, BM, BN, p.order, m0, n0);
    m0 = __builtin_amdgcn_readfirstlane(m0);                // (computed on the VALU: pinned to SGPRs, see gemm_anti_kernel)
    n0 = __builtin_amdgcn_readfirstlane(n0);
    const int64_t a_row0 = map_row_s(p.a_shift, p.a_stride, p.a_off, m0);
    const char* a_base = p.A + a_row0 * p.lda_b;            // resources are rebuilt from these at each use (loop-invariant SGPRs)
    const char* w_base = p.W + (int64_t)n0 * p.ldw_b;
    uint32_t a_src[LA], w_src[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int c = i * NT + tid, row = c >> 3, slot = (c & 7) ^ ((row >> 1) & 7);
        const int am = min(m0 + row, p.M - 1);
        a_src[i] = (uint32_t)((map_row_s(p.a_shift, p.a_stride, p.a_off, am) - a_row0) * p.lda_b) + slot * 16;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int c = i * NT + tid, row = c >> 3, slot = (c & 7) ^ ((row >> 1) & 7);
        w_src[i] = (uint32_t)((int64_t)(min(n0 + row, p.N - 1) - n0) * p.ldw_b) + slot * 16;
    }
    auto stage_one = [&](auto j_, char* dst, int64_t ko) {      // j-th of the LA+LB loads of one K-tile
        constexpr int j = decltype(j_)::value;
        if constexpr (j < LA)
            buffer_load_lds16(make_rsrc(a_base), dst + j * NT * 16, a_src[j], (int)ko);
        else
            buffer_load_lds16(make_rsrc(w_base), dst + BM * KT_BYTES + (j - LA) * NT * 16, w_src[j - LA], (int)ko);
    };

    const int sw = (r32 >> 1) & 7;
    const int a_off = (wr * TM * 32 + r32) * KT_BYTES, b_off = BM * KT_BYTES + (wc * TN * 32 + r32) * KT_BYTES;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t c0 = (uint32_t)((half ^ sw) << 4);
    const bool vec_ok = (p.N % 4 == 0) && (p.ldc % 4 == 0) && (p.resid == nullptr || p.ldr % 4 == 0) &&
                        ((uintptr_t)p.C % (4 * sizeof(OutT)) == 0) && ((uintptr_t)p.bias % 16 == 0) && ((uintptr_t)p.resid % 16 == 0);

    static_for<0, NS - 1>([&](auto d_) {                        // K-tiles 0 .. NS-2
        constexpr int d = decltype(d_)::value;
        if (d < nt) {
            char* dst0 = smem + d * STAGE_BYTES + wave * 1024;
            static_for<0, LA + LB>([&](auto j_) { stage_one(j_, dst0, kbase + (int64_t)d * KT_BYTES); });
        }
    });

    {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        // ---- main loop: K-tile t is in buffer t&1 (tile 0 was issued before we got here) ----
        for (int t = 0; t < nt; ++t) {
            if constexpr (NS == 2) {
                __syncthreads();             // tile t landed (vmcnt(0) + barrier); buffer (t+1)&1 is free again
            } else {                         // ring: K-tiles up to t + NS - 2 are in flight, tile t has to have landed
                constexpr int L = LA + LB;
                const int later = min(NS - 2, nt - 1 - t);
                if (later >= 2) wait_vmcnt<2 * L>();
                else if (later == 1) wait_vmcnt<L>();
                else wait_vmcnt<0>();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();    // ... in every wave; and everyone is done reading stage (t - 1) % NS, refilled below
                asm volatile("" ::: "memory");
            }
            const bool more = t + NS - 1 < nt;
            const int64_t ko = kbase + (int64_t)(t + NS - 1) * KT_BYTES;
            if constexpr (sizeof(T) <= 2) {
                char* dst = smem + ((t + NS - 1) & (NS - 1)) * STAGE_BYTES + wave * 1024;
                const uint32_t so = lds0 + (t & (NS - 1)) * STAGE_BYTES;
                auto issue = [&](auto kk_) {
                    constexpr int kk = decltype(kk_)::value;
                    if (more) static_for<kk * LQ, (kk + 1) * LQ>([&](auto j_) { stage_one(j_, dst, ko); });
                };
                if constexpr (sizeof(T) == 1 && SPRC_FP8_MX) pipe_ktile_mx<TM, TN, KT_BYTES>(so + a_off, so + b_off, c0, acc, issue);
                else if constexpr (MIX) {
                    if (t < nt16) pipe_ktile_bf16<TM, TN, KT_BYTES, T>(so + a_off, so + b_off, c0, acc, issue);
                    else pipe_ktile_mx<TM, TN, KT_BYTES, MX_SPLIT_SCALE>(so + a_off, so + b_off, c0, acc, issue);
                } else pipe_ktile_bf16<TM, TN, KT_BYTES, T>(so + a_off, so + b_off, c0, acc, issue);
            } else {
                if (more) {
                    char* dst = smem + ((t + 1) & 1) * STAGE_BYTES + wave * 1024;
                    static_for<0, LA + LB>([&](auto j_) { stage_one(j_, dst, ko); });
                }
                const char* st = smem + (t & 1) * STAGE_BYTES;
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
                for (int kk = 0; kk < 4; ++kk) {
                    if (kk + 1 < 4) load((kk + 1) & 1, kk + 1);
                    // each lane holds 4 consecutive k of its half; MFMA step e pairs k = {8kk+e, 8kk+4+e}
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                            for (int ni = 0; ni < TN; ++ni)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[kk & 1][ni][e], fa[kk & 1][mi][e], acc[mi][ni], 0, 0, 0);
                }
            }
        }

        __syncthreads();                     // every wave is done reading operand tiles: the epilogue reuses LDS
        if (p.ksplit > 1) {
            GemmParams pe = p;
            pe.C = reinterpret_cast<OutT*>(p.C) + ks * p.split_stride;
            gemm_epilogue<T, OutT, ACT, MAX32, TM, TN>(pe, acc, m0, n0, wr, wc, r32, half, vec_ok, smem, wave);
        } else {
            gemm_epilogue<T, OutT, ACT, MAX32, TM, TN>(p, acc, m0, n0, wr, wc, r32, half, vec_ok, smem, wave);
        }
    }
}

// out = act(sum_s partial[s] + bias) + resid for the split-K remainder launch (fixed summation order: deterministic)
template <typename OutT, int ACT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int64_t stride, int M, int N,
                                                            const float* __restrict__ bias, const float* resid, int64_t ldr,
                                                            OutT* C, int64_t ldc, const float* __restrict__ w_scale, float a_scale,
                                                            float out_scale) {
    const int n4 = N / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)M * n4) return;
    const int row = (int)(i / n4), col = (int)(i % n4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + (int64_t)row * N + col);
    for (int s = 1; s < S; ++s) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(part + s * stride + (int64_t)row * N + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += w[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (w_scale != nullptr) v[e] *= a_scale * w_scale[col + e];
        if (bias != nullptr) v[e] += bias[col + e];
        if constexpr (ACT == SPRC_ACT_GELU) v[e] = gelu_fast(v[e]);
        if constexpr (ACT == SPRC_ACT_QUICKGELU) v[e] = quick_gelu(v[e]);
        if (resid != nullptr) v[e] += resid[(int64_t)row * ldr + col + e];
        if constexpr (std::is_same<OutT, fp8_t>::value) v[e] *= out_scale;
        store_out1<OutT>(C + (int64_t)row * ldc + col + e, v[e], N, col + e);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Anti-phase bf16 variant of the 256 x 256 tile (the schedule idea of the guide's 8-phase template, T3/T4/T5).
// PMC on the lock-step kernel: matrix pipe busy 55 %, waves 54 % in issue stalls + 28 % in s_waitcnt/barrier -- the two
// waves that share a SIMD hit their MFMA clusters at the same time and then sit at the K-tile barrier together.
// Here the 8 waves form two groups (G0 = waves 0-3, G1 = waves 4-7; wave w and w+4 share a SIMD) that alternate:
// in every barrier interval ONE group runs a 16-MFMA cluster (half a K-tile: k-steps 2h, 2h+1) while the OTHER reads its
// next fragments from LDS (12 ds_read_b128) and waits for them.  G1 lags G0 by one interval (one extra s_barrier up
// front, one fewer at the end), so per K-tile a wave executes  NC(t,0) | C(t,0) | NC(t,1) | C(t,1)  with a raw s_barrier
// after each, and on every SIMD the matrix pipe always has exactly one producer.
// Staging (measured with s_memtime, tools/gemm_stamp.py: an NC interval carrying 12 ds_read_b128 + 4 global->LDS loads takes
// 500-700 cycles to ISSUE against a 560-cycle cluster, and draining to vmcnt(0) costs up to 500 more): the K-tile is cut
// into eight 8-KB pieces (64 rows x 128 B: A0..A3, B0..B3; group g reads A(2g), A(2g+1) and every B), two loads per thread
// per piece, TWO pieces per interval pair and group (as SRSRC buffer loads: a global_load_lds with its 64-bit lane
// addresses took 2-3x longer to issue) -- three loads in the NC interval, the fourth inside the following cluster -- and
// every wait is a counted vmcnt(3/4) that leaves the newest pieces in flight:
//   interval     4t-1            4t              4t+1            4t+2            4t+3
//   issues       G1: A0 A1(t+1)  G0: B0 B1(t+1)  G1: B2 B3(t+1)  G0: A2 A3(t+1)  G1: A0 A1(t+2)
//   first read of tile t+1: A0 A1 B* in 4t+4 (G0), A2 A3 in 4t+5 (G1); last read of tile t-1: A0 A1 in 4t-2, rest in 4t-1.
//   RAW  the issuing wave retires a piece (counted vmcnt) before the barrier closing the interval BEFORE its first read:
//        G0 after C(t,1) (4t+3: B0 B1 of t+1) and after NC(t,0) (4t: A2 A3 of t); G1 after NC(t,1) (4t+3: A0 A1 B2 B3 of
//        t+1).  Every piece has >= 2 intervals between issue and wait.
//   WAR  a piece is restaged >= 1 interval after the barrier that followed the last read of the region it overwrites.
// MIX (T = fp16): split-precision product.  Operand rows = K fp16 elements + p.k8 e4m3 elements; the K loop runs the fp16 K-tiles on
// v_mfma_f32_32x32x16_f16 and then the e4m3 K-tiles on the MX-scaled MFMA (block scales 2^-9 x 2^-9) into the SAME accumulators: the
// staging, the LDS image and the interval schedule do not know the element type (a K-tile is 128 bytes of a row either way); only the
// fragment registers and the cluster's instruction differ -- two copies of the steady interval pair, selected at compile time.
template <typename T, typename OutT, int ACT, bool MAX32, bool STAMP = false, bool MIX = false>
__global__ __launch_bounds__(512) void gemm_anti_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(sizeof(T) <= 2, "16-bit or fp8 operands");
    static_assert(!MIX || (std::is_same<T, f16_t>::value && SPRC_FP8_MX), "MIX: fp16 main segment + MX e4m3 segments");
    constexpr bool FP8 = sizeof(T) == 1;
    constexpr int WN = 4, TM = 4, TN = 2, KTB = 128;
    constexpr int BM = 256, BN = 256;
    // PERSISTENT tile loop: workgroup b takes tiles b, b + gridDim, ... (the launcher sizes the grid: one workgroup per tile, or -- SPRC_GEMM_PERSIST
    // -- one per CU; gridDim is a multiple of 8 there, so a workgroup's tiles keep its XCD).  Nothing but the tile index lives across an
    // iteration: every tile re-derives its state from the (laundered) kernarg pointer and a laundered thread index, so that no
    // loop-invariant register survives the epilogue.
    int total_tiles;
    { kparams_t k0 = kernarg_params(); total_tiles = k0->tiles_m * k0->tiles_n * (k0->dual ? 2 : 1); }
    for (int vb0 = blockIdx.x; vb0 < total_tiles; vb0 += (int)gridDim.x) {
    kparams_t kq = kernarg_params();
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = __builtin_amdgcn_readfirstlane(wave / WN), wc = wave % WN;   // wr doubles as the phase group (SGPR: barriers under it)
    const int r32 = lane & 31, half = lane >> 5;
    const int nwg = kq->tiles_m * kq->tiles_n;
    const int nt16 = MIX ? kq->K / 64 : 0;                  // MIX: fp16 K-tiles [0, nt16) (even), then e4m3 K-tiles
    const int nt = FP8 ? kq->K / 128 : kq->K / 64 + (MIX ? kq->k8 / 128 : 0);       // K-tiles of 128 bytes
    uint64_t tile_ts[4] = {0, 0, 0, 0};                      // STAMP build: entry | prologue done | K loop done | epilogue done
    uint64_t pro_ts[3] = {0, 0, 0};                         // STAMP build, inside the prologue: setup done | loads issued | loads landed
    if constexpr (STAMP) tile_ts[0] = __builtin_amdgcn_s_memtime();

    int vb = vb0;
    const bool second = kq->dual && vb >= kq->nwg0;         // second product of a paired launch
    if (second) vb -= kq->nwg0;
    const char* const w_sel = second ? kq->W2 : kq->W;
    const int a_off_s = second ? kq->a_off2 : kq->a_off;
    int m0, n0;
    tile_origin(vb, nwg, kq->tiles_m, kq->tiles_n, BM, BN, kq->order, m0, n0);
    // tile_origin divides in floating point, i.e. on the VALU: its (wave-uniform) results sit in VGPRs, and whether hipcc moves them to
    // SGPRs or builds everything downstream -- the two buffer descriptors included -- on the VALU depends on how many vector uses the
    // rest of the kernel has for them.  A descriptor in VGPRs is a v_readfirstlane waterfall loop around EVERY LDS-DMA load (guide T20;
    // seen after an epilogue change: 16 loops per K-tile pair, every product 3-9 % slower).  Pin them.
    m0 = __builtin_amdgcn_readfirstlane(m0);
    n0 = __builtin_amdgcn_readfirstlane(n0);
    // DEAD waves: a wave whose 64 columns lie at or beyond N (N = 1408 = 5.5 tiles: in every sixth-column tile of proj / fc2 the waves of column
    // quarters 2 and 3, a twelfth of those launches' matrix work; qkv's 17th column likewise) multiplies clamped copies of row N - 1 into
    // accumulators nobody stores.  The step is POWER-bound (a product alone on half the chip runs 1.25 x the clock: profiles/r06_cumask_probe.txt),
    // so what those MFMAs and fragment reads cost is clock for everybody else.  A dead wave keeps its share of the staging loads, the counted
    // waits and every barrier -- the interval skeleton below, without reads and clusters -- and skips the epilogue.  SPRC_GEMM_DEAD=0 (debug
    // bit 128): A/B switch; same box, alternating: 87.5 / 87.9 / 88.1 ms per step with, 88.4 / 88.6 / 88.5 without (profiles/r06_dead_waves_ab.txt).
    // (the same for a wave whose 128 ROWS lie at or beyond M: the lower half of a ragged last row panel)
    const bool dead = !(kq->debug & 128) && (__builtin_amdgcn_readfirstlane(n0 + wc * (TN * 32)) >= kq->N || m0 + wr * (TM * 32) >= kq->M);
    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    };
    uint64_t ts[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // ---- K loop.  The first form of this loop (round 1 / early round 2) followed the schedule above with run-time flags: its
    // s_memtime stamps showed an NC interval taking 500-600 cycles to ISSUE 12 ds_read_b128 + 3 loads -- ~20 branches per
    // K-tile (last tile? pieces left? which group?), per-read address arithmetic and a VGPR -> readfirstlane -> M0 chain in
    // front of every load -- and a 640-cycle cluster (the in-cluster load queued behind the other group's).  This form keeps
    // the schedule and the waits and strips the instruction stream: the steady state (K-tiles t+1 and t+2 exist) is unrolled
    // by two so the stage parity is a compile-time constant, fragment addresses are 8 loop-invariant VGPRs + immediates
    // (stages interleaved as [A s0 | A s1 | B s0 | B s1]: a parity is a 32-KB immediate), the LDS destination of a load is
    // SGPR + immediate, and the only branches left are the group's three counted waits; the last (up to three) K-tiles run
    // the same intervals on run-time flags.  NC issue 260 cycles, cluster 520, K-tile 3160 -> 2190 cycles (2048 = MFMA only).
    // ONE code path for both groups on purpose: per-group copies of the tail made the allocator spill 2400 registers at the
    // merge of the accumulator tuples.
    // (Round 2 tried a persistent tile loop that kept the kernel parameters and both tiles' descriptors live across the epilogue: they
    // spilled SGPRs into the steady loop -- 11 v_readlane per K-tile pair -- and it lost 4-9 %.  The round-5 loop above keeps nothing live:
    // tools/kloop_stat.py on a listing shows the steady loop unchanged.  Issuing the next tile's first K-tile before the epilogue, with the
    // strips moved out of the parity-0 stages, was measured on top of it: correct, no gain -- profiles/r05_early_ktile0_ab.txt.)
    const int ltid = tid & 255;
    const uint32_t wg_off = (uint32_t)__builtin_amdgcn_readfirstlane((wave & 3) * 1024);
    const int pc_row[4] = {wr ? 128 : 0, wr ? 192 : 64, wr ? 0 : 128, wr ? 64 : 192};  // first row of piece q in its operand
    uint32_t pc_off[4][2];                                  // byte offset of this lane's 16-B chunk from the tile's first row (K-tile 0)
    const int64_t a_row0 = map_row_s(kq->a_shift, kq->a_stride, a_off_s, m0);
    const char* a_base = kq->A + a_row0 * kq->lda_b;
    const char* w_base = w_sel + (int64_t)n0 * kq->ldw_b;
    // everything up to the first load is exposed once per tile (s_memtime: 2-3.5 k cycles of a 75 k tile before this diet):
    // offsets are < 2^32 by fits_u32(), rows of a tile < 2^8 -> 24-bit multiplies on the plain row map
    const bool plain_a = kq->a_shift < 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = j * 256 + ltid, row = pc_row[q] + (c >> 3);
            const int slot = (c & 7) ^ ((row >> 1) & 7);
            uint32_t o;
            if (q >= 2) {
                if (plain_a) o = __umul24((uint32_t)(min(m0 + row, kq->M - 1) - m0), (uint32_t)kq->lda_b);
                else o = (uint32_t)((map_row_s(kq->a_shift, kq->a_stride, a_off_s, min(m0 + row, kq->M - 1)) - a_row0) * kq->lda_b);
            } else {
                o = __umul24((uint32_t)(min(n0 + row, kq->N - 1) - n0), (uint32_t)kq->ldw_b);
            }
            pc_off[q][j] = o + slot * 16;
        }
    }
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(a_base), rs_w = make_rsrc(w_base);
    using std::integral_constant;
    typedef integral_constant<int, 0> I0;
    typedef integral_constant<int, 1> I1;
    typedef integral_constant<int, 2> I2;
    typedef integral_constant<int, 3> I3;


    constexpr uint32_t PAR_BYTES = 256 * KTB, B_BASE = 2 * PAR_BYTES;   // [A s0 | A s1 | B s0 | B s1], 32 KB each
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t c0 = (uint32_t)((half ^ ((r32 >> 1) & 7)) << 4);
    uint32_t addr_a[4], addr_b[4];                          // per (h, k): k-step 2h + k of a K-tile
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        addr_a[x] = lds0 + (wr * TM * 32 + r32) * KTB + (c0 ^ (uint32_t)(x << 5));
        addr_b[x] = lds0 + B_BASE + (wc * TN * 32 + r32) * KTB + (c0 ^ (uint32_t)(x << 5));
    }
    constexpr bool MX = FP8 && SPRC_FP8_MX;
    typedef std::integral_constant<bool, MIX || MX> MXL;    // element kind of a K-tile: MXL for every tile of a plain kernel; MIX kernels
                                                            // pass false_type for their fp16 tiles and MXL (= true) for the e4m3 ones
    constexpr int MXSC = MIX ? MX_SPLIT_SCALE : MX_UNIT_SCALE;
    u32x4 fa[2][TM], fb[2][TN];
    i32x8 fa8[TM], fb8[TN];                                 // MX: both k-steps of a cluster in one 8-register operand
    // (A residual PREFETCH -- throw-away dword loads over the tile's residual lines in the last K-tile -- paid while the epilogue
    // fetched each 32-row block only when it reached it; with the epilogue's own two-blocks-ahead loads it COSTS 2-7 % of the
    // fp32 + residual products: a dword load per 128-B line is 64 lines per wave instruction.  Removed.)
    auto barrier = [&]() {                                  // nothing -- MFMAs included -- is scheduled across it
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };

    // wave-uniform (SGPR) staging constants: LDS byte offset of piece q in parity 0 (+ this wave's 1-KB share of a 4-KB load)
    uint32_t base_q[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) base_q[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)((q >= 2 ? 0u : B_BASE) + pc_row[q] * KTB + wg_off));
    // load j (of 2) of piece q (G0: B0 B1 | A2 A3, G1: B2 B3 | A0 A1) of K-tile `tile` into the stage of parity par_bytes / 32 KB
    auto load_piece = [&](auto q_, auto j_, uint32_t par_bytes, int tile) {
        constexpr int q = decltype(q_)::value, j = decltype(j_)::value;
        // SPRC_LD_NT (A/B builds only): bit 0 / 1 = nt on the A / W loads (measured: +3-4 ms per step), bit 2 / 3 = sc0 on them
        if constexpr (q >= 2) buffer_load_lds16<((SPRC_LD_NT & 1) ? 2 : 0) | ((SPRC_LD_NT & 4) ? 1 : 0)>(rs_a, smem + (base_q[q] + par_bytes + j * 4096), pc_off[q][j], tile * KTB);
        else buffer_load_lds16<((SPRC_LD_NT & 2) ? 2 : 0) | ((SPRC_LD_NT & 8) ? 1 : 0)>(rs_w, smem + (base_q[q] + par_bytes + j * 4096), pc_off[q][j], tile * KTB);
    };
    auto piece = [&](auto q_, uint32_t par_bytes, int tile) { load_piece(q_, I0{}, par_bytes, tile); load_piece(q_, I1{}, par_bytes, tile); };
    // fragments of k-steps 2h, 2h+1 of the K-tile in the stage of parity PAR: compile-time parity = pure immediates
    auto reads_c = [&](auto par_, auto h_, auto mx_) {
        constexpr int PAR = decltype(par_)::value, h = decltype(h_)::value;
        if constexpr (decltype(mx_)::value) {
            static_for<0, TM>([&](auto i) {
                constexpr uint32_t off = PAR * PAR_BYTES + decltype(i)::value * 32 * KTB;
                fa8[decltype(i)::value] = lds_pair(addr_a[2 * h] + off, addr_a[2 * h + 1] + off);
            });
            static_for<0, TN>([&](auto i) {
                constexpr uint32_t off = PAR * PAR_BYTES + decltype(i)::value * 32 * KTB;
                fb8[decltype(i)::value] = lds_pair(addr_b[2 * h] + off, addr_b[2 * h + 1] + off);
            });
            return;
        }
        static_for<0, 2>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            static_for<0, TM>([&](auto i) { fa[k][i] = lds_read128<PAR * PAR_BYTES + decltype(i)::value * 32 * KTB>(addr_a[2 * h + k]); });
            static_for<0, TN>([&](auto i) { fb[k][i] = lds_read128<PAR * PAR_BYTES + decltype(i)::value * 32 * KTB>(addr_b[2 * h + k]); });
        });
    };
    auto reads_r = [&](uint32_t par_bytes, auto h_, auto mx_) {       // run-time parity (the last K-tiles)
        constexpr int h = decltype(h_)::value;
        if constexpr (decltype(mx_)::value) {
            static_for<0, TM>([&](auto i) {
                constexpr uint32_t off = decltype(i)::value * 32 * KTB;
                fa8[decltype(i)::value] = lds_pair(addr_a[2 * h] + par_bytes + off, addr_a[2 * h + 1] + par_bytes + off);
            });
            static_for<0, TN>([&](auto i) {
                constexpr uint32_t off = decltype(i)::value * 32 * KTB;
                fb8[decltype(i)::value] = lds_pair(addr_b[2 * h] + par_bytes + off, addr_b[2 * h + 1] + par_bytes + off);
            });
            return;
        }
        static_for<0, 2>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            const uint32_t an = addr_a[2 * h + k] + par_bytes, bn = addr_b[2 * h + k] + par_bytes;
            static_for<0, TM>([&](auto i) { fa[k][i] = lds_read128<decltype(i)::value * 32 * KTB>(an); });
            static_for<0, TN>([&](auto i) { fb[k][i] = lds_read128<decltype(i)::value * 32 * KTB>(bn); });
        });
    };
    // 16 MFMAs (k-steps 2h, 2h+1 of the 128 x 64 wave tile); the second load of piece q goes out after MFMA number
    // SPRC_ANTI_LDPOS (`ld`: a compile-time true in the steady state, a run-time flag in the last K-tiles)
    auto cluster = [&](auto q_, uint32_t par_bytes, auto ld, int tile, auto mx_) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (decltype(mx_)::value) {
            static_for<0, 8>([&](auto x_) {
                constexpr int x = decltype(x_)::value, mi = x >> 1, ni = x & 1;
                acc[mi][ni] = mfma_mx8(fb8[ni], fa8[mi], acc[mi][ni], MXSC);
                if constexpr (x == (SPRC_ANTI_LDPOS) / 2) { if (ld) load_piece(q_, I1{}, par_bytes, tile); }
            });
        } else {
            static_for<0, 16>([&](auto x_) {
                constexpr int x = decltype(x_)::value, k = x >> 3, mi = (x >> 1) & 3, ni = x & 1;
                acc[mi][ni] = mfma_frag<T>(fb[k][ni], fa[k][mi], acc[mi][ni]);
                if constexpr (x == (SPRC_ANTI_LDPOS)) { if (ld) load_piece(q_, I1{}, par_bytes, tile); }
            });
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto stamp = [&](auto i_, int t) {
        if constexpr (STAMP) {
            if (t == 8 && ((kq->ksplit >> decltype(i_)::value) & 1)) ts[decltype(i_)::value] = __builtin_amdgcn_s_memtime();   // ksplit = stamp mask here
        }
    };
    // NC(t,1) stages K-tile t+1 (G0) or t+2 (G1): tile index and stage parity of that piece pair
    const uint32_t p2_bytes[2] = {wr ? 0u : PAR_BYTES, wr ? PAR_BYTES : 0u};      // indexed by the parity of t
    // steady state: K-tiles t+1 and t+2 exist, parity of t known at compile time -> no branch but the group's waits
    auto steady = [&](auto par_, int t, auto mx_) {
        constexpr int PAR = decltype(par_)::value;
        constexpr uint32_t pn = (PAR ^ 1) * PAR_BYTES;
        const uint32_t p2 = p2_bytes[PAR];
        const int t_p2 = t + 1 + wr;
        stamp(integral_constant<int, 0>{}, t);
        reads_c(par_, I0{}, mx_);                           // NC(t,0)
        piece(I0{}, pn, t + 1);
        load_piece(I1{}, I0{}, pn, t + 1);
        stamp(integral_constant<int, 1>{}, t);
        wait_lgkmcnt<0>();
        if (wr == 0) wait_vmcnt<3>();                       // A2 A3 of t landed (G1 reads them in the next interval)
        stamp(integral_constant<int, 2>{}, t);
        barrier();
        stamp(integral_constant<int, 3>{}, t);
        cluster(I1{}, pn, std::true_type{}, t + 1, mx_);    // C(t,0)
        stamp(integral_constant<int, 4>{}, t);
        barrier();
        stamp(integral_constant<int, 5>{}, t);
        reads_c(par_, I1{}, mx_);                           // NC(t,1)
        piece(I2{}, p2, t_p2);
        load_piece(I3{}, I0{}, p2, t_p2);
        stamp(integral_constant<int, 6>{}, t);
        wait_lgkmcnt<0>();
        if (wr == 1) wait_vmcnt<3>();                       // A0 A1 B2 B3 of t+1 landed; A0 A1 of t+2 may fly
        stamp(integral_constant<int, 7>{}, t);
        barrier();
        stamp(integral_constant<int, 8>{}, t);
        cluster(I3{}, p2, std::true_type{}, t_p2, mx_);     // C(t,1)
        stamp(integral_constant<int, 9>{}, t);
        if (wr == 0) wait_vmcnt<4>();                       // B0 B1 of t+1 landed; A2 A3 of t+1 may fly
        stamp(integral_constant<int, 10>{}, t);
        barrier();
        stamp(integral_constant<int, 11>{}, t);
    };
    // the last (up to three) K-tiles: the same interval structure on run-time flags
    auto tail = [&](int t, bool n1, bool n2, auto mx_) {
        const uint32_t pb = (uint32_t)(t & 1) * PAR_BYTES, pn = pb ^ PAR_BYTES;
        const uint32_t p2 = wr ? pb : pn;
        const int t_p2 = t + 1 + wr;
        const bool has_p2 = wr ? n2 : n1;
        reads_r(pb, I0{}, mx_);                             // NC(t,0)
        if (n1) { piece(I0{}, pn, t + 1); load_piece(I1{}, I0{}, pn, t + 1); }
        wait_lgkmcnt<0>();
        if (wr == 0) { if (n1) wait_vmcnt<3>(); else wait_vmcnt<0>(); }
        barrier();
        cluster(I1{}, pn, n1, t + 1, mx_);                  // C(t,0)
        barrier();
        reads_r(pb, I1{}, mx_);                             // NC(t,1)
        if (has_p2) { piece(I2{}, p2, t_p2); load_piece(I3{}, I0{}, p2, t_p2); }
        wait_lgkmcnt<0>();
        if (wr == 1) { if (n2) wait_vmcnt<3>(); else wait_vmcnt<0>(); }
        barrier();
        cluster(I3{}, p2, has_p2, t_p2, mx_);               // C(t,1)
        if (wr == 0) { if (n1) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
        barrier();
    };
    // the same intervals for a DEAD wave: its loads in the same order (the counted waits count the same loads), its waits, every barrier
    auto steady_dead = [&](auto par_, int t) {
        constexpr int PAR = decltype(par_)::value;
        constexpr uint32_t pn = (PAR ^ 1) * PAR_BYTES;
        const uint32_t p2 = p2_bytes[PAR];
        const int t_p2 = t + 1 + wr;
        piece(I0{}, pn, t + 1);                             // NC(t,0)
        load_piece(I1{}, I0{}, pn, t + 1);
        if (wr == 0) wait_vmcnt<3>();
        barrier();
        load_piece(I1{}, I1{}, pn, t + 1);                  // C(t,0): the in-cluster load
        barrier();
        piece(I2{}, p2, t_p2);                              // NC(t,1)
        load_piece(I3{}, I0{}, p2, t_p2);
        if (wr == 1) wait_vmcnt<3>();
        barrier();
        load_piece(I3{}, I1{}, p2, t_p2);                   // C(t,1)
        if (wr == 0) wait_vmcnt<4>();
        barrier();
    };
    auto tail_dead = [&](int t, bool n1, bool n2) {
        const uint32_t pb = (uint32_t)(t & 1) * PAR_BYTES, pn = pb ^ PAR_BYTES;
        const uint32_t p2 = wr ? pb : pn;
        const int t_p2 = t + 1 + wr;
        const bool has_p2 = wr ? n2 : n1;
        if (n1) { piece(I0{}, pn, t + 1); load_piece(I1{}, I0{}, pn, t + 1); }
        if (wr == 0) { if (n1) wait_vmcnt<3>(); else wait_vmcnt<0>(); }
        barrier();
        if (n1) load_piece(I1{}, I1{}, pn, t + 1);
        barrier();
        if (has_p2) { piece(I2{}, p2, t_p2); load_piece(I3{}, I0{}, p2, t_p2); }
        if (wr == 1) { if (n2) wait_vmcnt<3>(); else wait_vmcnt<0>(); }
        barrier();
        if (has_p2) load_piece(I3{}, I1{}, p2, t_p2);
        if (wr == 0) { if (n1) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
        barrier();
    };
    // prologue: K-tile 0 resident for everyone (each group stages its four pieces), G1's early pieces of K-tile 1 in
    // flight, G1 one interval behind
    if constexpr (STAMP) pro_ts[0] = __builtin_amdgcn_s_memtime();
    static_for<0, 4>([&](auto q_) { piece(q_, 0u, 0); });
    if constexpr (STAMP) pro_ts[1] = __builtin_amdgcn_s_memtime();
    if (wr == 1 && nt > 1) { piece(I2{}, PAR_BYTES, 1); piece(I3{}, PAR_BYTES, 1); }
    asm volatile("" ::: "memory");
    zero_acc();                                         // 128 v_mov under the first loads' latency
    asm volatile("" ::: "memory");
    if (wr == 1 && nt > 1) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    if constexpr (STAMP) pro_ts[2] = __builtin_amdgcn_s_memtime();
    barrier();
    if (wr == 1) barrier();
    if constexpr (STAMP) tile_ts[1] = __builtin_amdgcn_s_memtime();
    if (dead) {
        int t = 0;
        for (; t + 3 < nt; t += 2) {
            steady_dead(I0{}, t);
            steady_dead(I1{}, t + 1);
        }
        for (; t < nt; ++t) tail_dead(t, t + 1 < nt, t + 2 < nt);
    } else {
        int t = 0;
        if constexpr (MIX) {                                // fp16 K-tiles (nt16 even; >= 4 e4m3 K-tiles follow, so t + 3 < nt holds throughout)
            for (; t + 1 < nt16; t += 2) {
                steady(I0{}, t, std::false_type{});
                steady(I1{}, t + 1, std::false_type{});
            }
        }
        for (; t + 3 < nt; t += 2) {
            steady(I0{}, t, MXL{});
            steady(I1{}, t + 1, MXL{});
        }
        for (; t < nt; ++t) tail(t, t + 1 < nt, t + 2 < nt, MXL{});
    }
    if (wr == 0) barrier();                                 // G1 spent its extra barrier up front
    if constexpr (STAMP) tile_ts[2] = __builtin_amdgcn_s_memtime();
    // The epilogue reads ITS parameters (output / residual / bias pointers, leading dimensions, the row map, scales) through the laundered
    // kernarg pointer, after the K loop: held in SGPRs from kernel entry they are live across the K loop, and one uniform value too
    // many there makes hipcc keep the two buffer descriptors in VGPRs -- a v_readfirstlane waterfall loop around every LDS-DMA load of the
    // steady state (seen: 16 per K-tile pair, the whole step 90.8 -> 93.9 ms, after an epilogue change added a handful of uniform values).
    GemmParams pe = *(const GemmParams*)kernarg_params();
    if (second) { pe.bias = pe.bias2; pe.c_off = pe.c_off2; }
    const bool vec_ok = (pe.N % 4 == 0) && (pe.ldc % 4 == 0) && (pe.resid == nullptr || pe.ldr % 4 == 0) &&
                        ((uintptr_t)pe.C % (4 * sizeof(OutT)) == 0) && ((uintptr_t)pe.bias % 16 == 0) && ((uintptr_t)pe.resid % 16 == 0);
    if (!dead || MAX32) gemm_epilogue<T, OutT, ACT, MAX32, TM, TN>(pe, acc, m0, n0, wr, wc, r32, half, vec_ok, smem, wave);
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the epilogue's stores have left the wave
        tile_ts[3] = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0) {
            uint64_t* o = reinterpret_cast<uint64_t*>(const_cast<float*>(kq->resid)) + (wave >> 2) * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) o[i] = ts[i];
        }
        // whole-tile timeline of a third-round workgroup (steady state): slots 32.. of the resid buffer
        if ((int)blockIdx.x == kq->nwg0 && lane == 0 && (wave & 3) == 0) {      // nwg0 = the workgroup to time (STAMP build)
            uint64_t* o = reinterpret_cast<uint64_t*>(const_cast<float*>(kq->resid)) + 32 + (wave >> 2) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = tile_ts[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) o[8 + i] = pro_ts[i];
        }
    }
    // next tile of this workgroup: its first LDS-DMA loads overwrite the strips the epilogue transposed through -- every wave has to be done
    // reading them.  The epilogue's stores stay in flight: loads return in order among themselves, so a counted vmcnt wait that still sees
    // older stores waits longer, never shorter.
    if (vb0 + (int)gridDim.x < total_tiles) barrier();
    }
}

static int ilog2_exact(int v) {
    if (v <= 0) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return ((1 << s) == v) ? s : -2;
}

// SPRC_GEMM_TILE: 0 = automatic (default), 2 = 128x128, 4 = 256x256
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? (int)strtol(e, nullptr, 0) : dflt;
}
constexpr int MAX_DEVICES = 64;
static int current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev >= 0 && dev < MAX_DEVICES ? dev : 0;
}
static int num_cus(hipStream_t st) { return stream_cus(st); }     // the stream's CU partition, or the device (core.hip)
// hipFuncSetAttribute applies to the function ON THE CURRENT DEVICE: opt in once per (kernel instantiation, device)
template <typename K>
static void optin_lds(K kern, int bytes, bool (&done)[MAX_DEVICES]) {
    const int dev = current_device();
    if (!done[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done[dev] = true;
    }
}

// one helper stream + fork / join events per device (streams and events belong to the device they were created on)
struct SideStream { hipStream_t st; hipEvent_t fork, join; };
static SideStream* side_stream() {
    static const int on = env_int("SPRC_GEMM_SIDE", 0);      // off: measured 89.7 -> 93.0 ms per bench step WITH it (see the call site)
    if (!on) return nullptr;
    static SideStream pool[MAX_DEVICES] = {};
    static int state[MAX_DEVICES] = {0};                // 0 = not created yet, 1 = ready, -1 = creation failed
    const int dev = current_device();
    if (state[dev] == 0) {
        SideStream& h = pool[dev];
        const bool ok = hipStreamCreateWithFlags(&h.st, hipStreamNonBlocking) == hipSuccess &&
                        hipEventCreateWithFlags(&h.fork, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&h.join, hipEventDisableTiming) == hipSuccess;
        state[dev] = ok ? 1 : -1;
    }
    return state[dev] == 1 ? &pool[dev] : nullptr;
}

template <typename T, typename OutT, int ACT, bool MAX32, int WM, int WN, int TM, int TN, int RESIDENT, int NS = 2, bool MIX = false>
static int launch_cfg(GemmParams p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, LDS = NS * (BM + BN) * 128;
    // (A persistent variant -- grid = CUs x residency with the next tile's first K-tile prefetched before the epilogue --
    // measured equal to one workgroup per tile on MI355X while costing ~45 VGPRs: all tiles take the same time, so the
    // CUs stay in lockstep and the output-write bursts still coincide.  Removed.)
    auto kern = gemm_kernel<T, OutT, ACT, MAX32, WM, WN, TM, TN, NS, MIX>;
    static bool attr_set[MAX_DEVICES] = {false};
    optin_lds(kern, LDS, attr_set);
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    static const int order = env_int("SPRC_GEMM_ORDER", 8 << 8);     // 8-m grouped order (mode 0, GROUP_M 8) is better for the K-heavy 128x128 GEMMs
    p.order = order;
    const int nwg = p.tiles_m * p.tiles_n;
    p.nwg0 = nwg;
    hipLaunchKernelGGL(kern, dim3(nwg * (p.ksplit > 1 ? p.ksplit : 1) * (p.dual ? 2 : 1)), dim3(64 * WM * WN), LDS, st, p);
    SPRC_CHECK_LAUNCH("sprc_gemm");
    return SPRC_OK;
}

template <typename T, typename OutT, int ACT, bool MAX32, bool MIX = false>
static int launch_anti(GemmParams p, hipStream_t st) {
    constexpr int LDS = 2 * 512 * 128;
    auto kern = gemm_anti_kernel<T, OutT, ACT, MAX32, false, MIX>;
    if constexpr (std::is_same<T, bf16_t>::value && std::is_same<OutT, bf16_t>::value && ACT == SPRC_ACT_NONE && !MAX32) {
        if ((p.debug & 64) && p.resid != nullptr) {         // phase-timestamp build (tools/gemm_stamp.py)
            auto sk = gemm_anti_kernel<T, OutT, ACT, MAX32, true>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            p.tiles_m = (p.M + 255) / 256;
            p.tiles_n = (p.N + 255) / 256;
            p.order = 4;
            p.ksplit = env_int("SPRC_GEMM_STAMP_MASK", 0xfff);
            p.nwg0 = env_int("SPRC_GEMM_STAMP_WG", 2 * 256 + 40);        // default: a third-round workgroup (steady state)
            hipLaunchKernelGGL(sk, dim3(p.tiles_m * p.tiles_n), dim3(512), LDS, st, p);
            SPRC_CHECK_LAUNCH("sprc_gemm(anti, stamp)");
            return SPRC_OK;
        }
    }
    static bool attr_set[MAX_DEVICES] = {false};
    optin_lds(kern, LDS, attr_set);
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    // Tile order: groups of GROUP_M = 4 m-tiles walking all their n-tiles (mode 0; 4 beats 8 by another 0.65-0.9 % of the step, 2 ties, 1 / 6 / 16 / 32 lose:
    // profiles/r06_order_ab.txt) -- since the 16-bit outputs are stored non-temporally this beats the W-resident
    // groups of 4 n-tiles (mode 4) that round 2 chose: same box, round-robin, 87.31 -> 86.62 ms per pipelined bench step (and 91.8 -> 91.1 on the slowest
    // box seen); in a traced single-stream step proj + fc2 go 327 -> 317 us per launch, fc1 524 -> 532, qkv stays (profiles/r06_order_ab.txt).  Mode 4 keeps
    // the one shape it was found on: N = 9216 (36 n-tiles, the cross-attention K|V projection: +8 %).  SPRC_GEMM_ORDER forces one mode everywhere (A/B).
    static const int order = env_int("SPRC_GEMM_ORDER", -1);
    p.order = order >= 0 ? order : (p.tiles_n >= 32 ? 4 : 0);
    p.nwg0 = p.tiles_m * p.tiles_n;
    // one workgroup per CU that walks its tiles (SPRC_GEMM_PERSIST=0: one workgroup per tile, the A/B switch).  Same box, pipelined bench step:
    // 90.13 / 90.24 ms (the kernel without the tile loop) -> 90.48 / 90.38 (this kernel, one workgroup per tile) -> 89.23 / 89.40 (persistent);
    // the class time of the serial, instrumented steps does not move (79.5 ms): what is saved are workgroup hand-overs while the Q-Former's
    // side-stream kernels compete for CUs (profiles/r05_persist_ab.txt)
    static const int persist = env_int("SPRC_GEMM_PERSIST", 1);
    const int total = p.nwg0 * (p.dual ? 2 : 1), ncu = num_cus(st) & ~7;
    hipLaunchKernelGGL(kern, dim3(persist && ncu > 0 && total > ncu ? ncu : total), dim3(512), LDS, st, p);
    SPRC_CHECK_LAUNCH("sprc_gemm(anti)");
    return SPRC_OK;
}

// duo kernel (gemm_duo.hpp; its instantiations live in gemm_duo.hip): 16-bit operands, out kind 0 = the operands' 16-bit type, 1 = fp32,
// 2 = SPRC_F16X3 split rows (fp16 operands).  SPRC_EUNSUPPORTED: no such instantiation -- the caller falls back to the other kernels.
int gemm_duo_launch(bool f16, int out_kind, int act, const GemmParams& p, hipStream_t st);

// The kernels address a tile with 32-bit offsets from its first row: the rows of one (<= 256-row) tile must span < 4 GiB.
static bool fits_u32(const GemmParams& p) {
    const int64_t span_a = p.a_shift < 0 ? 256 : (int64_t)((256 >> p.a_shift) + 2) * p.a_stride;
    const int64_t kbytes = (int64_t)p.K * 4;
    return span_a * p.lda_b + kbytes < ((int64_t)1 << 32) && 256 * p.ldw_b + kbytes < ((int64_t)1 << 32) &&
           p.lda_b < (1 << 24) && p.ldw_b < (1 << 24);          // the 256 x 256 kernel forms row offsets with 24-bit multiplies
}

// The vector epilogue forms a row's offset from the tile's first row with 24-bit multiplies in 32 bits (gemm_epilogue: mrow / __umul24): ldc and ldr
// < 2^24; with a C row map (logical row r -> (r >> shift) * stride + (r & mask) + offset) the group index M >> shift and the stride < 2^23 (signed
// __mul24), the map MONOTONE (stride >= rows per group: a tile's mapped rows then lie at or above its first mapped row -- the delta is an
// unsigned 24-bit operand) and that delta -- at most (256 >> shift) + 2 groups -- < 2^24; and the rows of one (<= 256-row) tile within 4 GiB of
// its first row in C and in the residual.  Anything else: SPRC_EUNSUPPORTED (the 64-bit path these checks replaced served any map; every
// map the library itself builds -- the Q-Former's query / text halves, the CLS row -- passes).
static bool epi_fits_u32(const GemmParams& p) {
    const int64_t span = p.c_shift < 0 ? 256 : (int64_t)((256 >> p.c_shift) + 2) * p.c_stride;
    const bool map_ok = p.c_shift < 0 || (p.c_stride >= (1 << p.c_shift) && p.c_stride < (1 << 23) && (p.M >> p.c_shift) < (1 << 23) && span < (1 << 24));
    return p.ldc < (1 << 24) && p.ldr < (1 << 24) && map_ok &&
           span * (p.ldc > p.ldr ? p.ldc : p.ldr) * 4 + ((int64_t)p.N + 256) * 4 < ((int64_t)1 << 32);
}

template <typename T, typename OutT, int ACT, bool MAX32, bool MIX = false>
static int launch(const GemmParams& p, hipStream_t st) {
    static const int forced = env_int("SPRC_GEMM_TILE", 0);
    if (!MAX32 && !epi_fits_u32(p)) {
        set_error("sprc_gemm: ldc / ldr / C row map outside the epilogue's 32-bit tile offsets (ldc=%lld ldr=%lld; cmap: %d rows per group, stride %d -- "
                  "the stride must be >= the rows per group and < 2^23, M / rows_per_group < 2^23)", (long long)p.ldc, (long long)p.ldr,
                  p.c_shift < 0 ? 0 : 1 << p.c_shift, p.c_stride);
        return SPRC_EUNSUPPORTED;
    }
    const int keff = MIX ? p.K + p.k8 / 2 : p.K;                // reduction length in units of 16-bit elements (time ~ bytes of a row)
    int cfg = forced;
    if constexpr (sizeof(T) == 2 && !MAX32 && !MIX && !std::is_same<OutT, fp8_t>::value) {
        static const int duo = env_int("SPRC_GEMM_DUO", 0);       // 1: every eligible product on the duo kernel (A/B switch)
        if (duo == 1 && p.ksplit <= 1 && p.K % 32 == 0) {
            constexpr int ok = std::is_same<OutT, float>::value ? 1 : std::is_same<OutT, f16x3_t>::value ? 2 : 0;
            const int rc = gemm_duo_launch(std::is_same<T, f16_t>::value, ok, ACT, p, st);
            if (rc != SPRC_EUNSUPPORTED) return rc;
        }
    }
    if constexpr (sizeof(T) <= 2) {
        if (cfg == 0) {
            // Cost model in units of one 256x256xK tile on a CU (measured on MI355X, tools/gemm_shapes.py):
            //   B  256x256 anti-phase kernel, one WG per CU:            rounds x 1
            //   A  128x128 kernel, two co-resident WGs per CU:           full rounds of 2 x CUs tiles x 0.7, a last round of
            //                                                            <= CUs tiles (one WG per CU, running alone) 0.42
            //   C  B on the first Mm rows (Mm a multiple of 256) + A on the M - Mm remaining ones, + 0.08 for the extra launch
            // C matters when a partial round of 256x256 tiles is nearly empty: the ViT GEMMs have M = 128 x 257 = 128.5 panels,
            // so N = 1408 is 774 tiles = 3.02 rounds (fc2: 732 us whole, 562 + 77 us peeled); the Q-Former Q|K|V product of
            // 233 fused queries is 531 tiles = 2.07 rounds (peel three panels: 504 tiles + 90 small ones).
            const int ncu = num_cus(st);
            const int64_t tn256 = (p.N + 255) / 256, tn128 = (p.N + 127) / 128;
            // a short reduction with an fp32 + residual epilogue spends as long writing out (an HBM burst no other workgroup
            // on the CU can hide) as in its K loop: +35 % per round (14912 x 768 x 768: 46 us on 256x256, 40 on 128x128;
            // 32896 x 1024 x 1024: 131 us peeled, 118 on 128x128)
            const double f256 = (keff <= 1024 && sizeof(OutT) == 4 && p.resid != nullptr) ? 1.35 : 1.0;
            const int64_t mult = p.dual ? 2 : 1;                 // a paired launch carries two products
            auto cost256 = [&](int m) { return f256 * (double)((mult * ((int64_t)(m + 255) / 256) * tn256 + ncu - 1) / ncu); };
            auto cost128 = [&](int m) {
                const int64_t t = mult * ((m + 127) / 128) * tn128, full = t / (2 * ncu), last = t % (2 * ncu);
                return 0.7 * (double)full + (last == 0 ? 0.0 : last <= ncu ? 0.42 : 0.7);
            };
            const double cA = cost128(p.M), cB = cost256(p.M);
            static const int peel = env_int("SPRC_GEMM_PEEL", 1);
            const bool can_peel = peel && !MAX32 && !p.dual && p.M > 256 && p.a_shift < 0 && p.c_shift < 0;
            double cC = 1e30;
            int Mm = 0;
            if (can_peel) {
                for (int j = 0; j <= 12; ++j) {                        // peel the partial panel plus j whole ones
                    const int m = (p.M / 256 - j) * 256;
                    if (m <= 0 || m == p.M) continue;
                    const double c = cost256(m) + cost128(p.M - m) + 0.08;
                    if (c < cC - 1e-9) { cC = c; Mm = m; }
                }
            }
            const int rem = p.M - Mm;
            // peel when it saves at least a quarter of a tile-round (the ViT fc1: 3096 tiles = 12.09 rounds -> 12 + 0.5)
            if (cC + 0.25 < (cA < cB ? cA : cB)) {
                GemmParams pm = p, pt = p;
                pm.M = Mm;
                pt.M = rem;
                pt.A = p.A + (int64_t)Mm * p.lda_b;
                pt.C = reinterpret_cast<char*>(p.C) + (int64_t)Mm * p.ldc * (MAX32 ? 4 : (int64_t)sizeof(OutT));
                if (p.resid != nullptr) pt.resid = p.resid + (int64_t)Mm * p.ldr;
                // SPRC_GEMM_SIDE=1 (A/B switch, OFF by default): the remainder rows (disjoint from the main launch's) go to a library-owned
                // side stream, forked before and joined after the main launch, so that their handful of latency-bound workgroups
                // (11-16 us per launch, 1.8 ms of a bench step when run behind the main kernel) could start as soon as CUs of the main
                // kernel's last round fall idle.  Measured on MI355X: the step got SLOWER, 89.7 -> 93.0 ms (same box, twice) -- the two
                // extra event pairs per product and the remainder workgroups taking CUs from the main kernel's first round cost more
                // than the 1.8 ms they could hide.
                SideStream* ss = side_stream();
                hipStream_t sr = st;
                if (ss != nullptr) {
                    (void)hipEventRecord(ss->fork, st);
                    (void)hipStreamWaitEvent(ss->st, ss->fork, 0);
                    sr = ss->st;
                }
                const int rc = launch_anti<T, OutT, ACT, MAX32, MIX>(pm, st);
                if (rc != SPRC_OK) return rc;
                const int rr = [&]() -> int {
                hipStream_t st = sr;                 // (shadows the caller's stream inside the remainder launches)
                // remainder rows: a long reduction on a handful of workgroups is latency-bound (11 WGs x 96 K-tiles = 77 us
                // for the ViT fc2) -> split K over 8 workgroups per tile into caller scratch and reduce in a fixed order
                constexpr int S = 8;
                if (rem <= 128 && keff >= 4096 && p.N % 4 == 0 && p.ldc % 4 == 0 && p.scratch != nullptr &&
                    p.scratch_elems >= (int64_t)S * rem * p.N) {
                    GemmParams ps = pt;
                    ps.C = p.scratch; ps.ldc = p.N; ps.bias = nullptr; ps.resid = nullptr; ps.ldr = 0; ps.w_scale = nullptr;
                    ps.ksplit = S; ps.split_stride = (int64_t)rem * p.N;
                    static const int ring = env_int("SPRC_GEMM_RING", 1);   // a handful of workgroups: ring of 4 stages (0: two stages, A/B)
                    const int rs = ring ? launch_cfg<T, float, SPRC_ACT_NONE, false, 2, 2, 2, 2, 1, 4, MIX>(ps, st)
                                        : launch_cfg<T, float, SPRC_ACT_NONE, false, 2, 2, 2, 2, 2, 2, MIX>(ps, st);
                    if (rs != SPRC_OK) return rs;
                    const int64_t n = (int64_t)rem * (p.N / 4);
                    hipLaunchKernelGGL((splitk_reduce_kernel<OutT, ACT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                                       p.scratch, S, ps.split_stride, rem, p.N, p.bias, pt.resid, p.ldr,
                                       reinterpret_cast<OutT*>(pt.C), p.ldc, p.w_scale, p.a_scale, p.out_scale);
                    SPRC_CHECK_LAUNCH("sprc_gemm(split-K reduce)");
                    return SPRC_OK;
                }
                if constexpr (sizeof(T) == 2 && !MAX32) {
                    static const int ring = env_int("SPRC_GEMM_RING", 1);
                    if (((rem + 127) / 128) * tn128 <= ncu)
                        return ring ? launch_cfg<T, OutT, ACT, MAX32, 2, 2, 1, 1, 2, 4, MIX>(pt, st) : launch_cfg<T, OutT, ACT, MAX32, 2, 2, 1, 1, 4, 2, MIX>(pt, st);
                }
                return launch_cfg<T, OutT, ACT, MAX32, 2, 2, 2, 2, 2, 2, MIX>(pt, st);
                }();
                if (ss != nullptr) {
                    (void)hipEventRecord(ss->join, sr);
                    (void)hipStreamWaitEvent(st, ss->join, 0);
                }
                return rr;
            }
            cfg = cB <= cA ? 4 : 2;
            // a 128x128 grid that cannot even give every CU one workgroup is latency-bound (12-48 K-tiles on a fraction of the
            // chip): 64x64 tiles (16 KB per LDS stage) spread it 4x wider.  Measured: 4096x768x768 17.6 -> 12.7 us,
            // 4096x768x3072 44.9 -> 37.6, the 128-row remainders of the ViT products 18.3 -> 11.4-12.2; grids above one
            // workgroup per CU (7456x768x768, 4096x2304x768) are equal or slower on 64x64 and keep 128x128.
            if (cfg == 2 && !MAX32 && sizeof(T) == 2 && mult * ((p.M + 127) / 128) * tn128 <= ncu) cfg = 1;
        }
        // 256x256 tile: anti-phase schedule by default (SPRC_GEMM_TILE=14 forces the lock-step kernel for A/B runs)
        if (cfg == 4 || cfg == 10) return launch_anti<T, OutT, ACT, MAX32, MIX>(p, st);
    } else {
        if (cfg == 0) cfg = (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256) >= 1024 ? 4 : 2;
    }
    if constexpr (!MIX) { if (cfg == 4 || cfg == 10 || cfg == 14) return launch_cfg<T, OutT, ACT, MAX32, 2, 4, 4, 2, 1>(p, st); }
    if constexpr (sizeof(T) == 2 && !MAX32) {
        if (cfg == 1) {                                     // 64 x 64 tile: small latency-bound products; ring of 4 stages (64 KB) or 2 (32 KB)
            // (the ring's 64 KB leave two workgroups per CU: only for grids that fit then -- 768 workgroups ran 5 % slower on it)
            static const int ring = env_int("SPRC_GEMM_RING", 1);
            const int64_t nwg64 = (p.dual ? 2 : 1) * (int64_t)((p.M + 63) / 64) * ((p.N + 63) / 64);
            return ring && nwg64 <= 2 * num_cus(st) ? launch_cfg<T, OutT, ACT, MAX32, 2, 2, 1, 1, 2, 4, MIX>(p, st)
                                                  : launch_cfg<T, OutT, ACT, MAX32, 2, 2, 1, 1, 4, 2, MIX>(p, st);
        }
    }
    return launch_cfg<T, OutT, ACT, MAX32, 2, 2, 2, 2, 2, 2, MIX>(p, st);
}

template <typename T>
static int dispatch(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st) {
    if constexpr (sizeof(T) != 1) { if (a->max32) return launch<T, float, SPRC_ACT_NONE, true>(p, st); }
    if constexpr (std::is_same<T, bf16_t>::value) {     // residual-branch delta (validated by the caller: plain epilogue)
        if (a->out_dtype == SPRC_F16) return launch<T, f16_t, SPRC_ACT_NONE, false>(p, st);
    } else if constexpr (sizeof(T) == 4) {
        if (a->out_dtype == SPRC_F16) { set_error("sprc_gemm: SPRC_F16 output needs bf16, fp16 or fp8 operands"); return SPRC_EUNSUPPORTED; }
    }
    if constexpr (sizeof(T) == 1) {                     // fp8 operands: the three epilogues of the fp8 ViT path
        if (a->out_dtype == SPRC_FP8) {
            switch (a->act) {
                case SPRC_ACT_NONE: return launch<T, fp8_t, SPRC_ACT_NONE, false>(p, st);
                case SPRC_ACT_GELU: return launch<T, fp8_t, SPRC_ACT_GELU, false>(p, st);
                case SPRC_ACT_QUICKGELU: return launch<T, fp8_t, SPRC_ACT_QUICKGELU, false>(p, st);
            }
        }
        if (a->act == SPRC_ACT_NONE && a->out_dtype == SPRC_BF16) return launch<T, bf16_t, SPRC_ACT_NONE, false>(p, st);
        if (a->act == SPRC_ACT_NONE && a->out_dtype == SPRC_F16) return launch<T, f16_t, SPRC_ACT_NONE, false>(p, st);
        if (a->act == SPRC_ACT_NONE && a->out_dtype == SPRC_F32) return launch<T, float, SPRC_ACT_NONE, false>(p, st);
        set_error("sprc_gemm(fp8): unsupported epilogue (act %d, out_dtype %d)", a->act, a->out_dtype);
        return SPRC_EUNSUPPORTED;
    } else {
        if (a->out_dtype == SPRC_FP8) { set_error("sprc_gemm: SPRC_FP8 output needs fp8 operands"); return SPRC_EUNSUPPORTED; }
    }
    // 16-bit output type of this operand type: fp16 operands write fp16, everything else bf16
    constexpr bool H = std::is_same<T, f16_t>::value;
    typedef typename std::conditional<H, f16_t, bf16_t>::type O16;
    if constexpr (H) {                                  // split-precision output of the fp16 engine's Q-Former (validated by the caller)
        if (a->out_dtype == SPRC_F16X3)
            return a->act == SPRC_ACT_GELU ? launch<T, f16x3_t, SPRC_ACT_GELU, false>(p, st) : launch<T, f16x3_t, SPRC_ACT_NONE, false>(p, st);
    }
    if (a->out_dtype != SPRC_F32 && a->out_dtype != (H ? SPRC_F16 : SPRC_BF16)) {
        set_error("sprc_gemm: out_dtype %d does not go with operand dtype %d", a->out_dtype, a->dtype);
        return SPRC_EUNSUPPORTED;
    }
    const bool o16 = a->out_dtype != SPRC_F32;
    switch (a->act) {
        case SPRC_ACT_NONE:
            return o16 ? launch<T, O16, SPRC_ACT_NONE, false>(p, st) : launch<T, float, SPRC_ACT_NONE, false>(p, st);
        case SPRC_ACT_GELU:
            return o16 ? launch<T, O16, SPRC_ACT_GELU, false>(p, st) : launch<T, float, SPRC_ACT_GELU, false>(p, st);
        case SPRC_ACT_QUICKGELU:
            return o16 ? launch<T, O16, SPRC_ACT_QUICKGELU, false>(p, st)
                       : launch<T, float, SPRC_ACT_QUICKGELU, false>(p, st);
    }
    set_error("sprc_gemm: unknown activation %d", a->act);
    return SPRC_EINVAL;
}


// split-precision products (fp16 + e4m3 segments): the epilogues the fp16 engine's Q-Former and patch embedding use
template <typename T>                       // (a template so that only gemm_f16e.hip instantiates the kernels)
static int dispatch_mix(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st) {
    static_assert(std::is_same<T, f16_t>::value, "split-precision products have an fp16 main segment");
    if (a->act != SPRC_ACT_NONE && a->act != SPRC_ACT_GELU) { set_error("sprc_gemm(k8): activation %d is not built for split-precision products", a->act); return SPRC_EUNSUPPORTED; }
    const bool gelu = a->act == SPRC_ACT_GELU;
    switch (a->out_dtype) {
        case SPRC_F16X3: return gelu ? launch<f16_t, f16x3_t, SPRC_ACT_GELU, false, true>(p, st) : launch<f16_t, f16x3_t, SPRC_ACT_NONE, false, true>(p, st);
        case SPRC_F16: return gelu ? launch<f16_t, f16_t, SPRC_ACT_GELU, false, true>(p, st) : launch<f16_t, f16_t, SPRC_ACT_NONE, false, true>(p, st);
        case SPRC_F32: return gelu ? launch<f16_t, float, SPRC_ACT_GELU, false, true>(p, st) : launch<f16_t, float, SPRC_ACT_NONE, false, true>(p, st);
    }
    set_error("sprc_gemm(k8): out_dtype %d does not go with a split-precision product", a->out_dtype);
    return SPRC_EUNSUPPORTED;
}

// per-operand-type dispatchers, one translation unit each
int gemm_dispatch_f16e(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st);
int gemm_dispatch_bf16(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st);
int gemm_dispatch_f16(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st);
int gemm_dispatch_f32(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st);
int gemm_dispatch_fp8(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st);

}  // namespace sprc
