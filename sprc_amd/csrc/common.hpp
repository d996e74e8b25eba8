// common.hpp -- shared device/host helpers for the gfx950 kernels of libsprc_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/sprc.h"

namespace sprc {

// ---- error plumbing --------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define SPRC_REQUIRE(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            ::sprc::set_error(__VA_ARGS__);             \
            return SPRC_EINVAL;                         \
        }                                               \
    } while (0)
#define SPRC_CHECK_LAUNCH(name)                                                     \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            ::sprc::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return SPRC_ELAUNCH;                                                    \
        }                                                                           \
    } while (0)

// ---- optional per-launch event timing (core.hip) ---------------------------------------------------
struct ProfScope {
    int slot;
    hipStream_t st;
    ProfScope(int cls, hipStream_t stream, double flops, double bytes, double exec_flops = -1.0);   // exec < 0: == flops
    ~ProfScope();
};

// CUs a launch on `st` can use: the count registered by sprc_stream_create_partition for a CU-masked stream, else the device's (core.hip)
int stream_cus(hipStream_t st);

// ---- types -----------------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {   // round-to-nearest-even, NaN-preserving
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {      // v_cvt_pk_bf16_f32 (RNE) on gfx950
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {       // RNE conversions, packed low | high
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
    const f16x2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
// hi / lo halves of the split-precision layout (SPRC_F16X3): hi = fp16(x), lo = fp16(x - hi).  x is PINNED first (empty asm):
// under hipcc's default -ffp-contract=fast the multiply that produced x is otherwise folded into each consumer separately --
// v_cvt_pk_f16_f32 of the rounded fp32 product for the stored hi, v_fma_mixlo_f16 of the EXACT product for the hi that lo is
// taken from -- and where the fp32 value is a tie of the fp16 grid the two disagree by one ulp (seen: 37 of 921600 outputs of a
// GELU epilogue off by 2^-10).
__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    asm("" : "+v"(x));
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}
// ---- SPRC_F16X3: the split-precision row [K fp16: hi = fp16(x) | K e4m3: (x - hi) 2^12 | K e4m3: x]  (sprc.h) -----------------------
// saturating fp32 -> 4 x e4m3fn (v_cvt_pk_fp8_f32: RNE; inputs clamped to +-448 first, the format has no infinity)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f); b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f); d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, lo, true);
}
constexpr float SPLIT_LO_SCALE = 4096.0f;        // 2^12: an fp16 rounding residual (<= 2^-11 |x|) lands in e4m3's normal range for |x| in [2^-6, 2^8]
// hi = fp16(x) and the EXACT residual x - hi (fp32); x is pinned first for the reason given at split_f16
__device__ __forceinline__ void split_f16_res(float x, _Float16& hi, float& res) {
    asm("" : "+v"(x));
    hi = (_Float16)x;
    res = x - (float)hi;
}
// four consecutive columns [col, col + 4) of a split row of logical width K; `row` = first byte of the row; col % 4 == 0
__device__ __forceinline__ void store_split4(char* row, int K, int col, float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(4))) _Float16 half4;
    _Float16 h0, h1, h2, h3;
    float r0, r1, r2, r3;
    split_f16_res(a, h0, r0); split_f16_res(b, h1, r1); split_f16_res(c, h2, r2); split_f16_res(d, h3, r3);
    *reinterpret_cast<half4*>(row + 2 * col) = half4{h0, h1, h2, h3};
    *reinterpret_cast<uint32_t*>(row + 2 * K + col) = pack_fp8x4(r0 * SPLIT_LO_SCALE, r1 * SPLIT_LO_SCALE, r2 * SPLIT_LO_SCALE, r3 * SPLIT_LO_SCALE);
    *reinterpret_cast<uint32_t*>(row + 3 * K + col) = pack_fp8x4(a, b, c, d);
}
// eight consecutive columns [col, col + 8) of a split row; col % 8 == 0, K % 8 == 0: 16 B of hi + 8 B + 8 B of e4m3
__device__ __forceinline__ void store_split8(char* row, int K, int col, const f32x4& a, const f32x4& b) {
    _Float16 h[8];
    float r[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { split_f16_res(a[e], h[e], r[e]); split_f16_res(b[e], h[4 + e], r[4 + e]); }
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    auto pk = [&](int i) { const h2 v = {h[i], h[i + 1]}; return __builtin_bit_cast(uint32_t, v); };
    *reinterpret_cast<u32x4*>(row + 2 * col) = u32x4{pk(0), pk(2), pk(4), pk(6)};
    *reinterpret_cast<u32x2*>(row + 2 * K + col) = u32x2{pack_fp8x4(r[0] * SPLIT_LO_SCALE, r[1] * SPLIT_LO_SCALE, r[2] * SPLIT_LO_SCALE, r[3] * SPLIT_LO_SCALE),
                                                          pack_fp8x4(r[4] * SPLIT_LO_SCALE, r[5] * SPLIT_LO_SCALE, r[6] * SPLIT_LO_SCALE, r[7] * SPLIT_LO_SCALE)};
    *reinterpret_cast<u32x2*>(row + 3 * K + col) = u32x2{pack_fp8x4(a[0], a[1], a[2], a[3]), pack_fp8x4(b[0], b[1], b[2], b[3])};
}
__device__ __forceinline__ void store_split1(char* row, int K, int col, float x) {
    _Float16 h;
    float r;
    split_f16_res(x, h, r);
    *reinterpret_cast<_Float16*>(row + 2 * col) = h;
    *reinterpret_cast<uint8_t*>(row + 2 * K + col) = (uint8_t)(pack_fp8x4(r * SPLIT_LO_SCALE, 0.f, 0.f, 0.f) & 0xffu);
    *reinterpret_cast<uint8_t*>(row + 3 * K + col) = (uint8_t)(pack_fp8x4(x, 0.f, 0.f, 0.f) & 0xffu);
}

typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
__device__ __forceinline__ void split_f16x4(float a, float b, float c, float d, f16x4& hi, f16x4& lo) {
    _Float16 h0, h1, h2, h3, l0, l1, l2, l3;
    split_f16(a, h0, l0); split_f16(b, h1, l1); split_f16(c, h2, l2); split_f16(d, h3, l3);
    hi = f16x4{h0, h1, h2, h3};
    lo = f16x4{l0, l1, l2, l3};
}

// two 16-bit operand values of the engine's compute dtype (F16: IEEE half, the reference's GPU autocast precision; else bf16)
template <bool F16>
__device__ __forceinline__ uint32_t pack16x2(float lo, float hi) {
    if constexpr (F16) return pack_f16x2(lo, hi);
    else return pack_bf16x2(lo, hi);
}

__host__ __device__ __forceinline__ int64_t map_row(const sprc_rowmap& m, int64_t r) {
    if (m.rows_per_group <= 0) return r;
    return (r / m.rows_per_group) * (int64_t)m.group_stride + (r % m.rows_per_group) + m.group_offset;
}

// counter-based dropout mask (sprc.h: sprc_dropout_f32): element `idx` of site `site` is KEPT iff the high word of the hash >= thresh
__host__ __device__ __forceinline__ bool drop_keep(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thresh) {
    uint64_t z = seed + (uint64_t)site * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32) >= thresh;
}
static inline uint32_t drop_thresh(float p) { return (uint32_t)((double)p * 4294967296.0); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// the 16-bit / fp8 epilogues' form: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of an IEEE division (~10 VALU ops): 5 ops per value.  The
// ViT-L fc1 products (K = 1024: 8 .. 16 K-tiles) are epilogue-bound -- the e4m3 one ran at 1180 TFLOP/s against fc2's 2070 on the same flops.
__device__ __forceinline__ float quick_gelu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (-1.702f * 1.44269504088896340736f)));
}

static inline bool is16(int dt) { return dt == SPRC_BF16 || dt == SPRC_F16; }   // 16-bit MFMA operand engines
static inline size_t dtype_size(int dt) { return dt == SPRC_FP8 ? 1 : (dt == SPRC_BF16 || dt == SPRC_F16 || dt == SPRC_F16X3) ? 2 : 4; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace sprc
