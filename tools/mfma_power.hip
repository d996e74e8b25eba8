// MFMA-only microbenchmark (round 6): one instruction kind in a register-resident loop -- no LDS, no memory -- on every SIMD of the chip, for
// tools/mfma_power.py to read socket power and shader clock under it: what a flop costs per instruction kind when nothing else runs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_power tools/mfma_power.hip ;  tools/bin/mfma_power <kind> <seconds>
// kinds: 0 v_mfma_f32_32x32x16_f16   1 v_mfma_f32_32x32x16_bf16   2 v_mfma_f32_16x16x32_f16   3 v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)
//        4 the same in the fp6 (e2m3) format   5 v_mfma_f32_32x32x2_f32 (the parity engine's)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int KIND>
__global__ __launch_bounds__(256) void mfma_loop(float* out, const uint32_t* seed, int iters) {
    // operands: pseudo-random bit patterns of NORMAL magnitude (random mantissas, exponents near 1.0), different per lane and per accumulator
    uint32_t s = seed[threadIdx.x & 63] ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    f32x16 acc[8];
    f32x4 acc4[8];
    for (int a = 0; a < 8; ++a) { for (int r = 0; r < 16; ++r) acc[a][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[a][r] = 0.f; }
    uint32_t ra[8], rb[8];
    for (int i = 0; i < 8; ++i) {
        if (KIND <= 2) { ra[i] = (rnd() & 0x83ff83ffu) | 0x38003800u; rb[i] = (rnd() & 0x83ff83ffu) | 0x38003800u; }   // fp16 / bf16 halves in [0.5, 1)
        else { ra[i] = (rnd() & 0x87878787u) | 0x30303030u; rb[i] = (rnd() & 0x87878787u) | 0x30303030u; }             // e4m3 bytes around 0.5 .. 1
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if constexpr (KIND == 0) {
                const f16x8 A = __builtin_bit_cast(f16x8, *(const uint4*)&ra[(a & 1) * 4]), B = __builtin_bit_cast(f16x8, *(const uint4*)&rb[(a & 1) * 4]);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[a], 0, 0, 0);
            } else if constexpr (KIND == 1) {
                const bf16x8 A = __builtin_bit_cast(bf16x8, *(const uint4*)&ra[(a & 1) * 4]), B = __builtin_bit_cast(bf16x8, *(const uint4*)&rb[(a & 1) * 4]);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[a], 0, 0, 0);
            } else if constexpr (KIND == 2) {
                const f16x8 A = __builtin_bit_cast(f16x8, *(const uint4*)&ra[(a & 1) * 4]), B = __builtin_bit_cast(f16x8, *(const uint4*)&rb[(a & 1) * 4]);
                acc4[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc4[a], 0, 0, 0);
            } else if constexpr (KIND == 3 || KIND == 4) {
                const i32x8 A = {(int)ra[0], (int)ra[1], (int)ra[2], (int)ra[3], (int)ra[4], (int)ra[5], (int)ra[6], (int)ra[7]};
                const i32x8 B = {(int)rb[0], (int)rb[1], (int)rb[2], (int)rb[3], (int)rb[4], (int)rb[5], (int)rb[6], (int)rb[7]};
                constexpr int F = KIND == 4 ? 2 : 0;
                acc[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[a], F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            } else {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float((ra[a] & 0x007fffffu) | 0x3f000000u), __uint_as_float((rb[a] & 0x007fffffu) | 0x3f000000u), acc[a], 0, 0, 0);
            }
        }
    }
    float t = 0.f;
    for (int a = 0; a < 8; ++a) { for (int r = 0; r < 16; ++r) t += acc[a][r]; for (int r = 0; r < 4; ++r) t += acc4[a][r]; }
    if (t == 12345.678f) out[threadIdx.x] = t;
}

// 16x16x32 has a 4-register accumulator: its own kernel (with both accumulator arrays alive the compiler shuffled them through v_accvgpr copies)
__global__ __launch_bounds__(256) void mfma_loop_16(float* out, const uint32_t* seed, int iters) {
    uint32_t s = seed[threadIdx.x & 63] ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    uint32_t ra[8], rb[8];
    for (int i = 0; i < 8; ++i) { ra[i] = (rnd() & 0x83ff83ffu) | 0x38003800u; rb[i] = (rnd() & 0x83ff83ffu) | 0x38003800u; }
    const f16x8 A0 = __builtin_bit_cast(f16x8, *(const uint4*)&ra[0]), A1 = __builtin_bit_cast(f16x8, *(const uint4*)&ra[4]);
    const f16x8 B0 = __builtin_bit_cast(f16x8, *(const uint4*)&rb[0]), B1 = __builtin_bit_cast(f16x8, *(const uint4*)&rb[4]);
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, B0, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, B1, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, B0, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, B1, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, B0, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, B1, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, B0, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, B1, c7, 0, 0, 0);
    }
    const f32x4 t4 = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    const float t = t4[0] + t4[1] + t4[2] + t4[3];
    if (t == 12345.678f) out[threadIdx.x] = t;
}

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 8.0;
    float* out; uint32_t* seed;
    hipMalloc(&out, 4096); hipMalloc(&seed, 256);
    uint32_t h[64]; for (int i = 0; i < 64; ++i) h[i] = 0x9e3779b9u * (i + 1);
    hipMemcpy(seed, h, 256, hipMemcpyHostToDevice);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * 2, iters = 20000;          // 2 workgroups of 4 waves per CU: two waves per SIMD
    const double flop_per_mfma[6] = {2.0 * 32 * 32 * 16, 2.0 * 32 * 32 * 16, 2.0 * 16 * 16 * 32, 2.0 * 32 * 32 * 64, 2.0 * 32 * 32 * 64, 2.0 * 32 * 32 * 2};
    auto launch = [&]() {
        switch (kind) {
            case 0: hipLaunchKernelGGL(mfma_loop<0>, dim3(grid), dim3(256), 0, 0, out, seed, iters); break;
            case 1: hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, out, seed, iters); break;
            case 2: hipLaunchKernelGGL(mfma_loop_16, dim3(grid), dim3(256), 0, 0, out, seed, iters); break;
            case 3: hipLaunchKernelGGL(mfma_loop<3>, dim3(grid), dim3(256), 0, 0, out, seed, iters); break;
            case 4: hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, seed, iters); break;
            default: hipLaunchKernelGGL(mfma_loop<5>, dim3(grid), dim3(256), 0, 0, out, seed, iters); break;
        }
    };
    launch(); hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double dt = 0;
    do { launch(); hipDeviceSynchronize(); ++n; dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } while (dt < secs);
    const double flops = (double)n * grid * 4 /* waves */ * iters * 8 * flop_per_mfma[kind < 6 ? kind : 5];
    printf("kind %d: %.1f TFLOP/s over %.1f s\n", kind, flops / dt / 1e12, dt);
    return 0;
}
