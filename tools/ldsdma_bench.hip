// Fill-path microbenchmark (tools only): how many bytes per clock one CU can pull from L2 / Infinity Cache into LDS
//   mode 0  buffer_load_dwordx4 ... lds   (the GEMM's staging instruction; 1 KiB per wave instruction)
//   mode 1  buffer_load_dwordx4 -> VGPR   (no LDS write)
//   mode 2  buffer_load_dwordx4 -> VGPR -> ds_write_b128
// 256 workgroups x 8 waves (one workgroup per CU, 128 KB of LDS like the GEMM); every wave keeps `DEPTH` loads in flight.
// window = bytes of source the whole grid cycles through: small (<= 2 MB) = L2 hits, large = Infinity Cache / HBM.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ldsdma_bench tools/ldsdma_bench.hip ; run: ldsdma_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lptr_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void fill_kernel(const char* src, uint32_t window, int iters, unsigned long long* cyc, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0xffffffff, 0x00020000);
    // every workgroup starts somewhere else in the window; one iteration = 4 loads per wave = 32 KB per workgroup
    uint32_t off = (uint32_t)(((uint64_t)blockIdx.x * 32768u * 7u) % window);
    const uint32_t lane_off = wave * 4096 + lane * 16;
    char* dst = smem + wave * 16384;
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int slot = (it & 3) * 4096;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + slot + j * 1024), 16, lane_off + j * 1024, (int)off, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
        } else {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + j * 1024, (int)off, 0);
            if constexpr (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(dst + slot + j * 1024 + lane * 16) = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc ^= v[j];
            }
        }
        off += 32768u;
        if (off >= window) off -= window;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (MODE != 0) {
        u32x4 r = *reinterpret_cast<u32x4*>(smem + (threadIdx.x * 16) % 65536);
        acc ^= r;
    }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[0] = acc[2] ^ acc[3];
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* src, uint32_t window, int iters, unsigned long long* dcyc, uint32_t* sink) {
    const int nwg = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(nwg), dim3(512), 131072, 0, src, window, iters, dcyc, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nwg);
        (void)hipMemcpy(h.data(), dcyc, nwg * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto c : h) mean += (double)c;
        mean /= nwg;
        const double bytes_wg = (double)iters * 32768.0;
        if (rep == 2)
            printf("%-34s window %8.2f MB  %6.2f B/clk/CU  (%.0f cycles per 64 KB)  chip %.2f TB/s  wall %.3f ms  clock ~%.2f GHz\n", name,
                   window / 1048576.0, bytes_wg / mean, mean / (bytes_wg / 65536.0), bytes_wg * nwg / (ms * 1e-3) / 1e12, ms, mean / (ms * 1e-3) / 1e9);
    }
}

int main() {
    const size_t cap = 256u << 20;
    char* src;
    unsigned long long* dcyc;
    uint32_t* sink;
    (void)hipMalloc(&src, cap);
    (void)hipMemset(src, 1, cap);
    (void)hipMalloc(&dcyc, 256 * 8);
    (void)hipMalloc(&sink, 64);
    const int iters = 2048;                                // 64 MB per workgroup
    for (uint32_t window : {1u << 20, 16u << 20, 128u << 20}) {
        run<0, 4>("lds-dma, 4+4 in flight", src, window, iters, dcyc, sink);
        run<0, 0>("lds-dma, drain every 4", src, window, iters, dcyc, sink);
        run<1, 0>("to VGPR (compiler waits)", src, window, iters, dcyc, sink);
        run<2, 0>("to VGPR + ds_write_b128", src, window, iters, dcyc, sink);
    }
    return 0;
}
