// Probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read): which 16-bit elements does lane l receive when every lane of a
// 16-lane group supplies the address of one 8-byte piece of a [4 rows][16 columns] block?  hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const int* in, int* out) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int addr = in[threadIdx.x];
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)((__attribute__((address_space(3))) char*)lds + addr));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = r[e];
}
int main() {
    int h_in[64], h_out[256], *d_in, *d_out;
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, sizeof h_out);
    for (int pitch : {32, 64, 208}) {
        for (int l = 0; l < 64; ++l) { const int g = l >> 4, i = l & 15; h_in[l] = g * 2048 + (i / 4) * pitch + (i % 4) * 8; }
        hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out);
        hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            const int g = l >> 4, i = l & 15, want = (g * 2048 + j * pitch + i * 2) / 2;      // element [row j][col i] of the group's block
            if (h_out[l * 4 + j] != want) ++bad;
        }
        printf("row pitch %3d B: lane (group g, i) elem j == block[row j][col i] for %d of 256 values\n", pitch, 256 - bad);
        if (bad) { for (int l = 0; l < 20; ++l) printf("  lane %2d (addr %4d): %d %d %d %d\n", l, h_in[l], h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]); }
    }
    return 0;
}
