// Explicit instantiations of the three 256 x 256 GEMM kernels the ViT spends its time in, for tests/test_kloop_listing.py and
// tools/kloop_stat.py: a listing of this file takes half a minute, one of gemm_f16.hip (forty instantiations) three.
#include "../sprc_amd/csrc/gemm_impl.hpp"
namespace sprc {
template __global__ void gemm_anti_kernel<f16_t, f16_t, SPRC_ACT_NONE, false, false, false>(GemmParams);     // qkv
template __global__ void gemm_anti_kernel<f16_t, f16_t, SPRC_ACT_GELU, false, false, false>(GemmParams);     // fc1
template __global__ void gemm_anti_kernel<f16_t, float, SPRC_ACT_NONE, false, false, false>(GemmParams);     // proj / fc2
}  // namespace sprc
