// gemm_f32.hip -- exact-fp32 MFMA instantiations of the GEMM (the parity engine); kernels: gemm_impl.hpp
#include "gemm_impl.hpp"

namespace sprc {
int gemm_dispatch_f32(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st) { return dispatch<float>(a, p, st); }
}  // namespace sprc
