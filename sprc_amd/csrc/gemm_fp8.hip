// gemm_fp8.hip -- e4m3fn-operand instantiations of the GEMM (BASELINE.json config C5); kernels: gemm_impl.hpp
#include "gemm_impl.hpp"

namespace sprc {
int gemm_dispatch_fp8(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st) { return dispatch<fp8_t>(a, p, st); }
}  // namespace sprc
