// gemm_f16e.hip -- the split-precision instantiations of the GEMM kernels (fp16 K-tiles followed by e4m3 K-tiles on the MX-scaled MFMA:
// sprc.h SPRC_F16X3; kernels: gemm_impl.hpp).  Its own translation unit: the six epilogues x three tile shapes compile next to the
// plain fp16 ones instead of after them.
#include "gemm_impl.hpp"

namespace sprc {
int gemm_dispatch_f16e(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st) { return dispatch_mix<f16_t>(a, p, st); }
}  // namespace sprc
