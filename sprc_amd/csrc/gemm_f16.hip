// gemm_f16.hip -- fp16-operand instantiations of the GEMM kernels (v_mfma_f32_32x32x16_f16; kernels: gemm_impl.hpp).
// fp16 is the reference's own GPU precision (fp16 autocast, blip2.py:36-44; eva_vit.py:410-425): same MFMA rate and LDS
// layout as bf16 with 11-bit significands, i.e. 8x finer operand rounding.
#include "gemm_impl.hpp"

namespace sprc {
int gemm_dispatch_f16(const sprc_gemm_args* a, const GemmParams& p, hipStream_t st) { return dispatch<f16_t>(a, p, st); }
}  // namespace sprc
