// kernels_gemm_sel.hip -- one instantiation family of the nn.Linear GEMM (see gemm_launch.h)
#include "gemm_launch.h"

namespace sttran {

hipError_t gemm_linear_sel(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                           const EpiLinear& epi, GemmPlan plan, float* slab) {
  return gemm_generic<EpiScalar4<EpiLinear>, B_KMAJOR>(s, A, B, M, N, K, EpiScalar4<EpiLinear>{epi}, plan, slab);
}

}  // namespace sttran
