// kernels_gemm_vec.hip -- one instantiation family of the nn.Linear GEMM (see gemm_launch.h)
#include "gemm_launch.h"

namespace sttran {

hipError_t gemm_linear_vec(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                           const EpiLinear& epi, GemmPlan plan, float* slab) {
  return gemm_generic<EpiLinearV, B_KMAJOR_PAD>(s, A, B, M, N, K, EpiLinearV{epi}, plan, slab);
}

}  // namespace sttran
