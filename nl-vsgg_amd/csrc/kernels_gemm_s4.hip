// kernels_gemm_s4.hip -- one instantiation family of the nn.Linear GEMM (see gemm_launch.h)
#include "gemm_launch.h"

namespace sttran {

hipError_t gemm_linear_s4(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                          const EpiLinear& epi, GemmPlan plan, float* slab) {
  return gemm_generic<EpiScalar4<EpiLinear>, B_KMAJOR_PAD>(s, A, B, M, N, K, EpiScalar4<EpiLinear>{epi}, plan, slab);
}

}  // namespace sttran
