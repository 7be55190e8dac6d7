// Instantiations of the fp32-MFMA contraction (see dmpnn_gemm_impl.hpp); split over several
// translation units so the build runs in parallel.
#include "dmpnn_gemm_impl.hpp"

namespace dmpnn {
namespace gemm {
DMPNN_DEFINE_GEMM(1, 1, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(1, 2, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(1, 4, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(1, 5, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(3, 1, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(3, 2, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(3, 4, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(3, 5, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(2, 1, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(2, 2, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(2, 4, 4, false, EPI_PLAIN)
DMPNN_DEFINE_GEMM(2, 5, 4, false, EPI_PLAIN)
}  // namespace gemm
}  // namespace dmpnn
