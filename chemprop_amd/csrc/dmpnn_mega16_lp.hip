// Instantiations of the whole-forward tile kernel on the HI halves alone (k_mpnn_tile16<..., LP = true>, dmpnn_mega16_impl.hpp;
// DMPNN_F_STORE16 on the tile route: OPT-IN, NOT fp32-class) — a translation unit of their own so that they compile beside dmpnn_mega16.hip.
#include "dmpnn_mega16_impl.hpp"

namespace dmpnn {
namespace mega16 {
DMPNN_DEFINE_MEGA16_LP(1, true, false, 4, true)
DMPNN_DEFINE_MEGA16_LP(1, false, false, 4, true)
DMPNN_DEFINE_MEGA16_LP(2, true, false, 4, true)
DMPNN_DEFINE_MEGA16_LP(2, false, false, 4, true)
DMPNN_DEFINE_MEGA16_LP(5, true, false, 4, true)
DMPNN_DEFINE_MEGA16_LP(5, false, false, 4, true)
DMPNN_DEFINE_MEGA16_LP(5, true, false, 8, true)
DMPNN_DEFINE_MEGA16_LP(5, false, false, 8, true)
}  // namespace mega16
}  // namespace dmpnn
