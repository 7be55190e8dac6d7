// Launcher + instantiations of the whole-forward tile kernel (dmpnn_mega_impl.hpp).
#include "dmpnn_mega_impl.hpp"

namespace dmpnn {
namespace mega {
DMPNN_DEFINE_MEGA(1)
DMPNN_DEFINE_MEGA(2)
DMPNN_DEFINE_MEGA(5)
}  // namespace mega

// Shapes the tile kernel takes (graph properties are decided on the device: plan flag bit 3).
bool mega_shapes_ok(const dmpnn_fwd_args& a) {
    // piece tiles come from the single-workgroup plan — or from the loader, at any batch size
    if (!(a.flags & DMPNN_F_LOADER_TILES) && !small_plan_fits(a.n_atoms, a.n_edges)) return false;
    if (a.n_atoms * a.ldv * 4 > 0x7FFFFFFF || a.n_edges * a.lde * 4 > 0x7FFFFFFF) return false;
    // (the generic path for oversize pieces stages operand rows of d_v + max(d_e, d_h) floats in the kernels' LDS)
    if (a.d_v + (a.d_e > a.d_h ? a.d_e : a.d_h) > 448) return false;
    return a.d_h % 4 == 0 && a.d_h <= 320 && a.d_v % 2 == 0 && a.d_e % 2 == 0 && a.ldv % 2 == 0 && a.lde % 2 == 0 &&
           (a.W_d != nullptr || a.ldout % 4 == 0) && a.ldh % 4 == 0 && a.depth >= 1 && !(a.flags & DMPNN_F_UNDIRECTED);
}

int launch_mega_forward(const dmpnn_fwd_args& a, float* out, int64_t ldout, hipStream_t s) {
    const int64_t nV = a.n_atoms, nE = a.n_edges;
    const PlanLayout L = plan_layout(nV, nE);
    const int* plan_i = static_cast<const int*>(a.plan);
    mega::MegaK g;
    memset(&g, 0, sizeof(g));
    g.mtile_row = plan_i + L.mtile_row; g.mtile_atom = plan_i + L.mtile_atom; g.row_ptr = plan_i + L.row_ptr;
    g.srcp = plan_i + L.srcp; g.perm = plan_i + L.perm; g.revp = plan_i + L.revp;
    g.flags = plan_i + DMPNN_HDR_FLAGS; g.poison_mask = kPlanNoMega;
    g.nV = (int)nV; g.nE = (int)nE; g.d_v = (int)a.d_v; g.d_e = (int)a.d_e; g.h = (int)a.d_h; g.depth = a.depth;
    g.V = a.V; g.ldv = (int)a.ldv; g.E = a.E ? a.E : a.V; g.lde = (int)a.lde;
    g.v_bytes = (unsigned)(nV * a.ldv * 4); g.e_bytes = a.E ? (unsigned)(nE * a.lde * 4) : 0u;
    g.W_i = a.W_i; g.b_i = a.b_i; g.W_h = a.W_h; g.b_h = a.b_h; g.W_o = a.W_o; g.b_o = a.b_o;
    g.act = a.act; g.slope = a.act_slope; g.slope_ptr = a.act_slope_ptr;
    g.out = out; g.ldout = (int)ldout;
    g.ldh = (int)a.ldh; g.slot = (long long)nE * a.ldh;
    if (a.flags & DMPNN_F_KEEP) { g.H0 = a.H0; g.Hs = a.Hs; g.Ms = a.Ms; g.Mv = a.Mv; }
    g.spill = (a.spill_ws && a.spill_bytes >= dmpnn_forward_spill_bytes(&a) && aligned16(a.spill_ws)) ? a.spill_ws : nullptr;
    const unsigned qn = (unsigned)(a.d_h / 4);
    g.qmagic = qn > 1 ? (unsigned)(((1ull << 32) + qn - 1) / qn) : 0u;
    const int n_tiles = (int)L.max_mtiles;
    if (a.d_h <= 64) return mega::launch_mega<1>(g, n_tiles, s);
    if (a.d_h <= 128) return mega::launch_mega<2>(g, n_tiles, s);
    return mega::launch_mega<5>(g, n_tiles, s);
}

}  // namespace dmpnn
