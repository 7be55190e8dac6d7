// Instantiations + host launcher of the backward tile kernel (dmpnn_mega16_bwd_impl.hpp).
#include <string.h>

#include "dmpnn_mega16_bwd_impl.hpp"

namespace dmpnn {
namespace mega16 {
DMPNN_DEFINE_MEGA16_BWD(1, true)
DMPNN_DEFINE_MEGA16_BWD(2, true)
DMPNN_DEFINE_MEGA16_BWD(5, true)
DMPNN_DEFINE_MEGA16_BWD(1, false)
DMPNN_DEFINE_MEGA16_BWD(2, false)
DMPNN_DEFINE_MEGA16_BWD(5, false)
// one tile as a 512-thread workgroup (launches of at most one tile per CU: tile_waves, dmpnn_mega16.hip)
DMPNN_DEFINE_MEGA16_BWD_NW(5, true, 8)
DMPNN_DEFINE_MEGA16_BWD_NW(5, false, 8)
}  // namespace mega16

static size_t al256b(size_t x) { return (x + 255) & ~size_t(255); }

// bytes of the two pre-split transposed [h, h] matrices
size_t mega16_bwd_wsplit_bytes(int64_t h) {
    const size_t NT = (size_t)(h + 15) / 16, nc = (size_t)(h + 31) / 32;
    return 2 * (al256b(NT * nc * 2048) + al256b((size_t)h * 4));
}

int launch_mega16_backward(const dmpnn_fwd_args& f, const float* gHO, int64_t ldg, const float* HO, int64_t ldho, float* gZO,
                           float* gZs, float* gH0, void* wsplit, float* sp_gM, float* sp_Ta, hipStream_t s, const float* g_edge, int64_t ld_gedge,
                           const Mega16BwdRows* rows) {
    const int64_t nV = f.n_atoms, nE = f.n_edges, h = f.d_h, dv = f.d_v;
    const size_t NT = (size_t)(h + 15) / 16, nc = (size_t)(h + 31) / 32;
    // the training forward already split both matrices behind its own pre-split weights (one launch for both passes) — unless the
    // caller vouched for a cached pre-split (DMPNN_F_WSPLIT_READY), which says nothing about this part
    const size_t fwd_part = mega16_fwd_wsplit_bytes(f);
    const bool from_fwd = (f.flags & DMPNN_F_KEEP) && !(f.flags & DMPNN_F_WSPLIT_READY) && f.wsplit &&
                          f.wsplit_bytes >= fwd_part + mega16_bwd_wsplit_bytes(h);
    unsigned char* ws = from_fwd ? static_cast<unsigned char*>(f.wsplit) + fwd_part : static_cast<unsigned char*>(wsplit);
    const size_t one = al256b(NT * nc * 2048) + al256b((size_t)h * 4);
    float* inv_o = reinterpret_cast<float*>(ws + al256b(NT * nc * 2048));
    float* inv_h = reinterpret_cast<float*>(ws + one + al256b(NT * nc * 2048));
    mega16::SplitArgs sp;
    memset(&sp, 0, sizeof(sp));
    sp.N = (int)h; sp.n_jobs = 2;
    // W'[n][k] = W_o[k][d_v + n]  and  W'[n][k] = W_h[k][n]: the matrices are read transposed
    sp.job[0] = mega16::SplitJob{f.W_o + dv, (int)(dv + h), 0, (int)h, 0, (int)h, ws, (int)nc, inv_o, 1};
    const bool atom = (f.flags & DMPNN_F_ATOM) != 0;   // atom messages: W_h is [h, h + d_e]; the data gradient runs through its first h columns
    sp.job[1] = mega16::SplitJob{f.W_h, (int)(atom ? h + f.d_e : h), 0, (int)h, 0, (int)h, ws + one, (int)nc, inv_h, 1};
    if (!from_fwd) {
        const unsigned waves = 2u * (unsigned)(((h + 15) / 16) * 16);
        hipLaunchKernelGGL(mega16::k_split_weights, dim3((waves + 3) / 4), dim3(256), 0, s, sp);
        DMPNN_CHECK_LAUNCH("k_split_weights");
    }

    const PlanLayout L = plan_layout(nV, nE);
    const int* plan_i = static_cast<const int*>(f.plan);
    mega16::Mega16BwdK g;
    memset(&g, 0, sizeof(g));
    g.mtile_row = plan_i + L.mtile_row; g.mtile_atom = plan_i + L.mtile_atom; g.row_ptr = plan_i + L.row_ptr; g.revp = plan_i + L.revp;
    g.flags = plan_i + DMPNN_HDR_FLAGS; g.poison_mask = kPlanNoMega;
    g.nV = (int)nV; g.nE = (int)nE; g.h = (int)h; g.depth = f.depth;
    g.act = f.act; g.slope = f.act_slope; g.slope_ptr = f.act_slope_ptr;
    g.gHO = gHO; g.ldg = (int)ldg; g.HO = HO; g.ldho = (int)ldho;
    g.H0 = f.H0; g.Hs = f.Hs; g.ldh = (int)f.ldh; g.slot = (long long)nE * f.ldh;
    g.gZO = gZO; g.gZs = gZs; g.gH0 = gH0;
    g.srcp = plan_i + L.srcp; g.d_v = (int)dv; g.W_o = f.W_o; g.W_h = f.W_h; g.sp_gM = sp_gM; g.sp_Ta = sp_Ta;
    g.atom = atom ? 1 : 0;
    g.g_edge = g_edge; g.ld_ge = (int)ld_gedge;
    if (rows && rows->gZ) {   // the gradients as split rows: the operands of the products on split rows (k_wgrad16r)
        g.gZrows = rows->gZ; g.gH0rows = rows->gH0; g.gZOrows = rows->gZO; g.tsr = rows->tsr; g.zrow_slot = (long long)nE * rows->tsr;
    }
    g.drop_scale = (f.dropout_p > 0.f && f.dropout_p < 1.f) ? 1.f / (1.f - f.dropout_p) : 0.f;
    if (f.keep_bits && (f.flags & DMPNN_F_TILE_PLAN)) {  // the forward kept H0 / H^(t) as sign bits (dmpnn_fwd_args.keep_bits)
        g.keep_bits = static_cast<const unsigned long long*>(f.keep_bits);
        g.bits_slot = (long long)L.max_mtiles * 256;
    }
    g.edge_index = reinterpret_cast<const long long*>(f.edge_index);   // (read only under a tile plan: header LIGHT == 2)
    g.rev64 = reinterpret_cast<const long long*>(f.rev_edge_index);
    g.WoMT = mega16::SplitW{ws, inv_o, (int)nc};
    g.WhT = mega16::SplitW{ws + one, inv_h, (int)nc};
    const int n_tiles = (f.n_tiles_launch > 0 && f.n_tiles_launch < L.max_mtiles) ? (int)f.n_tiles_launch : (int)L.max_mtiles;   // (as the forward: dmpnn_fwd_args.n_tiles_launch)
    const bool sa = f.act == DMPNN_ACT_NONE || f.act == DMPNN_ACT_RELU || f.act == DMPNN_ACT_LEAKYRELU;
    if (h <= 64) return sa ? mega16::launch_mega16_bwd<1, true>(g, n_tiles, s) : mega16::launch_mega16_bwd<1, false>(g, n_tiles, s);
    if (h <= 128) return sa ? mega16::launch_mega16_bwd<2, true>(g, n_tiles, s) : mega16::launch_mega16_bwd<2, false>(g, n_tiles, s);
    if (tile_waves(f, n_tiles) == 8) return sa ? mega16::launch_mega16_bwd<5, true, 8>(g, n_tiles, s) : mega16::launch_mega16_bwd<5, false, 8>(g, n_tiles, s);
    return sa ? mega16::launch_mega16_bwd<5, true>(g, n_tiles, s) : mega16::launch_mega16_bwd<5, false>(g, n_tiles, s);
}

}  // namespace dmpnn
