// Launcher + instantiations of the split-MFMA whole-forward tile kernel (dmpnn_mega16_impl.hpp).
#include <algorithm>
#include <cstdlib>

#include "dmpnn_mega16_impl.hpp"

namespace dmpnn {
namespace mega16 {
DMPNN_DEFINE_MEGA16(1, true, true)
DMPNN_DEFINE_MEGA16(1, true, false)
DMPNN_DEFINE_MEGA16(1, false, true)
DMPNN_DEFINE_MEGA16(1, false, false)
DMPNN_DEFINE_MEGA16(2, true, true)
DMPNN_DEFINE_MEGA16(2, true, false)
DMPNN_DEFINE_MEGA16(2, false, true)
DMPNN_DEFINE_MEGA16(2, false, false)
DMPNN_DEFINE_MEGA16(5, true, true)
DMPNN_DEFINE_MEGA16(5, true, false)
DMPNN_DEFINE_MEGA16(5, false, true)
DMPNN_DEFINE_MEGA16(5, false, false)
// one tile as a 512-thread workgroup (launches of at most one tile per CU)
DMPNN_DEFINE_MEGA16_NW(5, true, true, 8)
DMPNN_DEFINE_MEGA16_NW(5, true, false, 8)
DMPNN_DEFINE_MEGA16_NW(5, false, true, 8)
DMPNN_DEFINE_MEGA16_NW(5, false, false, 8)
// the HI-halves-alone forms (DMPNN_F_STORE16, inference): defined in dmpnn_mega16_lp.hip
#define DMPNN_DECLARE_MEGA16_LP(WN, SA, NW) template <> int launch_mega16<WN, SA, false, NW, true>(const Mega16K& g, int n_tiles, hipStream_t s);
DMPNN_DECLARE_MEGA16_LP(1, true, 4) DMPNN_DECLARE_MEGA16_LP(1, false, 4) DMPNN_DECLARE_MEGA16_LP(2, true, 4) DMPNN_DECLARE_MEGA16_LP(2, false, 4)
DMPNN_DECLARE_MEGA16_LP(5, true, 4) DMPNN_DECLARE_MEGA16_LP(5, false, 4) DMPNN_DECLARE_MEGA16_LP(5, true, 8) DMPNN_DECLARE_MEGA16_LP(5, false, 8)
#undef DMPNN_DECLARE_MEGA16_LP
}  // namespace mega16

// (per calling thread: a diagnostic hook, never shared mutable state between threads that drive the library)
thread_local long long* g_debug_stamps = nullptr;
thread_local AggRide g_agg_ride = {nullptr, 0, nullptr, nullptr, 0, 0, 0.f, false};

namespace {
inline size_t al256(size_t x) { return (x + 255) & ~size_t(255); }
struct WsLayout {
    size_t wi, wh, wom, wov, sc_i, sc_h, sc_o, whe, sc_e, total;
    int nc_i, nc_h, nc_v;
};
WsLayout ws_layout(const dmpnn_fwd_args& a) {
    WsLayout L;
    const size_t N = (size_t)a.d_h;
    const bool atom = (a.flags & DMPNN_F_ATOM) != 0;  // atom messages: W_i is [N, d_v]; W_h[:, N:N + d_e] gets a block of its own (one chunk)
    L.nc_i = (int)((a.d_v + (atom ? 0 : a.d_e) + 31) / 32); L.nc_h = (int)((a.d_h + 31) / 32); L.nc_v = (int)((a.d_v + 31) / 32);
    size_t o = 0;
    const size_t NT = (N + 15) / 16;  // column tiles: the packed layout is [tile][chunk][hi|lo][64 lanes][16 B]
    L.wi = o; o += al256(NT * L.nc_i * 2048);
    L.wh = o; o += al256(NT * L.nc_h * 2048);
    L.wom = o; o += al256(NT * L.nc_h * 2048);
    L.wov = o; o += al256(NT * L.nc_v * 2048);
    L.sc_i = o; o += al256(N * 4);
    L.sc_h = o; o += al256(N * 4);
    L.sc_o = o; o += al256(N * 4);
    L.whe = o; o += atom ? al256(NT * 1 * 2048) : 0;
    L.sc_e = o; o += atom ? al256(N * 4) : 0;
    L.total = o;
    return L;
}
}  // namespace

// (a training forward — DMPNN_F_KEEP — also splits the two transposed matrices of the backward tile kernel: one launch for both
//  passes; the backward finds them behind the forward's own part)
size_t mega16_wsplit_bytes(const dmpnn_fwd_args& a) {
    return ws_layout(a).total + ((a.flags & DMPNN_F_KEEP) ? mega16_bwd_wsplit_bytes(a.d_h) : 0);
}
size_t mega16_fwd_wsplit_bytes(const dmpnn_fwd_args& a) { return ws_layout(a).total; }

// the jobs of the forward's weight pre-split (W_i | W_h | W_o[:, d_v:] | W_o[:, :d_v]; training: + the backward tile kernel's two
// transposed matrices; atom messages: + W_h[:, N:]) into the caller's `wsplit`; false: no workspace / too small
bool mega16_split_args(const dmpnn_fwd_args& a, mega16::SplitArgs* spp) {
    const WsLayout W = ws_layout(a);
    if (!a.wsplit || a.wsplit_bytes < W.total) return false;
    unsigned char* ws = static_cast<unsigned char*>(a.wsplit);
    const int N = (int)a.d_h, dv = (int)a.d_v, de = (int)a.d_e;
    mega16::SplitArgs& sp = *spp;
    memset(&sp, 0, sizeof(sp));
    sp.N = N; sp.n_jobs = 4;
    const bool atom = (a.flags & DMPNN_F_ATOM) != 0;
    const int ki = atom ? dv : dv + de, ldwh = atom ? N + de : N;   // (atom messages: W_i [N, d_v], W_h [N, N + d_e])
    sp.job[0] = mega16::SplitJob{a.W_i, ki, 0, ki, 0, ki, ws + W.wi, W.nc_i, reinterpret_cast<float*>(ws + W.sc_i)};
    sp.job[1] = mega16::SplitJob{a.W_h, ldwh, 0, N, 0, N, ws + W.wh, W.nc_h, reinterpret_cast<float*>(ws + W.sc_h)};
    sp.job[2] = mega16::SplitJob{a.W_o, dv + N, dv, N, 0, dv + N, ws + W.wom, W.nc_h, reinterpret_cast<float*>(ws + W.sc_o)};
    sp.job[3] = mega16::SplitJob{a.W_o, dv + N, 0, dv, 0, dv + N, ws + W.wov, W.nc_v, nullptr};
    if ((a.flags & DMPNN_F_KEEP) && a.wsplit_bytes >= W.total + mega16_bwd_wsplit_bytes(N)) {
        // W'[n][k] = W_o[k][d_v + n]  and  W'[n][k] = W_h[k][n]: what k_mpnn_tile16_bwd contracts with, read transposed
        const size_t NT = (size_t)(N + 15) / 16, nch = (size_t)(N + 31) / 32;
        const size_t one = al256(NT * nch * 2048) + al256((size_t)N * 4);
        unsigned char* wb = ws + W.total;
        sp.job[4] = mega16::SplitJob{a.W_o + dv, dv + N, 0, N, 0, N, wb, (int)nch, reinterpret_cast<float*>(wb + al256(NT * nch * 2048)), 1};
        sp.job[5] = mega16::SplitJob{a.W_h, ldwh, 0, N, 0, N, wb + one, (int)nch, reinterpret_cast<float*>(wb + one + al256(NT * nch * 2048)), 1};
        sp.n_jobs = 6;
    }
    if (atom) {  // the bond-feature block W_h[:, N:N + d_e], its own row scales
        sp.job[sp.n_jobs] = mega16::SplitJob{a.W_h, ldwh, N, de, N, de, ws + W.whe, 1, reinterpret_cast<float*>(ws + W.sc_e)};
        ++sp.n_jobs;
    }
    return true;
}

// A training forward of the tile kernel (bond messages) keeps M^(t) as SPLIT ROWS in `msplit` — depth - 1 slots of n_edges rows of
// split_row_floats(d_h) floats — when the caller provides that buffer: the weight-gradient product (k_wgrad16r) reads them as they are,
// and so does dmpnn_backward decide (the same test on the same argument block)
bool mega16_keeps_rows(const dmpnn_fwd_args& a) {
    const unsigned need = DMPNN_F_FUSED | DMPNN_F_MEGA | DMPNN_F_SPLIT16 | DMPNN_F_KEEP;
    if ((a.flags & need) != need || (a.flags & DMPNN_F_ATOM) || a.depth < 2 || a.n_edges <= 0 || a.d_h > 320) return false;
    const size_t bytes = (size_t)(a.depth - 1) * (size_t)a.n_edges * (size_t)(split_row_floats(a.d_h) * 4);
    return a.msplit && aligned16(a.msplit) && a.msplit_bytes >= bytes;
}

// Waves per tile workgroup (d_h in (128, 320]).  The host does not know the tile count of a plan built on the device, only its
// launch bound; what it knows is the batch: a launch whose tiles fit the chip ONE per CU (about n_edges / 40 and n_atoms / 20 tiles
// for molecules packed whole into 48-row / 32-atom tiles) runs each tile as 8 waves — two per SIMD out of ONE tile — and a larger
// launch as 4-wave workgroups, two tiles per CU.  DMPNN_TILE_WAVES=4|8 overrides (measurement builds, tests of both forms).
int tile_waves(const dmpnn_fwd_args& a, int n_tiles) {
    static const int forced = [] { const char* e = getenv("DMPNN_TILE_WAVES"); return e ? atoi(e) : 0; }();
    if (forced == 4 || forced == 8) return forced;
    if (a.d_h <= 128 || a.d_h > 320) return 4;
    static const int n_cu = [] {   // (read-only device property, fetched once)
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    if (n_tiles <= n_cu) return 8;
    // (with a margin of 1 / 16: 576 QM9-shaped molecules — 251 by this estimate, 259 tiles in fact — ran their last three tiles as a second
    //  round of 8-wave workgroups, 64 us where two co-resident 4-wave tiles per CU take 45: profiles/r06_l2_warm_sizes.txt)
    const int64_t est = std::max<int64_t>((a.n_edges + 39) / 40, (a.n_atoms + 19) / 20);
    return est + est / 16 <= n_cu ? 8 : 4;
}

int launch_mega16_forward(const dmpnn_fwd_args& a, float* out, int64_t ldout, hipStream_t s) {
    const int64_t nV = a.n_atoms, nE = a.n_edges;
    const WsLayout W = ws_layout(a);
    if (!a.wsplit || a.wsplit_bytes < W.total) {
        set_error("forward(split16): wsplit workspace missing or too small (%zu < %zu bytes)", a.wsplit_bytes, W.total);
        return DMPNN_ENOSPC;
    }
    unsigned char* ws = static_cast<unsigned char*>(a.wsplit);
    const int N = (int)a.d_h, dv = (int)a.d_v, de = (int)a.d_e;
    // ---- pre-split of the weight matrices: once per forward — in a launch of its own here, unless it rode in K0's launch (the
    // caller then passes DMPNN_F_WSPLIT_READY: dmpnn_forward_tiles, dmpnn_train_step) ----
    if (!(a.flags & DMPNN_F_WSPLIT_READY)) {
        mega16::SplitArgs sp;
        if (!mega16_split_args(a, &sp)) { set_error("forward(split16): wsplit workspace missing or too small"); return DMPNN_ENOSPC; }
        hipLaunchKernelGGL(mega16::k_split_weights, dim3((unsigned)(((N + 15) / 16) * 4 * sp.n_jobs)), dim3(256), 0, s, sp);  // jobs x whole column tiles, 4 waves per block
        DMPNN_CHECK_LAUNCH("k_split_weights");
    }

    const PlanLayout L = plan_layout(nV, nE);
    const int* plan_i = static_cast<const int*>(a.plan);
    mega16::Mega16K G;
    memset(&G, 0, sizeof(G));
    mega::MegaK& g = G.m;
    g.mtile_row = plan_i + L.mtile_row; g.mtile_atom = plan_i + L.mtile_atom; g.row_ptr = plan_i + L.row_ptr;
    g.srcp = plan_i + L.srcp; g.perm = plan_i + L.perm; g.revp = plan_i + L.revp;
    g.flags = plan_i + DMPNN_HDR_FLAGS; g.poison_mask = kPlanNoMega;
    g.nV = (int)nV; g.nE = (int)nE; g.d_v = dv; g.d_e = (a.flags & DMPNN_F_ATOM) ? 0 : de; g.h = N; g.depth = a.depth;  // (atom: E is not a K1 operand)
    g.V = a.V; g.ldv = (int)a.ldv; g.E = a.E ? a.E : a.V; g.lde = (int)a.lde;
    g.v_bytes = (unsigned)(nV * a.ldv * 4); g.e_bytes = a.E ? (unsigned)(nE * a.lde * 4) : 0u;
    g.W_i = a.W_i; g.b_i = a.b_i; g.W_h = a.W_h; g.b_h = a.b_h; g.W_o = a.W_o; g.b_o = a.b_o;
    g.act = a.act; g.slope = a.act_slope; g.slope_ptr = a.act_slope_ptr;
    g.out = out; g.ldout = (int)ldout;
    g.ldh = (int)a.ldh; g.slot = (long long)nE * a.ldh;
    if (a.flags & DMPNN_F_KEEP) { g.H0 = a.H0; g.Hs = a.Hs; g.Ms = a.Ms; g.Mv = a.Mv; }
    if ((a.flags & DMPNN_F_KEEP) && a.keep_bits) {  // (validated by dmpnn_forward) H0 / H^(t) as sign bits: slot 0 = H0, slot t = H^(t)
        G.keep_bits = static_cast<unsigned long long*>(a.keep_bits);
        G.bits_slot = (long long)L.max_mtiles * 256;
    }
    g.spill = (a.spill_ws && a.spill_bytes >= dmpnn_forward_spill_bytes(&a) && aligned16(a.spill_ws)) ? a.spill_ws : nullptr;
    G.Wi = mega16::SplitW{ws + W.wi, reinterpret_cast<const float*>(ws + W.sc_i), W.nc_i};
    G.Wh = mega16::SplitW{ws + W.wh, reinterpret_cast<const float*>(ws + W.sc_h), W.nc_h};
    G.WoM = mega16::SplitW{ws + W.wom, reinterpret_cast<const float*>(ws + W.sc_o), W.nc_h};
    G.WoV = mega16::SplitW{ws + W.wov, reinterpret_cast<const float*>(ws + W.sc_o), W.nc_v};
    if (a.flags & DMPNN_F_ATOM) {
        G.WhE = mega16::SplitW{ws + W.whe, reinterpret_cast<const float*>(ws + W.sc_e), 1};
        G.atom_de = de;
        if ((a.flags & DMPNN_F_KEEP) && a.depth > 1 && nE > 0) {  // (validated by dmpnn_forward: `msplit` holds depth - 1 slots of [n_edges][16])
            G.atom_me = static_cast<float*>(a.msplit);
            G.me_slot = (long long)nE * mega16::kAtomK;
        }
    }
    if (mega16_keeps_rows(a)) {
        G.Mrows = static_cast<unsigned char*>(a.msplit);
        G.tsr = (int)(split_row_floats(a.d_h) * 4);
        G.mrow_slot = (long long)nE * G.tsr;
    }
    g.dbg = g_debug_stamps;
    if ((a.flags & DMPNN_F_KEEP) && g_agg_ride.Hm && g_agg_ride.n_mols > 0 && out == a.out && N % 4 == 0) {
        // (dmpnn_train_step: the aggregate of every molecule of a regular tile leaves with the tile, Mega16K::agg_*)
        G.agg_Hm = g_agg_ride.Hm; G.agg_ld = g_agg_ride.ld; G.agg_batch = reinterpret_cast<const long long*>(g_agg_ride.batch);
        G.agg_bounds = g_agg_ride.table; G.agg_done = g_agg_ride.table + 2 * g_agg_ride.n_mols + 4; G.agg_n_mols = g_agg_ride.n_mols;
        G.agg_mode = g_agg_ride.mode; G.agg_norm = g_agg_ride.norm;
        g_agg_ride.taken = true;
    }
    if (a.dropout_p > 0.f && a.dropout_p < 1.f) {  // (validated by dmpnn_forward: training forward, ReLU-class activation, no W_d)
        g.drop_thr = drop_threshold(a.dropout_p); g.drop_scale = 1.f / (1.f - a.dropout_p);
        g.seed_lo = (unsigned)(a.dropout_seed & 0xFFFFFFFFull); g.seed_hi = (unsigned)(a.dropout_seed >> 32);
    }
    g.edge_index = reinterpret_cast<const long long*>(a.edge_index);
    g.rev64 = reinterpret_cast<const long long*>(a.rev_edge_index);
    const int n_tiles = (a.n_tiles_launch > 0 && a.n_tiles_launch < L.max_mtiles) ? (int)a.n_tiles_launch : (int)L.max_mtiles;
    const bool sa = !(a.act == DMPNN_ACT_TANH || a.act == DMPNN_ACT_ELU), kp = (a.flags & DMPNN_F_KEEP) != 0;
    const int wn = a.d_h <= 64 ? 1 : (a.d_h <= 128 ? 2 : 5);
    if (a.flags & DMPNN_F_STORE16) {   // (validated by dmpnn_forward: inference, bond messages) every product on the hi halves alone
        if (wn == 1) return sa ? mega16::launch_mega16<1, true, false, 4, true>(G, n_tiles, s) : mega16::launch_mega16<1, false, false, 4, true>(G, n_tiles, s);
        if (wn == 2) return sa ? mega16::launch_mega16<2, true, false, 4, true>(G, n_tiles, s) : mega16::launch_mega16<2, false, false, 4, true>(G, n_tiles, s);
        if (tile_waves(a, n_tiles) == 8)
            return sa ? mega16::launch_mega16<5, true, false, 8, true>(G, n_tiles, s) : mega16::launch_mega16<5, false, false, 8, true>(G, n_tiles, s);
        return sa ? mega16::launch_mega16<5, true, false, 4, true>(G, n_tiles, s) : mega16::launch_mega16<5, false, false, 4, true>(G, n_tiles, s);
    }
#define DMPNN_PICK(WN) (sa ? (kp ? mega16::launch_mega16<WN, true, true>(G, n_tiles, s) : mega16::launch_mega16<WN, true, false>(G, n_tiles, s)) \
                           : (kp ? mega16::launch_mega16<WN, false, true>(G, n_tiles, s) : mega16::launch_mega16<WN, false, false>(G, n_tiles, s)))
    if (wn == 1) return DMPNN_PICK(1);
    if (wn == 2) return DMPNN_PICK(2);
    if (tile_waves(a, n_tiles) == 8)
        return sa ? (kp ? mega16::launch_mega16<5, true, true, 8>(G, n_tiles, s) : mega16::launch_mega16<5, true, false, 8>(G, n_tiles, s))
                  : (kp ? mega16::launch_mega16<5, false, true, 8>(G, n_tiles, s) : mega16::launch_mega16<5, false, false, 8>(G, n_tiles, s));
    return DMPNN_PICK(5);
#undef DMPNN_PICK
}

}  // namespace dmpnn
