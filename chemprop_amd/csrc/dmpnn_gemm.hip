// K1 / K3 / K5 — dispatcher of the dense contractions of the path onto the fp32-MFMA kernel
// template of dmpnn_gemm_impl.hpp.
//
//   C[r, :] = act( [A1[g1(r), 0:K1] || A2[g2(r), 0:K2]] . W^T + bias + Cadd[r, :] )
//
//   K1 initialize  mixins.py:8-9    A1 = V gathered by src(e), A2 = E          W = W_i  (no act: H0)
//   K3 update      base.py:135-141  A1 = M                    Cadd = H0        W = W_h  act = tau
//   K5 finalize    base.py:180-183  A1 = V, A2 = Mv           bias = b_o       W = W_o  act = tau
//                  base.py:185-188  A1 = H_v, A2 = V_d        bias = b_d       W = W_d  (no act)
//
// The reference materialises torch.cat(...) ([E, d_v+d_e] / [V, d_v+d_h]) and the gathered
// V[src] before every nn.Linear; here concatenation and gather happen in the A-operand loader.
//
// Instantiation choice
//   G   operand load granularity (floats per buffer load): 4 when every operand row is 16-byte
//       aligned and K1, K2 are multiples of 4 (W_h, W_o: 300/372 columns), 2 for 8-byte alignment
//       (W_i: 86 = 72 + 14 columns; CGR 134 = 106 + 28), 1 otherwise (odd test shapes).
//   WN  column tiles per wave: 5 (BN = 320 >= d_h = 300: whole rows per workgroup), 4, 2 or 1 —
//       the one that pads N least.
//   RT  row tiles: 3 (48 rows) unless the matrix is so short that 16-row panels fill more CUs.
#include <stdlib.h>

#include "dmpnn_gemm_impl.hpp"

namespace dmpnn {

namespace {

using gemm::GemmK;


inline bool al(const void* p, int bytes) { return (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(bytes - 1)) == 0; }

// largest G in {4, 2, 1} every operand of the contraction supports
int pick_g(const dmpnn_gemm_args& a) {
    for (int G : {4, 2}) {
        const int B = 4 * G;
        bool ok = (a.K1 % G == 0) && (a.K2 % G == 0) && (a.ldw % G == 0) && al(a.W, B);
        if (a.K1 > 0) ok = ok && (a.lda1 % G == 0) && al(a.A1, B);
        if (a.K2 > 0) ok = ok && (a.lda2 % G == 0) && al(a.A2, B);
        if (a.Cadd) ok = ok && (a.ldcadd % G == 0) && al(a.Cadd, B) && (a.N % G == 0);
        if (ok) return G;
    }
    return 1;
}

int pick_wn(int64_t N, int64_t M) {
    static const int cand[] = {5, 4, 2, 1};
    int best = 5;
    int64_t best_cost = INT64_MAX;
    for (int wn : cand) {
        const int64_t cost = ((N + 64 * wn - 1) / (64 * wn)) * wn;
        if (cost < best_cost) { best_cost = cost; best = wn; }
    }
    // A short operand (a predictor layer on 512 molecules: 32 row tiles) leaves most of the chip idle behind one column block per
    // row tile, and each of those workgroups streams the WHOLE weight matrix through a latency-bound chunk loop.  The narrowest
    // build with the same padded width gives every 64-column slice its own workgroup (same arithmetic per output element:
    // row-tile scale, per-column weight scale, chunk order).
    const int64_t row_tiles = (M + 15) / 16;
    if (row_tiles * ((N + 64 * best - 1) / (64 * best)) <= 128)
        for (int wn : {1, 2, 4})
            if (wn < best && ((N + 64 * wn - 1) / (64 * wn)) * wn == best_cost && row_tiles * ((N + 64 * wn - 1) / (64 * wn)) <= 512) return wn;
    return best;
}

// Row-tile height: minimise (#rounds over 256 CUs) x (rows per tile + fixed per-tile cost).
int pick_rt(int64_t M, int64_t n_col_blocks) {
    int best = 3;
    double best_cost = 1e300;
    for (int rt : {3, 2, 1}) {
        const int64_t tiles = ((M + 16 * rt - 1) / (16 * rt)) * n_col_blocks;
        const int64_t rounds = (tiles + 255) / 256;
        const double cost = (double)rounds * (rt + 0.3);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = rt; }
    }
    return best;
}

}  // namespace

int launch_linear_ex(const dmpnn_gemm_args& a0, const GemmExtra& x, hipStream_t s) {
    dmpnn_gemm_args a = a0;
    DMPNN_CHECK_ARG(a.M >= 0 && a.N >= 0 && a.K1 >= 0 && a.K2 >= 0, "linear: negative size");
    if (a.M == 0 || a.N == 0) return DMPNN_OK;
    DMPNN_CHECK_ARG(a.W, "linear: null W");
    DMPNN_CHECK_ARG(a.C || a.Zpre || x.seg, "linear: no output");
    DMPNN_CHECK_ARG(a.K1 + a.K2 > 0, "linear: empty contraction");
    DMPNN_CHECK_ARG(a.K1 == 0 || a.A1, "linear: null A1 with K1 > 0");
    DMPNN_CHECK_ARG(a.K2 == 0 || a.A2, "linear: null A2 with K2 > 0");
    DMPNN_CHECK_ARG(a.M < (int64_t(1) << 31) && a.N < (1 << 24) && a.K1 + a.K2 < (1 << 24), "linear: size out of range");
    const int* gather2 = x.gather2;
    int64_t rows1 = a.gather1 ? a.gather1_rows : a.M, rows2 = gather2 ? x.gather2_rows : a.M;
    if (a.K1 == 0) {  // single operand lives in the A1 slot
        a.A1 = a.A2; a.lda1 = a.lda2; a.K1 = a.K2; a.gather1 = gather2; rows1 = rows2;
        a.A2 = nullptr; a.K2 = 0; a.lda2 = 0; gather2 = nullptr;
    }
    // 32-bit buffer addressing: a gathered source tensor must span < 2 GiB; contiguous operands are
    // addressed relative to their row tile, so only one tile's extent matters
    const int64_t lim = 0x7FFFFFFF;
    if (a.gather1) {
        const int64_t b1 = (rows1 > 0 ? rows1 : (lim / 4) / (a.lda1 > 0 ? a.lda1 : 1)) * a.lda1 * 4;
        DMPNN_CHECK_ARG(rows1 <= 0 || b1 <= lim, "linear: gathered A1 spans %lld bytes (>= 2 GiB): split the batch", (long long)b1);
    }
    if (gather2 && a.K2 > 0) {
        const int64_t b2 = rows2 * a.lda2 * 4;
        DMPNN_CHECK_ARG(rows2 > 0 && b2 <= lim, "linear: gathered A2 spans %lld bytes (>= 2 GiB or unknown)", (long long)b2);
    }
    DMPNN_CHECK_ARG(a.N * a.ldw * 4 <= lim, "linear: weight matrix >= 2 GiB");
    DMPNN_CHECK_ARG(a.lda1 * 4 * 64 <= lim && a.lda2 * 4 * 64 <= lim && a.ldcadd * 4 * 64 <= lim, "linear: leading dimension too large");

    GemmK g;
    memset(&g, 0, sizeof(g));
    g.M = (int)a.M; g.N = (int)a.N; g.K1 = (int)a.K1; g.K2 = (int)a.K2;
    g.A1 = a.A1; g.A2 = a.K2 > 0 ? a.A2 : a.A1; g.W = a.W; g.bias = a.bias; g.Cadd = a.Cadd;
    g.C = a.C; g.Zpre = a.Zpre;
    g.gather1 = a.gather1; g.gather2 = a.K2 > 0 ? gather2 : nullptr;
    g.lda1 = (int)a.lda1; g.lda2 = (int)(a.K2 > 0 ? a.lda2 : 0); g.ldw = (int)a.ldw; g.ldcadd = (int)a.ldcadd;
    g.ldc = (int)a.ldc; g.ldz = (int)a.ldz;
    g.a1_bytes = a.gather1 ? (rows1 > 0 ? (unsigned)(rows1 * a.lda1 * 4) : 0x7FFFFFFFu) : 0u;
    g.a2_bytes = (gather2 && a.K2 > 0) ? (unsigned)(rows2 * a.lda2 * 4) : 0u;
    g.act = a.act; g.slope = a.act_slope; g.slope_ptr = a.act_slope_ptr;
    const bool vec_c = (a.N % 4 == 0) && (!a.C || (a.ldc % 4 == 0 && al(a.C, 16))) && (!a.Zpre || (a.ldz % 4 == 0 && al(a.Zpre, 16)));
    g.poison_flags = x.poison_flags; g.poison_mask = x.poison_mask;

    // builds with G >= 2 store 16-byte row segments: outputs that cannot take them use the any-shape build
    const int G = vec_c ? pick_g(a) : 1;
    const bool has_a2 = a.K2 > 0;

    if (x.seg) {
        DMPNN_CHECK_ARG(x.tile_row && x.tile_atom && x.row_ptr && x.revp && x.n_tiles >= 0, "linear(seg): missing tile tables");
        DMPNN_CHECK_ARG(a.N % 4 == 0 && a.N <= 320, "linear(seg): d_h must be a multiple of 4 and <= 320 (got %lld)", (long long)a.N);
        DMPNN_CHECK_ARG(G >= 2, "linear(seg): operands must be 8-byte aligned with even widths");
        DMPNN_CHECK_ARG(!x.Mout || (x.ldm % 4 == 0 && al(x.Mout, 16)), "linear(seg): Mout misaligned");
        DMPNN_CHECK_ARG(!x.Sout || (x.lds % 4 == 0 && al(x.Sout, 16)), "linear(seg): Sout misaligned");
        DMPNN_CHECK_ARG(vec_c, "linear(seg): C / Zpre must be 16-byte aligned");
        if (x.n_tiles == 0) return DMPNN_OK;
        g.tile_row = x.tile_row; g.tile_atom = x.tile_atom; g.row_ptr = x.row_ptr; g.revp = x.revp;
        g.Mout = x.Mout; g.ldm = (int)x.ldm; g.Sout = x.Sout; g.lds = (int)x.lds;
        const unsigned qn = (unsigned)(a.N / 4);
        g.qmagic = qn > 1 ? (unsigned)(((1ull << 32) + qn - 1) / qn) : 0u;
        const int wn = a.N <= 64 ? 1 : (a.N <= 128 ? 2 : 5);
        const int key = wn * 100 + G * 10 + (has_a2 ? 1 : 0);
        switch (key) {
#define SEG_CASE(WN_, G_, A2_) \
    case WN_ * 100 + G_ * 10 + (A2_ ? 1 : 0): return gemm::launch_gemm<3, WN_, G_, A2_, gemm::EPI_SEG>(g, x.n_tiles, s);
            SEG_CASE(1, 4, false) SEG_CASE(1, 4, true) SEG_CASE(1, 2, true)
            SEG_CASE(2, 4, false) SEG_CASE(2, 4, true) SEG_CASE(2, 2, true)
            SEG_CASE(5, 4, false) SEG_CASE(5, 4, true) SEG_CASE(5, 2, true)
#undef SEG_CASE
            case 120: return gemm::launch_gemm<3, 1, 2, true, gemm::EPI_SEG>(g, x.n_tiles, s);  // G = 2 single operand: A2 slot reads nothing
            case 220: return gemm::launch_gemm<3, 2, 2, true, gemm::EPI_SEG>(g, x.n_tiles, s);
            case 520: return gemm::launch_gemm<3, 5, 2, true, gemm::EPI_SEG>(g, x.n_tiles, s);
        }
        set_error("linear(seg): no instantiation for wn=%d G=%d a2=%d", wn, G, (int)has_a2);
        return DMPNN_EINVAL;
    }

    if (x.tile_row) { g.tile_row = x.tile_row; }
    const int wn = pick_wn(a.N, x.tile_row ? (int64_t(1) << 40) : a.M);
    const int64_t ncb = (a.N + 64 * wn - 1) / (64 * wn);
    int rt = x.tile_row ? 3 : pick_rt(a.M, ncb);
    if (rt == 2 && G != 4) rt = 3;  // 32-row panels are only built for the 16-byte operand path
    const int n_tiles = x.tile_row ? x.n_tiles : (int)((a.M + 16 * rt - 1) / (16 * rt));
    // variant: 0 = G4 single operand, 1 = G4 two operands, 2 = G2 (two-operand build), 3 = G1 (two-operand build)
    const int var = G == 4 ? (has_a2 ? 1 : 0) : (G == 2 ? 2 : 3);
    const int key = rt * 100 + wn * 10 + var;
    switch (key) {
#define PLAIN_CASES(RT_, WN_)                                                                                          \
    case RT_ * 100 + WN_ * 10 + 0: return gemm::launch_gemm<RT_, WN_, 4, false, gemm::EPI_PLAIN>(g, n_tiles, s);       \
    case RT_ * 100 + WN_ * 10 + 1: return gemm::launch_gemm<RT_, WN_, 4, true, gemm::EPI_PLAIN>(g, n_tiles, s);        \
    case RT_ * 100 + WN_ * 10 + 2: return gemm::launch_gemm<RT_, WN_, 2, true, gemm::EPI_PLAIN>(g, n_tiles, s);        \
    case RT_ * 100 + WN_ * 10 + 3: return gemm::launch_gemm<RT_, WN_, 1, true, gemm::EPI_PLAIN>(g, n_tiles, s);
        PLAIN_CASES(1, 1) PLAIN_CASES(1, 2) PLAIN_CASES(1, 4) PLAIN_CASES(1, 5)
#define PLAIN_G4(RT_, WN_)                                                                                           \
    case RT_ * 100 + WN_ * 10 + 0: return gemm::launch_gemm<RT_, WN_, 4, false, gemm::EPI_PLAIN>(g, n_tiles, s);     \
    case RT_ * 100 + WN_ * 10 + 1: return gemm::launch_gemm<RT_, WN_, 4, true, gemm::EPI_PLAIN>(g, n_tiles, s);
        PLAIN_G4(2, 1) PLAIN_G4(2, 2) PLAIN_G4(2, 4) PLAIN_G4(2, 5)
#undef PLAIN_G4
        PLAIN_CASES(3, 1) PLAIN_CASES(3, 2) PLAIN_CASES(3, 4) PLAIN_CASES(3, 5)
#undef PLAIN_CASES
    }
    set_error("linear: no instantiation for rt=%d wn=%d G=%d", rt, wn, G);
    return DMPNN_EINVAL;
}

int launch_linear(const dmpnn_gemm_args& a, hipStream_t s) {
    GemmExtra x;
    memset(&x, 0, sizeof(x));
    return launch_linear_ex(a, x, s);
}

}  // namespace dmpnn
