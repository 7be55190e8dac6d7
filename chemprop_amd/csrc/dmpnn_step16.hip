// Host side of the per-step FUSED route on the f16 matrix pipe (dmpnn_step16_impl.hpp): instantiations and the launch
// chain of one inference forward  K1 (+ first message)  ->  (depth - 1) x update (+ next message / Mv)  ->  finalize.
#include <stdlib.h>
#include <string.h>

#include "dmpnn_step16_impl.hpp"

namespace dmpnn {
extern thread_local long long* g_debug_stamps;
namespace step16 {
DMPNN_DEFINE_STEP16(1, 4)
DMPNN_DEFINE_STEP16(2, 4)
DMPNN_DEFINE_STEP16(3, 4)
DMPNN_DEFINE_STEP16(4, 4)
DMPNN_DEFINE_STEP16(5, 4)
DMPNN_DEFINE_STEP16(3, 8)
DMPNN_DEFINE_STEP16(4, 8)
DMPNN_DEFINE_STEP16(5, 8)
}  // namespace step16
namespace rows16 {
DMPNN_DEFINE_ROWS16_X(1, 4, true)
DMPNN_DEFINE_ROWS16_X(2, 4, true)
DMPNN_DEFINE_ROWS16_X(3, 4, true)
DMPNN_DEFINE_ROWS16_X(4, 4, true)
DMPNN_DEFINE_ROWS16_X(5, 4, true)
}  // namespace rows16

int64_t split_row_floats(int64_t d_h) { return step16::split_row_bytes((int)d_h) / 4; }
static inline bool half_store(const dmpnn_fwd_args& a) { return (a.flags & DMPNN_F_STORE16) != 0; }
static inline int msg_row_bytes(const dmpnn_fwd_args& a) { return half_store(a) ? step16::half_row_bytes((int)a.d_h) : step16::split_row_bytes((int)a.d_h); }

// LEAN training forward (round 4): DMPNN_F_KEEP with `keep_bits` on this route.  Nothing is kept that the backward step kernels
// (dmpnn_bstep16.hip) do not read: the split message rows of EVERY step (depth - 1 slots in `msplit` instead of two ping-pong
// slots + an fp32 copy each), the split K1 operand (in `H0`: the residual is recomputed per step as in inference, no H0 tensor),
// one SIGN bit per element of H0 / H^(t) (`keep_bits`, instead of fp32 rows), and the fp32 per-atom sums `Mv`.
// (the split K1 operand's rows are packed at their own stride — split_operand_bytes(d_v + d_e) = ceil((d_v + d_e) / 32) * 128 + 16
//  bytes — in the `H0` buffer, which in lean mode must hold n_edges * max(4 ldh, that stride) bytes: the host allocates it so)
static bool x_path_shapes(const dmpnn_fwd_args& a) {
    return a.d_h <= 320 && (a.d_v + a.d_e + 31) / 32 <= step16::kXChunks;
}
bool fused16_lean_shapes(const dmpnn_fwd_args& a) {
    const unsigned need = DMPNN_F_FUSED | DMPNN_F_SPLIT16;
    if ((a.flags & need) != need || (a.flags & (DMPNN_F_MEGA | DMPNN_F_UNDIRECTED | DMPNN_F_STORE16 | DMPNN_F_ATOM))) return false;
    if (!(a.act == DMPNN_ACT_NONE || a.act == DMPNN_ACT_RELU || a.act == DMPNN_ACT_LEAKYRELU)) return false;
    // (depth <= kWProdMaxJobs: the backward pass of this route forms every weight gradient as ONE launch of at most that many product
    //  jobs per matrix, dmpnn_backward.hip — a deeper block keeps the fp32 tensors and trains on the fused16 route as before)
    if (a.W_d || a.dropout_p > 0.f || a.depth < 2 || a.depth > kWProdMaxJobs || a.n_edges <= 0 || a.n_atoms <= 0) return false;
    if (a.d_h <= 0 || a.d_h % 4 != 0 || a.ldh % 4 != 0 || a.d_v % 2 || a.d_e % 2 || a.ldv % 2 || a.lde % 2) return false;
    return x_path_shapes(a);
}
size_t fused16_lean_bits_bytes(const dmpnn_fwd_args& a) {
    return fused16_lean_shapes(a) ? (size_t)a.depth * (size_t)a.n_edges * (size_t)(step16::block_cols((int)a.d_h) / 8) : 0;
}
bool fused16_lean(const dmpnn_fwd_args& a) {
    if (!(a.flags & DMPNN_F_KEEP) || !a.keep_bits || !fused16_lean_shapes(a)) return false;
    if (a.keep_bits_bytes < fused16_lean_bits_bytes(a)) return false;
    if (!a.msplit || a.msplit_bytes < (size_t)(a.depth - 1) * (size_t)a.n_edges * step16::split_row_bytes((int)a.d_h)) return false;
    return !(reinterpret_cast<uintptr_t>(a.msplit) & 15u) && a.H0 && !(reinterpret_cast<uintptr_t>(a.H0) & 15u);
}

bool fused16_shapes_ok(const dmpnn_fwd_args& a) {
    const int64_t h = a.d_h;
    if (a.flags & DMPNN_F_UNDIRECTED) return false;  // directed graphs
    if ((a.flags & DMPNN_F_KEEP) && a.keep_bits) return fused16_lean(a) && a.n_atoms * a.ldv * 4 <= 0x7FFFFFFF && a.n_edges * a.lde * 4 <= 0x7FFFFFFF;
    if (a.flags & DMPNN_F_KEEP) {
        // training: the kept fp32 tensors the backward pass reads + the two split ping-pong slots in `msplit`
        if (a.flags & DMPNN_F_STORE16) return false;
        if (a.n_edges > 0 && a.depth > 1 && (!a.Hs || a.n_hslots < a.depth - 1 || !a.Ms || a.n_mslots < a.depth - 1)) return false;
        if (a.n_edges > 0 && (!a.msplit || a.msplit_bytes < 2 * (size_t)a.n_edges * step16::split_row_bytes((int)a.d_h))) return false;
        if ((reinterpret_cast<uintptr_t>(a.msplit) & 15u) || (a.Hs && (reinterpret_cast<uintptr_t>(a.Hs) & 15u)) || (reinterpret_cast<uintptr_t>(a.Ms) & 15u)) return false;
    }
    if (h <= 0 || h % 4 != 0 || h > 640 || a.ldh % 4 != 0) return false;
    if (h > 320 && step16::split_operand_bytes((int)(a.d_v + a.d_e)) > step16::split_row_bytes((int)h)) return false;
    if (a.d_v % 2 || a.d_e % 2 || a.ldv % 2 || a.lde % 2) return false;
    if (a.n_atoms * a.ldv * 4 > 0x7FFFFFFF || a.n_edges * a.lde * 4 > 0x7FFFFFFF) return false;
    if ((a.n_edges + 64) * (int64_t)step16::split_row_bytes((int)h) > ((int64_t)1 << 40)) return false;
    return true;
}

static unsigned qmagic_of(int64_t N) {
    const unsigned qn = (unsigned)(N / 4);
    return qn > 1 ? (unsigned)(((1ull << 32) + qn - 1) / qn) : 0u;
}

// K1 with the segment epilogue: H0 = W_i [V[srcp] || E[perm]] (+ b_i) stored; tau; first message (split rows) or Mv
static int launch_k1_seg(const dmpnn_fwd_args& a, const PlanLayout& L, const SplitWView& W, unsigned char* Mout, float* Sout, float* M32, hipStream_t s) {
    const int* plan_i = static_cast<const int*>(a.plan);
    rows16::Rows16K g;
    memset(&g, 0, sizeof(g));
    g.M = (int)a.n_edges; g.N = (int)a.d_h; g.K1 = (int)a.d_v; g.K2 = (int)a.d_e;
    g.A1 = a.V; g.lda1 = (int)a.ldv; g.gather1 = plan_i + L.srcp; g.a1_bytes = (unsigned)(a.n_atoms * a.ldv * 4);
    g.A2 = a.d_e ? a.E : nullptr; g.lda2 = (int)a.lde; g.gather2 = plan_i + L.perm; g.a2_bytes = (unsigned)(a.n_edges * a.lde * 4);
    g.W.p = W.p; g.W.inv_scale = W.inv_scale; g.W.nc = W.nc;
    g.bias = a.b_i;
    g.Zpre = a.H0; g.ldz = (int)a.ldh;
    g.act = a.act; g.slope = a.act_slope; g.slope_ptr = a.act_slope_ptr;
    g.poison_flags = plan_i + DMPNN_HDR_FLAGS; g.poison_mask = kPlanNoFuse;
    g.vec_out = 1;
    g.tile_row = plan_i + L.tile_row; g.tile_atom = plan_i + L.tile_atom; g.row_ptr = plan_i + L.row_ptr; g.revp = plan_i + L.revp;
    g.Mout = Mout; g.ts = msg_row_bytes(a); g.half_out = half_store(a) ? 1 : 0; g.Sout = Sout; g.lds = (int)a.ldh; g.qmagic = qmagic_of(a.d_h);
    g.M32 = Mout ? M32 : nullptr; g.ldm32 = (int)a.ldh;
    const int n_tiles = (int)L.max_tiles;
    switch ((int)((a.d_h + 63) / 64)) {
        case 1: return rows16::launch_rows16<1, 4, true>(g, n_tiles, 1, s);
        case 2: return rows16::launch_rows16<2, 4, true>(g, n_tiles, 1, s);
        case 3: return rows16::launch_rows16<3, 4, true>(g, n_tiles, 1, s);
        case 4: return rows16::launch_rows16<4, 4, true>(g, n_tiles, 1, s);
        default: return rows16::launch_rows16<5, 4, true>(g, n_tiles, 1, s);
    }
}

// hin: the operand rows (g.A, g.ts) are in half storage
static int launch_step(const step16::Step16K& g0, int64_t d_h, int n_tiles, bool hin, hipStream_t s) {
    step16::Step16K g = g0;
    const int bn = step16::block_cols((int)d_h);
    if (!g.A2) g.ts2 = 0;
    const size_t tile_a = g.A ? (size_t)step16::BM * g.ts : 0, tile_t = (size_t)step16::BM * (bn + 4) * 4;
    g.tile_bytes = (int)(((tile_a > tile_t ? tile_a : tile_t) + 15) & ~size_t(15));
    const bool xp = g.A2 != nullptr;
#define DMPNN_STEP(WN, NW) (xp ? (hin ? step16::launch_step16<WN, NW, true, true>(g, n_tiles, s) : step16::launch_step16<WN, NW, false, true>(g, n_tiles, s)) \
                               : (hin ? step16::launch_step16<WN, NW, true, false>(g, n_tiles, s) : step16::launch_step16<WN, NW, false, false>(g, n_tiles, s)))
    if (d_h <= 320) {
        switch (bn / 64) {
            case 1: return DMPNN_STEP(1, 4);
            case 2: return DMPNN_STEP(2, 4);
            case 3: return DMPNN_STEP(3, 4);
            case 4: return DMPNN_STEP(4, 4);
            default: return DMPNN_STEP(5, 4);
        }
    }
    switch (bn / 128) {
        case 3: return DMPNN_STEP(3, 8);
        case 4: return DMPNN_STEP(4, 8);
        default: return DMPNN_STEP(5, 8);
    }
#undef DMPNN_STEP
}

static step16::Step16K step_args(const dmpnn_fwd_args& a, const PlanLayout& L) {
    const int* plan_i = static_cast<const int*>(a.plan);
    step16::Step16K g;
    memset(&g, 0, sizeof(g));
    g.M = (int)a.n_edges; g.N = (int)a.d_h;
    g.tile_row = plan_i + L.tile_row; g.tile_atom = plan_i + L.tile_atom; g.row_ptr = plan_i + L.row_ptr; g.revp = plan_i + L.revp;
    g.lds = (int)a.ldh;
    g.act = a.act; g.slope = a.act_slope; g.slope_ptr = a.act_slope_ptr;
    g.poison_flags = plan_i + DMPNN_HDR_FLAGS; g.poison_mask = kPlanNoFuse;
    g.qmagic = qmagic_of(a.d_h);
    g.dbg = g_debug_stamps;
    return g;
}

// `xrows` (or null): the K1 operand [V[src] || E] of every row, exactly split (k_split_rows) — the residual H0 = W_i x + b_i is
// then recomputed inside the step instead of read back (x_path_ok)
static int launch_update(const dmpnn_fwd_args& a, const PlanLayout& L, const SplitWView& W, const SplitWView* Wi, const unsigned char* xrows,
                         const unsigned char* Min, unsigned char* Mout, float* Sout, unsigned char* SoutS, float* Hout, float* M32, hipStream_t s,
                         unsigned char* bits = nullptr, const float* h0q = nullptr) {
    step16::Step16K g = step_args(a, L);
    g.bits = bits; g.bstride = step16::block_cols((int)a.d_h) / 8;
    g.A = Min; g.ts = msg_row_bytes(a);
    g.W.p = W.p; g.W.inv_scale = W.inv_scale; g.W.nc = W.nc;
    g.bias = a.b_h;
    if (xrows) {
        g.A2 = xrows; g.ts2 = step16::split_operand_bytes((int)(a.d_v + a.d_e));
        g.W2.p = Wi->p; g.W2.inv_scale = Wi->inv_scale; g.W2.nc = Wi->nc; g.bias2 = a.b_i;
    } else if (h0q) {
        g.H0q_in = h0q;      // H0 as row quads (dmpnn_fwd_args.h0_bytes)
    } else {
        g.Cadd = a.H0; g.ldcadd = (int)a.ldh;
    }
    g.Mout = Mout; g.Sout = Sout; g.SoutS = SoutS; g.half_out = half_store(a) ? 1 : 0;
    g.Hout = Hout; g.ldho = (int)a.ldh; g.M32 = Mout ? M32 : nullptr; g.ldm32 = (int)a.ldh;
    return launch_step(g, a.d_h, (int)L.max_tiles, half_store(a), s);
}

// The step kernel with the K1 operand as a second operand held in registers: its rows must fit the H0 buffer they are kept
// in (which holds no H0 then) and their chunks the lane's fragment registers.  DMPNN_F_H0_RESIDUAL: off (H0 written and read back).
// Measured (MI355X, same box): 40-atom x 4096 molecules 1 743 -> 1 669 us, CGR-512 305 -> 299, 40-atom x 512 345 -> 326; ZINC-512
// h 512 depth 6 (the 8-wave workgroups) 616 -> 649 — so it is the rule for d_h <= 320 only.
static bool x_path_ok(const dmpnn_fwd_args& a) {
    if ((a.flags & DMPNN_F_H0_RESIDUAL) || a.depth < 2 || a.d_h > 320) return false;
    const int ts2 = step16::split_operand_bytes((int)(a.d_v + a.d_e));
    return (int64_t)ts2 <= a.ldh * 4 && (a.d_v + a.d_e + 31) / 32 <= step16::kXChunks;
}

// K1 on the update kernel (d_h > 320, and the x path): the gathered fp32 operand is split into rows first (`scratch`: the second
// message slot, or — x path, keep_h0 false — the H0 buffer, where the rows stay for the depth steps and no H0 is written)
static int launch_k1_split(const dmpnn_fwd_args& a, const PlanLayout& L, const SplitWView& W, unsigned char* scratch, bool keep_h0,
                           unsigned char* Mout, float* Sout, float* M32, hipStream_t s, unsigned char* bits = nullptr,
                           const mega16::SplitArgs** pending = nullptr, float* h0q_out = nullptr) {
    const int* plan_i = static_cast<const int*>(a.plan);
    step16::SplitRowsK k;
    memset(&k, 0, sizeof(k));
    k.tile_row = plan_i + L.tile_row; k.n_tiles = (int)L.max_tiles;
    k.A1 = a.V; k.lda1 = (int)a.ldv; k.g1 = plan_i + L.srcp; k.K1 = (int)a.d_v; k.a1_bytes = (unsigned)(a.n_atoms * a.ldv * 4);
    k.A2 = a.d_e ? a.E : nullptr; k.lda2 = (int)a.lde; k.g2 = plan_i + L.perm; k.K2 = (int)a.d_e; k.a2_bytes = (unsigned)(a.n_edges * a.lde * 4);
    k.out = scratch; k.ts = step16::split_operand_bytes((int)(a.d_v + a.d_e));
    if (pending && *pending) {   // the weights' pre-split rides in this launch (which does not read them)
        const mega16::SplitArgs& sp = **pending;
        const unsigned waves = (unsigned)(((sp.N + 15) / 16) * 16) * (unsigned)sp.n_jobs;
        hipLaunchKernelGGL(step16::k_split_rows_w, dim3((unsigned)L.max_tiles + (waves + 3) / 4), dim3(256), 0, s, k, sp, (int)L.max_tiles);
        DMPNN_CHECK_LAUNCH("k_split_rows_w");
        *pending = nullptr;
    } else {
        hipLaunchKernelGGL(step16::k_split_rows, dim3((unsigned)L.max_tiles), dim3(256), 0, s, k);
        DMPNN_CHECK_LAUNCH("k_split_rows");
    }
    step16::Step16K g = step_args(a, L);
    g.A = scratch; g.ts = k.ts;
    g.W.p = W.p; g.W.inv_scale = W.inv_scale; g.W.nc = W.nc;
    g.bias = a.b_i;
    if (keep_h0) { g.Zpre = a.H0; g.ldz = (int)a.ldh; }
    g.H0q_out = h0q_out;
    g.Mout = Mout; g.Sout = Sout; g.half_out = half_store(a) ? 1 : 0;
    g.M32 = Mout ? M32 : nullptr; g.ldm32 = (int)a.ldh;
    g.bits = bits; g.bstride = step16::block_cols((int)a.d_h) / 8;
    return launch_step(g, a.d_h, (int)L.max_tiles, false, s);  // (the K1 operand [V || E] is always split exactly)
}

// The finalize on the step kernel: out = tau(W_o[:, d_v:] Mv + W_o[:, :d_v] V + b_o) over uniform 48-atom tiles — the last depth
// step leaves Mv as split rows (in the message slot it does not read), V is split once into 400-byte rows (where the fp32 Mv
// would have been), the atoms' rows arrive by LDS-DMA like any operand tile.  Replaces the row kernel (k_rows16: three
// dependent load -> maximum -> split -> contract groups per tile, 18 % of a large forward).  DMPNN_F_ROW_FINALIZE: off.
static bool fin16_ok(const dmpnn_fwd_args& a, const float* out, int64_t ldout) {
    if ((a.flags & DMPNN_F_ROW_FINALIZE) || a.depth < 2 || a.n_edges <= 0 || a.n_atoms > a.n_edges) return false;
    if ((a.d_v + 31) / 32 > step16::kXChunks || (int64_t)step16::split_operand_bytes((int)a.d_v) > a.ldh * 4) return false;
    return ldout % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0 && a.d_h % 4 == 0;
}

static int launch_fin16(const dmpnn_fwd_args& a, const PlanLayout& L, const SplitWView& WoM, const SplitWView& WoV, const unsigned char* MvS,
                        float* out, int64_t ldout, hipStream_t s) {
    const int* plan_i = static_cast<const int*>(a.plan);
    unsigned char* VS = reinterpret_cast<unsigned char*>(a.Mv);  // (no fp32 Mv on this path)
    const int n_tiles = (int)((a.n_atoms + step16::BM - 1) / step16::BM);
    step16::SplitRowsK k;
    memset(&k, 0, sizeof(k));
    k.n_tiles = n_tiles; k.n_rows = (int)a.n_atoms;
    k.A1 = a.V; k.lda1 = (int)a.ldv; k.K1 = (int)a.d_v; k.a1_bytes = (unsigned)(a.n_atoms * a.ldv * 4);
    k.out = VS; k.ts = step16::split_operand_bytes((int)a.d_v);
    hipLaunchKernelGGL(step16::k_split_rows, dim3((unsigned)n_tiles), dim3(256), 0, s, k);
    DMPNN_CHECK_LAUNCH("k_split_rows");
    step16::Step16K g;
    memset(&g, 0, sizeof(g));
    g.M = (int)a.n_atoms; g.N = (int)a.d_h; g.uniform = 1;
    g.A = MvS; g.ts = step16::split_row_bytes((int)a.d_h);
    g.W.p = WoM.p; g.W.inv_scale = WoM.inv_scale; g.W.nc = WoM.nc; g.bias = a.b_o;
    g.A2 = VS; g.ts2 = k.ts; g.W2.p = WoV.p; g.W2.inv_scale = WoV.inv_scale; g.W2.nc = WoV.nc;
    g.Yout = out; g.ldy = (int)ldout;
    g.act = a.act; g.slope = a.act_slope; g.slope_ptr = a.act_slope_ptr;
    g.poison_flags = plan_i + DMPNN_HDR_FLAGS; g.poison_mask = kPlanNoFuse;
    g.qmagic = qmagic_of(a.d_h);
    g.dbg = nullptr;
    (void)L;
    return launch_step(g, a.d_h, n_tiles, false, s);
}

static const int64_t kH0QuadsMaxEdges = 131072;
// H0 as row quads (dmpnn_fwd_args.h0_bytes): bytes of the buffer — every tile owns ceil(nrows / 4) quads from ((first row + 3) >> 2) + tile on
size_t fused16_h0q_bytes(const dmpnn_fwd_args& a) {
    const unsigned need = DMPNN_F_FUSED | DMPNN_F_SPLIT16;
    // (DMPNN_F_H0_RESIDUAL asks for H0 as fp32 ROWS in the buffer — what tests / diagnostics read back: not this form)
    if ((a.flags & need) != need || (a.flags & (DMPNN_F_MEGA | DMPNN_F_KEEP | DMPNN_F_UNDIRECTED | DMPNN_F_ATOM | DMPNN_F_H0_RESIDUAL))) return 0;
    if (a.depth < 2 || a.n_edges <= 0 || a.n_atoms <= 0 || !fused16_shapes_ok(a)) return 0;
    // K1 runs on the step kernel over the split K1 operand, whose rows live in the second message slot meanwhile: they must fit there
    if (step16::split_operand_bytes((int)(a.d_v + a.d_e)) > step16::split_row_bytes((int)a.d_h)) return 0;
    // a SIZE rule (measured, profiles/r05_h0_quads_ab.txt): the quads cost 58 KB of reads per 48-row tile and step where the x path costs
    // 19 KB + 135 MFMAs — 3-7 % faster at 20 k .. 44 k directed edges (BASELINE configs 2 / 3 at 512 molecules / 4), 1.4 % slower at 356 k
    // (configs 3 at 4 096 molecules per GPU, where the launch runs at 2.8 TB/s already)
    if (a.n_edges > kH0QuadsMaxEdges) return 0;
    const PlanLayout L = plan_layout(a.n_atoms, a.n_edges);
    return (size_t)(a.n_edges / 4 + L.max_tiles + 4) * (size_t)step16::block_cols((int)a.d_h) * 16u;
}

// a.Ms: two slots of n_edges split rows (split_row_floats(d_h) floats each); a.H0 [n_edges, ldh]; a.Mv [n_atoms, ldh];
// w16: pre-split W_i | W_h | W_o (| W_d).  `out` / `ldout`: the finalize output (Hv when W_d follows).
int launch_fused16_forward(const dmpnn_fwd_args& a, const SplitWView* w16, float* out, int64_t ldout, hipStream_t s,
                           const mega16::SplitArgs* pending_in) {
    const int64_t nV = a.n_atoms, nE = a.n_edges, h = a.d_h;
    const mega16::SplitArgs* pending = pending_in;
    // the weights' pre-split: in the first launch of the chain when that launch is k_split_rows (it does not read them), else now
    // H0 as row quads (ABI 12): an inference forward whose H0 buffer is large enough keeps the residual in the fragments' own layout
    const size_t h0q_need = fused16_h0q_bytes(a);
    const bool h0q = h0q_need > 0 && a.h0_bytes >= h0q_need && a.H0 && !(reinterpret_cast<uintptr_t>(a.H0) & 15u);
    const bool rides = nE > 0 && (fused16_lean(a) || h0q || (!(a.flags & DMPNN_F_KEEP) && x_path_ok(a)) || h > 320);
    if (pending && !rides) { DMPNN_TRY(launch_split_args(*pending, s)); pending = nullptr; }
    const PlanLayout L = plan_layout(nV, nE);
    const int T = a.depth;
    // (a slot is n_edges message rows; the first always spans split_row_bytes per edge — the K1 operand scratch of wide layers lives in the second)
    const size_t slot_bytes = (size_t)nE * step16::split_row_bytes((int)h);
    // training (DMPNN_F_KEEP): the split ping-pong slots live in `msplit`; H0 / Hs / Ms / Mv are the fp32 tensors the backward reads
    const bool keep = (a.flags & DMPNN_F_KEEP) != 0;
    unsigned char* Ms = reinterpret_cast<unsigned char*>(keep ? a.msplit : a.Ms);
    const int64_t slot32 = nE * a.ldh;
    if (nE == 0 && nV > 0) {
        hipError_t e = hipMemsetAsync(a.Mv, 0, (size_t)nV * a.ldh * sizeof(float), s);
        if (e != hipSuccess) { set_error("forward(fused16): memset failed: %s", hipGetErrorString(e)); return DMPNN_EHIP; }
    }
    // (training keeps H0 and an fp32 Mv — the residual is read back, the finalize runs on the row kernel)
    const bool fin16 = !keep && fin16_ok(a, out, ldout);
    if (nE > 0 && fused16_lean(a)) {
        // ---- lean training forward: split rows of every step, sign bits, no fp32 copies (see fused16_lean_shapes) ----
        unsigned char* xrows = reinterpret_cast<unsigned char*>(a.H0);
        unsigned char* Mk = reinterpret_cast<unsigned char*>(a.msplit);
        unsigned char* bits = static_cast<unsigned char*>(a.keep_bits);
        const size_t bslot = (size_t)nE * (size_t)(step16::block_cols((int)h) / 8);
        DMPNN_TRY(launch_k1_split(a, L, w16[0], xrows, false, Mk, nullptr, nullptr, s, bits, &pending));   // M^(1) -> slot 0; signs of H0 -> site 0
        for (int t = 1; t < T; ++t) {
            const bool last = t == T - 1;
            DMPNN_TRY(launch_update(a, L, w16[1], &w16[0], xrows, Mk + (size_t)(t - 1) * slot_bytes, last ? nullptr : Mk + (size_t)t * slot_bytes,
                                    last ? a.Mv : nullptr, nullptr, nullptr, nullptr, s, bits + (size_t)t * bslot));
        }
    } else if (nE > 0 && h0q) {
        // ---- inference with H0 kept as row quads: K1 on the step kernel over the split K1 operand (scratch: the second message slot,
        // dead before update 1 writes there) leaves H0 in the fragments' layout; every update reads it back coalesced ----
        float* H0q = a.H0;
        DMPNN_TRY(launch_k1_split(a, L, w16[0], Ms + slot_bytes, false, Ms, nullptr, nullptr, s, nullptr, &pending, H0q));
        for (int t = 1; t < T; ++t) {
            const bool last = t == T - 1;
            unsigned char* free_slot = Ms + (t % 2) * slot_bytes;
            DMPNN_TRY(launch_update(a, L, w16[1], &w16[0], nullptr, Ms + ((t - 1) % 2) * slot_bytes, last ? nullptr : free_slot,
                                    (last && !fin16) ? a.Mv : nullptr, (last && fin16) ? free_slot : nullptr, nullptr, nullptr, s, nullptr, H0q));
        }
        if (fin16) return launch_fin16(a, L, w16[4], w16[5], Ms + ((T - 1) % 2) * slot_bytes, out, ldout, s);
    } else if (nE > 0) {
        const bool xpath = !keep && x_path_ok(a);
        unsigned char* xrows = xpath ? reinterpret_cast<unsigned char*>(a.H0) : nullptr;
        float* m32_0 = (keep && T > 1) ? a.Ms : nullptr;  // M^(1): what update step 1 consumes, what gW_h's first product reads
        if (xpath) DMPNN_TRY(launch_k1_split(a, L, w16[0], xrows, false, T > 1 ? Ms : nullptr, T > 1 ? nullptr : a.Mv, nullptr, s, nullptr, &pending));
        else if (h > 320) DMPNN_TRY(launch_k1_split(a, L, w16[0], Ms + slot_bytes, true, T > 1 ? Ms : nullptr, T > 1 ? nullptr : a.Mv, m32_0, s, nullptr, &pending));
        else DMPNN_TRY(launch_k1_seg(a, L, w16[0], T > 1 ? Ms : nullptr, T > 1 ? nullptr : a.Mv, m32_0, s));
        for (int t = 1; t < T; ++t) {
            const bool last = t == T - 1;
            unsigned char* free_slot = Ms + (t % 2) * slot_bytes;  // (the slot this step does not read)
            DMPNN_TRY(launch_update(a, L, w16[1], &w16[0], xrows, Ms + ((t - 1) % 2) * slot_bytes, last ? nullptr : free_slot,
                                    (last && !fin16) ? a.Mv : nullptr, (last && fin16) ? free_slot : nullptr,
                                    keep ? a.Hs + (int64_t)(t - 1) * slot32 : nullptr, (keep && !last) ? a.Ms + (int64_t)t * slot32 : nullptr, s));
        }
        if (fin16) return launch_fin16(a, L, w16[4], w16[5], Ms + ((T - 1) % 2) * slot_bytes, out, ldout, s);
    }
    dmpnn_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.M = nV; g.N = h; g.K1 = a.d_v; g.K2 = h;
    g.A1 = a.V; g.lda1 = a.ldv; g.A2 = a.Mv; g.lda2 = a.ldh;
    g.W = a.W_o; g.ldw = a.d_v + h; g.bias = a.b_o;
    g.C = out; g.ldc = ldout;
    g.act = a.act; g.act_slope = a.act_slope; g.act_slope_ptr = a.act_slope_ptr;
    const int* plan_i = static_cast<const int*>(a.plan);
    if (linear16_ok(g)) return launch_linear16_view(g, w16[2], plan_i + DMPNN_HDR_FLAGS, kPlanNoFuse, s);
    GemmExtra xf;
    memset(&xf, 0, sizeof(xf));
    xf.poison_flags = plan_i + DMPNN_HDR_FLAGS; xf.poison_mask = kPlanNoFuse;
    return launch_linear_ex(g, xf, s);
}

}  // namespace dmpnn

extern "C" int64_t dmpnn_split_row_floats(int64_t d_h) { return dmpnn::split_row_floats(d_h); }

extern "C" int dmpnn_forward_can_fuse16(const dmpnn_fwd_args* a) { return (a && dmpnn::fused16_shapes_ok(*a)) ? 1 : 0; }
