// The data-gradient chain of BondMessagePassing's backward for one tile of whole molecules in ONE launch — the mirror
// of k_mpnn_tile16 (dmpnn_mega16_impl.hpp), same tiles, same arithmetic (f16 pipe, exact 3-term operand split):
//
//   gZO   = gHO * tau'(HO)                                    finalize (base.py:180-194), atoms            -> stored
//   gMv   = gZO . W_o[:, d_v:]                                contraction with the transposed pre-split
//   gH    = gMv[dst(r)]                                       aggregation backward (base.py:208-211): incidence MFMA
//   for t = T-1 .. 1:
//       gZt = gH * tau'(H^(t));  gH0 += gZt                   update backward (base.py:135-141)             -> gZt stored
//       gM  = gZt . W_h                                       contraction
//       gH  = C^T gM   (gH[r'] = sum_{r: src r = dst r'} gM[r] - gM[rev r'])   message backward (mixins.py:11-18)
//   gH0  += gH * tau'(tau(H0))                                                                              -> stored
//
// What is left for separate launches are the four weight gradients (reductions over ALL edges): gZO with [V || Mv],
// every gZt with M^(t-1), gH0 with [V[src] || E].  Replaces act_bwd + gMv contraction + gather + (depth-1) x (gM
// contraction + message backward): nine launches of 4..30 us at 512 molecules.
#pragma once

#include <type_traits>

#include "dmpnn_mega16_impl.hpp"

namespace dmpnn {
namespace mega16 {

struct Mega16BwdK {
    const int* mtile_row; const int* mtile_atom; const int* row_ptr; const int* revp;
    const int* flags; int poison_mask;
    int nV, nE, h, depth;
    int act; float slope; const float* slope_ptr;
    const float* gHO; int ldg;           // [V, ldg]  gradient w.r.t. the finalize output
    const float* HO; int ldho;           // [V, ldho] finalize output (tau applied)
    const float* H0; const float* Hs; int ldh; long long slot;  // kept by the forward: H0 (pre-activation), Hs[t-1] = H^(t)
    float* gZO;                          // [V, ldh]
    float* gZs;                          // [(depth-1)][E, ldh]: slot t-1 = gZ^(t)
    float* gH0;                          // [E, ldh]
    SplitW WoMT, WhT;                    // pre-split W_o[:, d_v:]^T and W_h^T  ([h, h] each)
    // the generic path for pieces larger than the tile (dmpnn_spill_impl.hpp): plain weights, two scratch tensors
    const int* srcp; int d_v; const float* W_o; const float* W_h;
    float* sp_gM; float* sp_Ta;          // [E, ldh], [V, ldh]
    float drop_scale;                    // active dropout in the forward: 1 / (1 - p) (else 0).  The kept H^(t) and the finalize output
                                         // are POST-dropout; for a ReLU-class activation their sign carries the mask (0: dropped or inactive)
    // tile plan (header LIGHT == 2, dmpnn_prepare_tiles): the forward kept its tensors in the CALLER's edge order and the rows of a
    // tile are its edges in that order — src / dst / rev straight from the caller's arrays, no CSR tables (row_ptr / revp / srcp unused)
    const long long* edge_index; const long long* rev64;
    // the forward kept H0 / H^(t) as sign bits (Mega16K::keep_bits: slot 0 = H0, slot t = H^(t), bits_slot words per slot) — the
    // fp32 rows H0 / Hs then only hold the molecules beyond the tile
    const unsigned long long* keep_bits; long long bits_slot;
    // DMPNN_F_ATOM (mixins.py:21-30): the message is the plain sum over the edges entering the source atom — its transpose has no
    // reverse-edge term, gH[r'] = sum_{r: src r = dst r'} gM[r]; W_h's first h columns are what WhT holds
    int atom;
    // a second gradient input (dmpnn_bwd_args.g_edge): dL/dH^(depth-1) from a consumer of the kept edge states themselves — the edge
    // read-out of the mol-atom-bond blocks (mol_atom_bond.py:221-264) — added to the aggregation's gradient before tau'.  [E, ld_ge]
    // in the kept tensors' row order, 16-byte aligned rows; NULL: none
    const float* g_edge; int ld_ge;
    // Round 4: gZ^(t) (slot t - 1, zrow_slot bytes apart), gH0 and gZO leave as SPLIT ROWS of `tsr` bytes — the pieces this kernel stages
    // for its own contractions, with the tile's scale in the row tails — the operands of the weight-gradient product on split rows
    // (k_wgrad16r); the fp32 tensors gZs / gH0 / gZO are then written by the generic path of a molecule beyond the tile only (which
    // converts its rows at the end).  null: fp32 rows as before
    unsigned char* gZrows; unsigned char* gH0rows; unsigned char* gZOrows; int tsr; long long zrow_slot;
};

template <int WN>
constexpr size_t bwd_lds_bytes() {
    return (size_t)kMegaBM * (64 * WN * 4 + 16) + (size_t)(3 * kMegaBM + kMegaBA + 24) * sizeof(int) + 10 * 64 * 16;
}

// SA: identity / ReLU / LeakyReLU (mask from the sign of the output: compare + select); tanh / ELU — whose derivative code
// would otherwise be inlined per fragment element at every call site — get their own instantiation (code size is
// instruction-fetch latency for a kernel that runs its code once per tile).
// NW: waves per workgroup, as in the forward (dmpnn_mega16_impl.hpp): 8 = one tile as a 512-thread workgroup whose waves own 3+3+3+3+2+2+2+2
// column tiles (d_h in (128, 320]), for launches of at most one tile per CU; each wave class runs its own instantiation of the body.
template <int WN, bool SA, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void k_mpnn_tile16_bwd(Mega16BwdK g) {
    static_assert(NW == 4 || (NW == 8 && WN == 5), "the 8-wave form splits 20 column tiles 3+3+3+3+2+2+2+2");
    constexpr int KT = 64 * NW;
    constexpr int BM = kMegaBM, BA = kMegaBA, BN = 64 * WN, QN = BN / 4;
    constexpr int TS = BN * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* T16 = lds;                                          // [BM][TS] split A operand
    int* revl = reinterpret_cast<int*>(lds + BM * TS);                 // [BM]
    int* aor = revl + BM;                                              // [BM] destination atom of a row
    int* asrc = aor + BM;                                           // [BM] source atom of a row
    int* rp = asrc + BM;                                                // [BA + 1]
    unsigned* maxbits = reinterpret_cast<unsigned*>(rp + BA + 1);        // [0..3] rotating tile maxima
    h8* cfrag = reinterpret_cast<h8*>(lds + BM * TS + (((3 * BM + BA + 1 + 8) * 4 + 15) / 16) * 16);  // [9][64]

#if !defined(DMPNN_NO_KERNARG_WARM)
    warm_kernargs<(int)sizeof(Mega16BwdK)>();
#endif
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int li = lane & 15, lg = lane >> 4;
    auto launder = [&]() {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wave = tid >> 6; li = lane & 15; lg = lane >> 4;
    };
    const int t = blockIdx.x;
    const int rs = g.mtile_row[t], re = g.mtile_row[t + 1];
    const int va = g.mtile_atom[t], vb = g.mtile_atom[t + 1];
    // (the forward kernel forces these four words and the two header words into ONE round trip; the same six lines made this kernel
    //  fault on the GPU at 230 tiles — both wave counts, not at 29 tiles — and bought nothing measurable in the forward: left alone here)
    const int hdr_light = g.flags[DMPNN_HDR_LIGHT], hdr_flags = g.flags[0], hdr_tiles = g.flags[DMPNN_HDR_NMTILES];
    const int nrows = re - rs, na = vb - va;
    const int N = g.h, qn = N >> 2;
    const int T_steps = g.depth;
    const float nanv = __int_as_float(0x7fc00000);
    const bool lean = hdr_light == 2;
    if ((hdr_flags & (lean ? kPlanNoMegaLean : g.poison_mask)) != 0 || (lean && g.nE > 0 && (!g.edge_index || !g.rev64)) ||
        hdr_tiles > (int)gridDim.x) {  // a graph this route cannot represent, or a launch with fewer workgroups than the plan has tiles: every output NaN
        const long long tot_e = (long long)g.nE * N, tot_v = (long long)g.nV * N;
        for (long long i = (long long)blockIdx.x * KT + tid; i < tot_e; i += (long long)gridDim.x * KT) {
            g.gH0[(i / N) * g.ldh + (i % N)] = nanv;
            for (int s = 0; s < T_steps - 1; ++s) g.gZs[(long long)s * g.slot + (i / N) * g.ldh + (i % N)] = nanv;
        }
        for (long long i = (long long)blockIdx.x * KT + tid; i < tot_v; i += (long long)gridDim.x * KT)
            g.gZO[(i / N) * g.ldh + (i % N)] = nanv;
        return;
    }
    if (na <= 0 || nrows < 0) return;
    if (nrows > BM || na > BA) {  // a piece larger than the matrix-pipe tile: the generic fp32 path, any size
        if (NW > 4 && threadIdx.x >= kThreads) return;  // (written for 256 threads; s_barrier counts live waves only)
        const Mega16BwdK& g = *spill::fresh_kernargs<Mega16BwdK>();  // (shadows the hot path's copy: see fresh_kernargs)
        if (g.atom) {  // ... which knows bond messages only: the forward tile kernel returned NaN for this molecule, so do its gradients
            for (int i = tid; i < nrows * N; i += kThreads) {
                const long long o = (long long)(rs + i / N) * g.ldh + (i % N);
                g.gH0[o] = nanv;
                for (int sl = 0; sl < T_steps - 1; ++sl) g.gZs[(long long)sl * g.slot + o] = nanv;
            }
            for (int i = tid; i < na * N; i += kThreads) g.gZO[(long long)(va + i / N) * g.ldh + (i % N)] = nanv;
            return;
        }
        const float slope = g.slope_ptr ? *g.slope_ptr : g.slope;
        spill::BwdView v;
        v.rs = rs; v.nrows = nrows; v.va = va; v.na = na; v.h = N; v.depth = T_steps; v.d_v = g.d_v;
        v.row_ptr = g.row_ptr; v.srcp = g.srcp; v.revp = g.revp;
        v.lean = lean; v.edge_index = g.edge_index; v.rev64 = g.rev64; v.nE = g.nE;
        v.act = g.act; v.slope = slope;
        v.gHO = g.gHO; v.ldg = g.ldg; v.HO = g.HO; v.ldho = g.ldho;
        v.H0 = g.H0; v.Hs = g.Hs; v.ldh = g.ldh; v.slot = g.slot;
        v.gZO = g.gZO; v.gZs = g.gZs; v.gH0 = g.gH0;
        v.W_o = g.W_o; v.W_h = g.W_h; v.gM = g.sp_gM; v.Ta = g.sp_Ta;
        v.g_edge = g.g_edge; v.ld_ge = g.ld_ge;
        spill::backward(v, reinterpret_cast<float*>(lds));
        if (g.gZrows) {   // this molecule's gradients, fp32 rows, also as the split rows the products read
            __threadfence_block();
            __syncthreads();
            const int w_ = (int)(threadIdx.x >> 6), l_ = (int)(threadIdx.x & 63);
            for (int sl = 0; sl < T_steps - 1; ++sl)
                rows_to_sr(g.gZs + (long long)sl * g.slot, g.ldh, rs, nrows, N, g.gZrows + (long long)sl * g.zrow_slot, g.tsr, w_, l_, kThreads / 64);
            rows_to_sr(g.gH0, g.ldh, rs, nrows, N, g.gH0rows, g.tsr, w_, l_, kThreads / 64);
            rows_to_sr(g.gZO, g.ldh, va, na, N, g.gZOrows, g.tsr, w_, l_, kThreads / 64);
        }
        return;
    }
    const float slope = g.slope_ptr ? *g.slope_ptr : g.slope;
    auto dact = [&](float gval, float y, bool preact) -> float {  // g * tau'(.) from the output (or the pre-activation)
        if constexpr (SA) {
            // identity: 1; ReLU: [y > 0]; LeakyReLU: y > 0 ? 1 : slope  (sign(tau(z)) == sign(z) for slope > 0, so the
            // pre-activation serves as well as the output)
            const float neg = g.act == DMPNN_ACT_NONE ? 1.f : (g.act == DMPNN_ACT_RELU ? 0.f : slope);
            if (g.drop_scale != 0.f && !preact) {
                // y = dropout(tau(z)): exactly 0 where dropped (or inactive), else tau(z) / (1 - p) with the sign of z
                return y > 0.f ? gval * g.drop_scale : (y < 0.f ? neg * gval * g.drop_scale : 0.f);
            }
            return (g.act == DMPNN_ACT_NONE || y > 0.f) ? gval : neg * gval;
        } else {
            if (preact) y = apply_act(y, g.act, slope);
            return gval * act_grad_from_out(y, g.act, slope);
        }
    };

    // ---- metadata, incidence fragments ----
    if (lean) {
        // the tile's rows are its edges in the caller's order; the tile checks that it is closed, like the forward (a tile that is
        // not gives NaN gradients for its rows and atoms — its forward output was NaN already)
        bool row_bad = false;
        int rv = 0, ad = 0, as = -1;
        if (tid < nrows) {
            const long long e = rs + tid;
            const long long r_l = g.rev64[e] - rs, s_l = g.edge_index[e] - va, d_l = g.edge_index[(long long)g.nE + e] - va;
            row_bad = r_l < 0 || r_l >= nrows || d_l < 0 || d_l >= na || s_l < 0 || s_l >= na;
            if (!row_bad) { rv = (int)r_l; ad = (int)d_l; as = (int)s_l; }
        }
        if (tid < BM) { revl[tid] = rv; aor[tid] = ad; asrc[tid] = as; }
        if (tid < 8) maxbits[tid] = 0u;
        __syncthreads();
        if (row_bad) atomicOr(&maxbits[5], 1u);
        __syncthreads();
        if (maxbits[5]) {  // (uniform)
            for (int i = tid; i < nrows * N; i += KT) {
                const long long o = (long long)(rs + i / N) * g.ldh + (i % N);
                g.gH0[o] = nanv;
                for (int s = 0; s < T_steps - 1; ++s) g.gZs[(long long)s * g.slot + o] = nanv;
            }
            for (int i = tid; i < na * N; i += KT) g.gZO[(long long)(va + i / N) * g.ldh + (i % N)] = nanv;
            if (g.gZrows) {   // ... and as split rows (what the products read): NaN in every hi half, scale 1
                const int nh = ((N + 31) >> 5) * 32;   // hi halfs of a row's live chunks
                const _Float16 hn = (_Float16)nanv;
                auto nan_rows = [&](unsigned char* base, long long r0, int n) {
                    for (int i = tid; i < n * nh; i += KT) {
                        const int r = i / nh, c = i - r * nh;
                        *reinterpret_cast<_Float16*>(base + (r0 + r) * g.tsr + (c >> 5) * 128 + (c & 31) * 2) = hn;
                    }
                    for (int r = tid; r < n; r += KT) *reinterpret_cast<float4*>(base + (r0 + r) * g.tsr + (g.tsr - 16)) = make_float4(1.f, 0.f, 0.f, 0.f);
                };
                for (int sl = 0; sl < T_steps - 1; ++sl) nan_rows(g.gZrows + (long long)sl * g.zrow_slot, rs, nrows);
                nan_rows(g.gH0rows, rs, nrows);
                nan_rows(g.gZOrows, va, na);
            }
            return;
        }
    } else {
        if (tid < BM) revl[tid] = tid < nrows ? g.revp[rs + tid] - rs : 0;
        if (tid <= BA) rp[tid] = g.row_ptr[va + (tid <= na ? tid : na)] - rs;
        if (tid < 8) maxbits[tid] = 0u;
        __syncthreads();
        if (tid < na)
            for (int r = rp[tid]; r < rp[tid + 1]; ++r) aor[r] = tid;
        __syncthreads();
        if (tid < BM) asrc[tid] = tid < nrows ? aor[revl[tid]] : -1;  // src r = dst rev r (symmetric graph: checked by the plan)
        __syncthreads();
    }
    // k order of the incidence MFMAs = the order C/D fragments hold rows (see dmpnn_mega16_impl.hpp):
    //   k-step 0, lane group lg, slot s -> row lg*4+s (s<4) | 16+lg*4+(s-4);  k-step 1 -> 32+lg*4+s (s<4) | none.
    // fragments 0..2 (jt): gather,  B[k = atom][j = row r'] = [dst r' == atom]
    // fragments 3..8 (jt, ks): message backward,  B[k = row r][j = row r'] = [src r == dst r'] - [r' == rev r]
    // (a lane's k rows depend on (k-step, lg) only: their source atoms are read once; f16 bit patterns 1.0 = 0x3C00, -1.0 = 0xBC00)
    int as_[12];
    {
        const int4 q0 = *reinterpret_cast<const int4*>(asrc + lg * 4), q1 = *reinterpret_cast<const int4*>(asrc + 16 + lg * 4),
                   q2 = *reinterpret_cast<const int4*>(asrc + 32 + lg * 4);
        const int qa[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
        for (int i = 0; i < 12; ++i) as_[i] = ((i >> 2) * 16 + lg * 4 + (i & 3)) < nrows ? qa[i] : -3;
    }
    // ... and their reverse rows: the transpose of the forward's  - [r = rev r']  is  - [r' = rev r]  — the same thing for an
    // involution (every molecular graph), and exact for any rev map inside the tile (a tile plan does not examine symmetry)
    int rk_[12];
    {
        const int4 q0 = *reinterpret_cast<const int4*>(revl + lg * 4), q1 = *reinterpret_cast<const int4*>(revl + 16 + lg * 4),
                   q2 = *reinterpret_cast<const int4*>(revl + 32 + lg * 4);
        const int qa[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
        for (int i = 0; i < 12; ++i) rk_[i] = ((i >> 2) * 16 + lg * 4 + (i & 3)) < nrows ? qa[i] : -3;
    }
    for (int f = wave; f < 9; f += NW) {
        const bool gat = f < 3;
        const int jt = gat ? f : (f - 3) >> 1, ks = gat ? 0 : (f - 3) & 1;
        const int j = jt * 16 + li;  // row r'
        const int a_t = j < nrows ? aor[j] : -2;
        unsigned hb[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const int k = ks == 0 ? (sl < 4 ? lg * 4 + sl : 16 + lg * 4 + (sl - 4)) : (sl < 4 ? 32 + lg * 4 + sl : -1);
            const int sk = ks == 0 ? as_[sl] : (sl < 4 ? as_[8 + sl] : -3);
            const int rk = ks == 0 ? rk_[sl] : (sl < 4 ? rk_[8 + sl] : -3);
            const bool in = gat ? (k < na && k == a_t) : sk == a_t, isrev = !gat && !g.atom && j < nrows && rk == j;
            hb[sl] = in ? (isrev ? 0u : 0x3C00u) : (isrev ? 0xBC00u : 0u);
        }
        const u32x4 pk = {hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16)};
        cfrag[f * 64 + lane] = __builtin_bit_cast(h8, pk);
    }
    __syncthreads();

    auto wave_max = [&](float v) -> float {
        int u = (int)__float_as_uint(v);
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0xB1, 0xf, 0xf, true));
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0x4E, 0xf, 0xf, true));
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0x141, 0xf, 0xf, true));
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0x140, 0xf, 0xf, true));
        const int m = max(max(__builtin_amdgcn_readlane(u, 0), __builtin_amdgcn_readlane(u, 16)),
                          max(__builtin_amdgcn_readlane(u, 32), __builtin_amdgcn_readlane(u, 48)));
        return __uint_as_float((unsigned)m);
    };
    int scale_phase = 0;
    float tile_mx = 0.f;   // the maximum the last tile_scale call saw (0: an all-zero tile — the tails of its split rows say so)
    auto tile_scale = [&](float local_max) -> float {
        const int slot = scale_phase & 3;
        local_max = wave_max(local_max);
        if (lane == 0) atomicMax(&maxbits[slot], __float_as_uint(local_max));
        __syncthreads();
        const float mx = __uint_as_float(maxbits[slot]);
        tile_mx = mx;
        if (tid == 0) maxbits[(slot + 2) & 3] = 0u;
        ++scale_phase;
        return scale_for(mx);
    };

    using RE = std::integral_constant<int, RT_E>;
    using RA = std::integral_constant<int, RT_A>;
    // ================= finalize backward: gZO = gHO * tau'(HO) on the tile's atoms, row-major =================
    float sA;
    {
        constexpr int ITEMS_A = BA * QN / KT;
        float4 z[ITEMS_A];
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < ITEMS_A; ++j) {
            const int it = tid + KT * j;
            const int a = it / QN, q = it - a * QN;
            const bool ok = a < na && q < qn;
            const long long row = va + (ok ? a : 0);
            const float4 gv = *reinterpret_cast<const float4*>(g.gHO + row * g.ldg + (ok ? 4 * q : 0));
            const float4 yv = *reinterpret_cast<const float4*>(g.HO + row * g.ldho + (ok ? 4 * q : 0));
            z[j] = ok ? make_float4(dact(gv.x, yv.x, false), dact(gv.y, yv.y, false), dact(gv.z, yv.z, false), dact(gv.w, yv.w, false))
                      : make_float4(0.f, 0.f, 0.f, 0.f);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(z[j].x), fabsf(z[j].y)), fmaxf(fabsf(z[j].z), fabsf(z[j].w))));
            if (ok && !g.gZOrows) store_keep4(g.gZO + row * g.ldh + 4 * q, z[j]);
        }
        sA = tile_scale(mx);
#pragma unroll
        for (int j = 0; j < ITEMS_A; ++j) {
            const int it = tid + KT * j;
            const int a = it / QN, q = it - a * QN;
            h4 hi, lo;
            split4(z[j], sA, hi, lo);
            unsigned char* p = T16 + a * TS + (q >> 3) * 128 + (q & 7) * 8;
            *reinterpret_cast<h4*>(p) = hi;
            *reinterpret_cast<h4*>(p + 64) = lo;
            if (g.gZOrows && a < na && q < qn) {   // (gZOrows: uniform) the same pieces to the atoms' split rows
                unsigned char* o = g.gZOrows + (long long)(va + a) * g.tsr + (q >> 3) * 128 + (q & 7) * 8;
                *reinterpret_cast<h4*>(o) = hi;
                *reinterpret_cast<h4*>(o + 64) = lo;
            }
        }
        if (g.gZOrows && tid < na) *reinterpret_cast<float4*>(g.gZOrows + (long long)(va + tid) * g.tsr + (g.tsr - 16)) = make_float4(sA, tile_mx > 0.f ? 0.f : 1.f, 0.f, 0.f);
    }
    // ---- the tile's body per wave class (WL column tiles from ct0(); NW = 4: WL = WN for every wave) ----
    auto body = [&](auto wl_c) __attribute__((always_inline)) {
    constexpr int WL = decltype(wl_c)::value;
    auto ct0 = [&]() -> int { return (NW == 8 && WL == 2) ? 2 * wave + 4 : WL * wave; };
    // ---- contraction acc[RT][WL] += T16 . W'^T (10 chunks for d_h = 300), weight fragments straight from L2 ----
    auto contract = [&](auto rt_c, f32x4 (&acc)[decltype(rt_c)::value][WL], const SplitW& W) {
        constexpr int RT = decltype(rt_c)::value;
        const int n_chunks = (N + 31) / 32;
        const gemm::rsrc_t rW = gemm::make_rsrc(W.p, (unsigned)(((N + 15) / 16) * W.nc * 2048));
        unsigned offB[WL];
        launder();
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) offB[ct] = (unsigned)(ct0() + ct) * (unsigned)(W.nc * 2048) + (unsigned)lane * 16u;
        auto read_a = [&](int c, h8 (&ah)[RT], h8 (&al)[RT]) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const unsigned char* p = T16 + (rt * 16 + li) * TS + c * 128 + lg * 16;
                ah[rt] = *reinterpret_cast<const h8*>(p);
                al[rt] = *reinterpret_cast<const h8*>(p + 64);
            }
        };
        // the forward tile kernel's contraction (dmpnn_mega16_impl.hpp): one set of weight fragments as a ring over the column
        // tiles, two workgroups per CU (73 KB of LDS, <= 256 registers)
        h8 bh[WL], bl[WL], a0h[RT], a0l[RT], a1h[RT], a1l[RT];
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            bh[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, offB[ct], 0, 0));
            bl[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, offB[ct] + 1024u, 0, 0));
        }
        __syncthreads();  // the split A tile is complete
        launder();
        read_a(0, a0h, a0l);
        auto chunk = [&](int c, h8 (&ah)[RT], h8 (&al)[RT], h8 (&nah)[RT], h8 (&nal)[RT]) {
            const bool more = c + 1 < n_chunks;
#pragma unroll
            for (int ct = 0; ct < WL; ++ct) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bh[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bl[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt], bh[ct], acc[rt][ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned o = more ? offB[ct] + (unsigned)(c + 1) * 2048u : gemm::kOOB;
                bh[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o, 0, 0));
                bl[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, more ? o + 1024u : gemm::kOOB, 0, 0));
                if (ct == (WL > 1 ? WL - 2 : 0)) read_a(more ? c + 1 : c, nah, nal);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma nounroll
        for (int c = 0; c < n_chunks; c += 2) {
            chunk(c, a0h, a0l, a1h, a1l);
            if (c + 1 < n_chunks) chunk(c + 1, a1h, a1l, a0h, a0l);
        }
    };
    // split domain -> fp32 (no bias): acc / (sA sW[col])
    auto unscale = [&](auto rt_c, f32x4 (&acc)[decltype(rt_c)::value][WL], float inv_sA, const float* inv_sW) {
        constexpr int RT = decltype(rt_c)::value;
        launder();
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            const int col = (ct0() + ct) * 16 + li;
            const float isw = inv_sW[col < N ? col : 0] * inv_sA;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rt][ct][r] *= isw;
        }
    };
    // incidence MFMA on C/D fragments X (RT row tiles) -> transposed fragments m[ct][jt] (row jt*16+li, 4 columns
    // ct*16 + lg*4 ..): m = C . X.  f0: first incidence fragment; two k-steps when RT == 3, one when RT == 2.
    auto incidence = [&](auto rt_c, const f32x4 (&X)[decltype(rt_c)::value][WL], int f0, f32x4 (&m)[WL][RT_E]) {
        constexpr int RT = decltype(rt_c)::value;
        launder();
        float hm = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < WL; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) hm = fmaxf(hm, fabsf(X[rt][ct][r]));
        const float sX = scale_for(wave_max(hm));
        h8 cf[RT_E][2];
#pragma unroll
        for (int jt = 0; jt < RT_E; ++jt) {
            cf[jt][0] = cfrag[(RT == 2 ? f0 + jt : f0 + 2 * jt) * 64 + lane];
            cf[jt][1] = RT == 3 ? cfrag[(f0 + 2 * jt + 1) * 64 + lane] : cf[jt][0];
        }
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            h8 ah0, al0, ah1, al1;
            {   // (the pair split on the mixed-precision FMA: mega16::split2)
                unsigned wh0[4], wl0[4], wh1[4] = {0u, 0u, 0u, 0u}, wl1[4] = {0u, 0u, 0u, 0u};
                mega16::split2(X[0][ct][0], X[0][ct][1], sX, wh0[0], wl0[0]);
                mega16::split2(X[0][ct][2], X[0][ct][3], sX, wh0[1], wl0[1]);
                mega16::split2(X[1][ct][0], X[1][ct][1], sX, wh0[2], wl0[2]);
                mega16::split2(X[1][ct][2], X[1][ct][3], sX, wh0[3], wl0[3]);
                if constexpr (RT == 3) {
                    mega16::split2(X[RT - 1][ct][0], X[RT - 1][ct][1], sX, wh1[0], wl1[0]);
                    mega16::split2(X[RT - 1][ct][2], X[RT - 1][ct][3], sX, wh1[1], wl1[1]);
                }
                ah0 = __builtin_bit_cast(h8, u32x4{wh0[0], wh0[1], wh0[2], wh0[3]}); al0 = __builtin_bit_cast(h8, u32x4{wl0[0], wl0[1], wl0[2], wl0[3]});
                ah1 = __builtin_bit_cast(h8, u32x4{wh1[0], wh1[1], wh1[2], wh1[3]}); al1 = __builtin_bit_cast(h8, u32x4{wl1[0], wl1[1], wl1[2], wl1[3]});
            }
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt) {
                f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, cf[jt][0], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, cf[jt][0], z, 0, 0, 0);
                if (RT == 3) {
                    z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, cf[jt][1], z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, cf[jt][1], z, 0, 0, 0);
                }
                m[ct][jt] = z;
            }
        }
        const float isX = mega16::rcp_pow2_exact(sX);
#pragma unroll
        for (int ct = 0; ct < WL; ++ct)
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) m[ct][jt][r] *= isX;
    };
    // m (edge gradient in transposed fragments) -> gz = m * tau'(Y rows); optional store; returns max |gz|
    auto mask_rows = [&](f32x4 (&m)[WL][RT_E], const float* Y, bool preact, float* store, int bslot) -> float {
        launder();
        if constexpr (SA) {
            if (g.keep_bits) {  // (uniform) the kept tensor as sign bits in the forward's fragment order: this lane's element (row jt 16 + li,
                // column ct 16 + 4 lg + c) is bit (li >> 2) 16 + 4 lg + c of word (jt WN + ct) 4 + (li & 3) of its wave
                // (ONE layout whatever the wave split: the word of global column tile gct sits at wave gct / WN, slot (jt WN + gct % WN) 4 + r)
                const unsigned long long* bw = g.keep_bits + (long long)bslot * g.bits_slot + (long long)t * 256 + (li & 3);
                const int sh = (li >> 2) * 16 + lg * 4;
                const float neg = g.act == DMPNN_ACT_NONE ? 1.f : (g.act == DMPNN_ACT_RELU ? 0.f : slope);
                float mxb = 0.f;
#pragma unroll
                for (int ct = 0; ct < WL; ++ct) {
                    unsigned nib[RT_E];
#pragma unroll
                    for (int jt = 0; jt < RT_E; ++jt) {
                        const int gct = ct0() + ct;
                        nib[jt] = (unsigned)(bw[(gct / WN) * 64 + (jt * WN + gct % WN) * 4] >> sh) & 0xFu;
                    }
#pragma unroll
                    for (int jt = 0; jt < RT_E; ++jt) {
                        const int row = jt * 16 + li, col4 = (ct0() + ct) * 16 + lg * 4;
                        const bool ok = row < nrows && col4 < N;
                        f32x4 v;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float gv = m[ct][jt][c];
                            v[c] = ok ? ((g.act == DMPNN_ACT_NONE || ((nib[jt] >> c) & 1u)) ? gv : neg * gv) : 0.f;
                        }
                        m[ct][jt] = v;
                        mxb = fmaxf(mxb, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                        if (store && ok) store_keep4(store + (long long)(rs + row) * g.ldh + col4, make_float4(v[0], v[1], v[2], v[3]));
                    }
                }
                return mxb;
            }
        }
        // the kept rows of column tile ct + 1 are requested while column tile ct is processed (clamped addresses: no load under a
        // branch); two column tiles in flight instead of all five: the registers are the second workgroup's
        auto load_y = [&](int ct, float4 (&y)[RT_E]) {
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt) {
                const int row = jt * 16 + li, col4 = (ct0() + ct) * 16 + lg * 4;
                const bool ok = row < nrows && col4 < N;
                y[jt] = *reinterpret_cast<const float4*>(Y + (long long)(rs + (ok ? row : 0)) * g.ldh + (ok ? col4 : 0));
            }
        };
        float4 y[2][RT_E];
        load_y(0, y[0]);
        float mx = 0.f;
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            if (ct + 1 < WL) load_y(ct + 1, y[(ct + 1) & 1]);
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt) {
                const int row = jt * 16 + li, col4 = (ct0() + ct) * 16 + lg * 4;
                const bool ok = row < nrows && col4 < N;
                const float4 yv = y[ct & 1][jt];
                f32x4 v;
                v[0] = ok ? dact(m[ct][jt][0], yv.x, preact) : 0.f;
                v[1] = ok ? dact(m[ct][jt][1], yv.y, preact) : 0.f;
                v[2] = ok ? dact(m[ct][jt][2], yv.z, preact) : 0.f;
                v[3] = ok ? dact(m[ct][jt][3], yv.w, preact) : 0.f;
                m[ct][jt] = v;
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                if (store && ok) store_keep4(store + (long long)(rs + row) * g.ldh + col4, make_float4(v[0], v[1], v[2], v[3]));
            }
        }
        return mx;
    };
    // transposed fragments -> split A tile of the next contraction (all 48 rows, zero where there is no row / column)
    // `rows` (or null): the same pieces also to the split rows [n_edges][tsr] of a product operand, the scale into the rows' tails;
    // to_lds false: those alone (gH0: nothing contracts it here)
    auto stage_rows = [&](const f32x4 (&m)[WL][RT_E], float s, unsigned char* rows, bool to_lds) {
        launder();
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            const int col4 = (ct0() + ct) * 16 + lg * 4;
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt) {
                const int row = jt * 16 + li;
                h4 hi, lo;
                split4(make_float4(m[ct][jt][0], m[ct][jt][1], m[ct][jt][2], m[ct][jt][3]), s, hi, lo);
                if (to_lds) {
                    unsigned char* p = T16 + row * TS + (col4 >> 5) * 128 + (col4 & 31) * 2;
                    *reinterpret_cast<h4*>(p) = hi;
                    *reinterpret_cast<h4*>(p + 64) = lo;
                }
                if (rows && row < nrows && col4 < N) {   // (rows: uniform)
                    unsigned char* q = rows + (long long)(rs + row) * g.tsr + (col4 >> 5) * 128 + (col4 & 31) * 2;
                    *reinterpret_cast<h4*>(q) = hi;
                    *reinterpret_cast<h4*>(q + 64) = lo;
                }
            }
        }
        if (rows && wave == 0 && lg == 0) {
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt)
                if (jt * 16 + li < nrows)
                    *reinterpret_cast<float4*>(rows + (long long)(rs + jt * 16 + li) * g.tsr + (g.tsr - 16)) = make_float4(s, tile_mx > 0.f ? 0.f : 1.f, 0.f, 0.f);
        }
    };
    // the gradient with respect to H0 leaves: fp32 rows, or (split rows) with a tile scale of its own
    auto store_gh0 = [&](const f32x4 (&x)[WL][RT_E]) {
        launder();
        if (g.gH0rows) {   // (uniform)
            float mx = 0.f;
#pragma unroll
            for (int ct = 0; ct < WL; ++ct)
#pragma unroll
                for (int jt = 0; jt < RT_E; ++jt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) mx = fmaxf(mx, fabsf(x[ct][jt][c]));
            const float s0 = tile_scale(mx);
            stage_rows(x, s0, g.gH0rows, false);
            return;
        }
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            const int col4 = (ct0() + ct) * 16 + lg * 4;
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt) {
                const int row = jt * 16 + li;
                if (row < nrows && col4 < N)
                    store_keep4(g.gH0 + (long long)(rs + row) * g.ldh + col4, make_float4(x[ct][jt][0], x[ct][jt][1], x[ct][jt][2], x[ct][jt][3]));
            }
        }
    };

    // gMv = gZO . W_o[:, d_v:]
    f32x4 m[WL][RT_E];
    {
        f32x4 acc[RT_A][WL];
#pragma unroll
        for (int rt = 0; rt < RT_A; ++rt)
#pragma unroll
            for (int ct = 0; ct < WL; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        contract(RA{}, acc, g.WoMT);
        unscale(RA{}, acc, mega16::rcp_pow2_exact(sA), g.WoMT.inv_scale);
        incidence(RA{}, acc, 0, m);  // gH[r] = gMv[dst r]
    }
    if (g.g_edge) {  // (uniform) + dL/dH^(T-1) of the edge read-out
        launder();
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            float4 y[RT_E];
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt) {
                const int row = jt * 16 + li, col4 = (ct0() + ct) * 16 + lg * 4;
                const bool ok = row < nrows && col4 < N;
                y[jt] = *reinterpret_cast<const float4*>(g.g_edge + (long long)(rs + (ok ? row : 0)) * g.ld_ge + (ok ? col4 : 0));
                if (!ok) y[jt] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt) {
                m[ct][jt][0] += y[jt].x; m[ct][jt][1] += y[jt].y; m[ct][jt][2] += y[jt].z; m[ct][jt][3] += y[jt].w;
            }
        }
    }
    f32x4 gh0[WL][RT_E];
    if (T_steps == 1) {
        mask_rows(m, g.H0, true, g.gH0rows ? nullptr : g.gH0, 0);
        if (g.gH0rows) store_gh0(m);
        return;
    }
    // gZ^(T-1) = gH * tau'(H^(T-1));  gH0 = gZ^(T-1)
    {
        const float mx = mask_rows(m, g.Hs + (long long)(T_steps - 2) * g.slot, false, g.gZrows ? nullptr : g.gZs + (long long)(T_steps - 2) * g.slot, T_steps - 1);
#pragma unroll
        for (int ct = 0; ct < WL; ++ct)
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt) gh0[ct][jt] = m[ct][jt];
        sA = tile_scale(mx);  // (barrier: every wave is past its reads of T16)
        stage_rows(m, sA, g.gZrows ? g.gZrows + (long long)(T_steps - 2) * g.zrow_slot : nullptr, true);
    }
    for (int t = T_steps - 1; t >= 1; --t) {
        f32x4 acc[RT_E][WL];
#pragma unroll
        for (int rt = 0; rt < RT_E; ++rt)
#pragma unroll
            for (int ct = 0; ct < WL; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        contract(RE{}, acc, g.WhT);              // gM = gZ^(t) . W_h
        unscale(RE{}, acc, mega16::rcp_pow2_exact(sA), g.WhT.inv_scale);
        incidence(RE{}, acc, 3, m);              // gH^(t-1) = C^T gM
        if (t - 1 >= 1) {
            const float mx = mask_rows(m, g.Hs + (long long)(t - 2) * g.slot, false, g.gZrows ? nullptr : g.gZs + (long long)(t - 2) * g.slot, t - 1);
#pragma unroll
            for (int ct = 0; ct < WL; ++ct)
#pragma unroll
                for (int jt = 0; jt < RT_E; ++jt) gh0[ct][jt] += m[ct][jt];
            sA = tile_scale(mx);
            stage_rows(m, sA, g.gZrows ? g.gZrows + (long long)(t - 2) * g.zrow_slot : nullptr, true);
        } else {
            mask_rows(m, g.H0, true, nullptr, 0);   // through H^(0) = tau(H0)
#pragma unroll
            for (int ct = 0; ct < WL; ++ct)
#pragma unroll
                for (int jt = 0; jt < RT_E; ++jt) gh0[ct][jt] += m[ct][jt];
        }
    }
    store_gh0(gh0);
    };  // body
    if constexpr (NW == 4) {
        body(std::integral_constant<int, WN>{});
    } else {
        if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) < 4) body(std::integral_constant<int, 3>{});
        else body(std::integral_constant<int, 2>{});
    }
}

template <int WN, bool SA, int NW = 4>
int launch_mega16_bwd(const Mega16BwdK& g, int n_tiles, hipStream_t s);

#define DMPNN_DEFINE_MEGA16_BWD(WN, SA) DMPNN_DEFINE_MEGA16_BWD_NW(WN, SA, 4)
#define DMPNN_DEFINE_MEGA16_BWD_NW(WN, SA, NW)                                                              \
    template <>                                                                                             \
    int launch_mega16_bwd<WN, SA, NW>(const Mega16BwdK& g, int n_tiles, hipStream_t s) {                    \
        constexpr size_t lds = bwd_lds_bytes<WN>();                                                         \
        static bool attr_set = false;                                                                       \
        if (!attr_set) {                                                                                    \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mpnn_tile16_bwd<WN, SA, NW>),       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
            if (e != hipSuccess) {                                                                          \
                set_error("hipFuncSetAttribute(k_mpnn_tile16_bwd<%d>): %s", WN, hipGetErrorString(e));      \
                return DMPNN_EHIP;                                                                          \
            }                                                                                               \
            attr_set = true;                                                                                \
        }                                                                                                   \
        hipLaunchKernelGGL((k_mpnn_tile16_bwd<WN, SA, NW>), dim3((unsigned)n_tiles), dim3(64 * NW), lds, s, g); \
        DMPNN_CHECK_LAUNCH("k_mpnn_tile16_bwd");                                                            \
        return DMPNN_OK;                                                                                    \
    }

}  // namespace mega16
}  // namespace dmpnn
