// Whole BondMessagePassing.forward (base.py:196-212) for ONE tile of whole molecules in ONE launch.
//
// No edge crosses a molecule (data/collate.py:48-56), so a row tile made of whole connected pieces
// (plan: mtile_row / mtile_atom, <= 48 edge rows, <= 32 atoms) is closed under rev(): the message of
// every depth step stays inside the tile.  One workgroup therefore carries its tile through
//     K1  H0 = W_i [V[src] || E]        (kept in REGISTERS, C/D fragment layout: the residual)
//     K2  M  = S[dst] - tau(H0)         (LDS tile, formed in place)
//     K3  H  = tau(H0 + W_h M)          x (depth - 1), A operand read straight from the LDS tile
//     K4  Mv = S                        (LDS tile rows 0 .. atoms-1)
//     K5  out = tau(W_o [V || Mv] + b_o)
// with no kernel boundary and no HBM round trip of H / M in between (inference writes ONLY `out`;
// training additionally streams H0, H^(t), M^(t), Mv out for the backward pass).  The weights stream
// from L2 through a 2-slot LDS ring exactly as in k_gemm (dmpnn_gemm_impl.hpp), same MFMA / k order.
//
// LDS: T[48][LDC] (tile: A operand of the updates, epilogue staging; the A staging ring of the two
// global-A contractions overlays it) | B ring [2][64*WN][36] | tile metadata.  d_h = 300: 154 KB.
#pragma once

#include <type_traits>

#include "dmpnn_gemm_impl.hpp"
#include "dmpnn_spill_impl.hpp"

namespace dmpnn {
namespace mega {

using gemm::BK;
using gemm::BKP;
using gemm::f32x4;
using gemm::kOOB;
using gemm::kThreads;
using gemm::rsrc_t;
using gemm::u32x4;

constexpr int RT_E = kMegaBM / 16;  // 3 row tiles of edge rows
constexpr int RT_A = kMegaBA / 16;  // 2 row tiles of atom rows (finalize)

struct MegaK {
    const int* mtile_row; const int* mtile_atom; const int* row_ptr; const int* srcp; const int* perm; const int* revp;
    const int* flags; int poison_mask;
    int nV, nE, d_v, d_e, h, depth;
    const float* V; int ldv; const float* E; int lde; unsigned v_bytes, e_bytes;
    const float* W_i; const float* b_i; const float* W_h; const float* b_h; const float* W_o; const float* b_o;
    int act; float slope; const float* slope_ptr;
    float* out; int ldout;
    float* H0; float* Hs; float* Ms; float* Mv; int ldh; long long slot;  // kept tensors (training) or null
    unsigned qmagic;
    long long* dbg;   // optional [32] cycle stamps of workgroup 0 (dmpnn_debug_timestamps), else null
    // tile plan (dmpnn_prepare_tiles, header LIGHT == 2; split-MFMA kernel only): the caller's own index arrays
    const long long* edge_index;  // [2, nE] row 0 = src atom, row 1 = dst atom
    const long long* rev64;       // [nE]
    float* spill;                 // inference scratch of the generic path for oversize pieces ([3 nE + nV][ldh]) or null
    // active dropout (dmpnn_fwd_args.dropout_p; split-MFMA training forward only): threshold floor(p 2^32) (0: off), 1 / (1 - p), seed
    unsigned drop_thr; float drop_scale; unsigned seed_lo, seed_hi;
};

// the view the generic path (dmpnn_spill_impl.hpp) takes of a piece tile that exceeds the matrix-pipe tile
__device__ __forceinline__ spill::FwdView spill_view(const MegaK& g, bool lean, int rs, int nrows, int va, int na, float slope) {
    spill::FwdView v;
    v.lean = lean; v.rs = rs; v.nrows = nrows; v.va = va; v.na = na; v.nV = g.nV; v.nE = g.nE;
    v.row_ptr = g.row_ptr; v.srcp = g.srcp; v.perm = g.perm; v.revp = g.revp;
    v.edge_index = g.edge_index; v.rev64 = g.rev64;
    v.V = g.V; v.ldv = g.ldv; v.E = g.E; v.lde = g.lde; v.d_v = g.d_v; v.d_e = g.d_e; v.h = g.h; v.depth = g.depth;
    v.W_i = g.W_i; v.b_i = g.b_i; v.W_h = g.W_h; v.b_h = g.b_h; v.W_o = g.W_o; v.b_o = g.b_o;
    v.act = g.act; v.slope = slope;
    v.out = g.out; v.ldout = g.ldout;
    v.ldh = g.ldh; v.slot = g.slot;
    const bool kept = g.H0 != nullptr;  // DMPNN_F_KEEP: the kept tensors are the workspace
    const int steps = g.depth > 1 ? g.depth - 1 : 1;
    v.H0 = kept ? g.H0 : g.spill;
    v.Hs = kept ? g.Hs : (g.spill ? g.spill + g.slot : nullptr);
    v.Ms = kept ? g.Ms : (g.spill ? g.spill + 2 * g.slot : nullptr);
    v.Mv = kept ? g.Mv : (g.spill ? g.spill + 3 * g.slot : nullptr);
    v.n_hslots = kept ? steps : 1;
    v.n_mslots = kept ? steps : 1;
    return v;
}

template <int WN>
constexpr size_t lds_bytes() {
    return (size_t)(kMegaBM * (64 * WN + 4) + 2 * (64 * WN) * BKP) * sizeof(float) + (size_t)(2 * kMegaBM + kMegaBA + 8) * sizeof(int);
}

template <int WN>
__global__ __launch_bounds__(kThreads) void k_mpnn_tile(MegaK g) {
    constexpr int BM = kMegaBM, BA = kMegaBA, BN = 64 * WN, LDC = BN + 4, QN = BN / 4;
    constexpr int ITEMS = BM * QN / kThreads;  // (row, quad) items per thread in the tile passes
    constexpr int SLOTS_B = BN * (BK / 4) / kThreads;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T = smem;                       // [BM][LDC]
    float* As = smem;                      // [2][BM][BKP]  (overlays T while T is free)
    float* Bs = smem + BM * LDC;           // [2][BN][BKP]
    int* revl = reinterpret_cast<int*>(Bs + 2 * BN * BKP);  // [BM] tile-local row of the reverse edge
    int* aor = revl + BM;                  // [BM] tile-local atom of a row
    int* rp = aor + BM;                    // [BA + 1] tile-local row pointer

    // Thread coordinates are re-derived through an opaque asm at every phase (launder()): otherwise
    // hipcc hoists the address arithmetic of every later phase out of the depth loop and spills.
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int li = lane & 15, lg = lane >> 4;
    int kq = tid & 7;
    auto launder = [&]() {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wave = tid >> 6; li = lane & 15; lg = lane >> 4; kq = tid & 7;
    };
    const int t = blockIdx.x;
    const int rs = g.mtile_row[t], re = g.mtile_row[t + 1];
    const int va = g.mtile_atom[t], vb = g.mtile_atom[t + 1];
    const int nrows = re - rs, na = vb - va;
    const int N = g.h, qn = N >> 2;
    const bool poison = (g.flags[0] & g.poison_mask) != 0;
    if (poison) {  // a graph this kernel cannot represent (tables may be empty): the whole output is NaN, loudly
        const float nanv = __int_as_float(0x7fc00000);
        const long long total = (long long)g.nV * N;
        for (long long i = (long long)blockIdx.x * kThreads + tid; i < total; i += (long long)gridDim.x * kThreads)
            g.out[(i / N) * g.ldout + (i % N)] = nanv;
        return;
    }
    if (na <= 0 || nrows < 0) return;  // trailing slots of the launch bound
    if (nrows > BM || na > BA) {  // a piece larger than the matrix-pipe tile: the generic fp32 path (any size)
        const MegaK& gs = *spill::fresh_kernargs<MegaK>();
        spill::forward(spill_view(gs, false, rs, nrows, va, na, gs.slope_ptr ? *gs.slope_ptr : gs.slope), smem);
        return;
    }
    const float slope = g.slope_ptr ? *g.slope_ptr : g.slope;
    const float neg_slope = g.act == DMPNN_ACT_NONE ? 1.f : (g.act == DMPNN_ACT_RELU ? 0.f : slope);
    const bool simple_act = !(g.act == DMPNN_ACT_TANH || g.act == DMPNN_ACT_ELU);
    auto tau = [&](float z) -> float {
        if (simple_act) return (z > 0.f ? z : neg_slope * z) + 0.f;
        return apply_act(z, g.act, slope);
    };

    // ---- tile metadata -> LDS; zero the tile (its pad columns must stay finite: they meet 0-weights) ----
    for (int i = tid; i < BM * LDC / 4; i += kThreads) reinterpret_cast<float4*>(T)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < BM) revl[tid] = tid < nrows ? g.revp[rs + tid] - rs : 0;
    if (tid <= BA) rp[tid] = g.row_ptr[va + (tid <= na ? tid : na)] - rs;
    __syncthreads();
    if (tid < na) {
        for (int r = rp[tid]; r < rp[tid + 1]; ++r) aor[r] = tid;
    }
    // (visible after the barriers inside the first contraction)

    // ---- one contraction phase: acc[RT][WN] (+)= A[rows, 0:K] . W[:, 0:K]^T, W rows of ldw floats ----
    // A_LDS: A = T[:, 0:K] (fragments straight from the tile);  else A = [A1 | A2] from global memory through
    // the staging ring (which overlays T).  G = floats per buffer load of the global operands.
    auto contract = [&](auto rt_c, auto g_c, auto a_lds_c, auto has_a2_c, f32x4 (&acc)[decltype(rt_c)::value][WN], int K1,
                        int K2, rsrc_t rA1, rsrc_t rA2, const unsigned (&offA1)[2], const unsigned (&offA2)[2],
                        const float* W, int ldw, unsigned w_bytes) {
        constexpr int RT = decltype(rt_c)::value, G = decltype(g_c)::value, NS = 4 / G;
        constexpr bool A_LDS = decltype(a_lds_c)::value, HAS_A2 = decltype(has_a2_c)::value;
        constexpr int BMr = 16 * RT;
        constexpr int SLOTS_A = A_LDS ? 0 : (BMr * (BK / 4) + kThreads - 1) / kThreads;  // <= 2
        const int K = K1 + K2;
        const int n_chunks = (K + BK - 1) / BK;
        const rsrc_t rW = gemm::make_rsrc(W, w_bytes);
        launder();
        unsigned offB[SLOTS_B];
#pragma unroll
        for (int j = 0; j < SLOTS_B; ++j) {
            const int col = (tid + kThreads * j) >> 3;
            offB[j] = col < N ? (unsigned)col * (unsigned)ldw * 4u : kOOB;
        }
        u32x4 stA[SLOTS_A > 0 ? SLOTS_A : 1], stB[SLOTS_B];
        auto load_chunk = [&](int c) {
            const int kk = c * BK + kq * 4;
            unsigned k1o[NS], k2o[NS], kbo[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k = kk + s * G;
                k1o[s] = k < K1 ? (unsigned)k * 4u : kOOB;
                k2o[s] = (k >= K1 && k < K) ? (unsigned)(k - K1) * 4u : kOOB;
                kbo[s] = k < K ? (unsigned)k * 4u : kOOB;
            }
#pragma unroll
            for (int j = 0; j < SLOTS_A; ++j) {
                unsigned o1[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) o1[s] = gemm::join_off(offA1[j], k1o[s]);
                u32x4 v = gemm::load_quad<G>(rA1, o1);
                if constexpr (HAS_A2) {
                    unsigned o2[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) o2[s] = gemm::join_off(offA2[j], k2o[s]);
                    v = v | gemm::load_quad<G>(rA2, o2);
                }
                stA[j] = v;
            }
#pragma unroll
            for (int j = 0; j < SLOTS_B; ++j) {
                unsigned ob[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) ob[s] = gemm::join_off(offB[j], kbo[s]);
                stB[j] = gemm::load_quad<G>(rW, ob);
            }
        };
        auto store_chunk = [&](int slot) {
            float* Ad = As + slot * BMr * BKP;
            float* Bd = Bs + slot * BN * BKP;
#pragma unroll
            for (int j = 0; j < SLOTS_A; ++j) {
                const int r = (tid + kThreads * j) >> 3;
                if (r < BMr) *reinterpret_cast<float4*>(Ad + r * BKP + kq * 4) = gemm::as_f4(stA[j]);
            }
#pragma unroll
            for (int j = 0; j < SLOTS_B; ++j) {
                const int n = (tid + kThreads * j) >> 3;
                *reinterpret_cast<float4*>(Bd + n * BKP + kq * 4) = gemm::as_f4(stB[j]);
            }
        };
        auto read_frags = [&](int c, float (&fa)[RT][8], float (&fb)[WN][8]) {
            const int slot = c & 1;
            const float* Bc = Bs + slot * BN * BKP + wave * (16 * WN) * BKP;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float* p = A_LDS ? T + (rt * 16 + li) * LDC + c * BK + lg * 8
                                       : As + slot * BMr * BKP + (rt * 16 + li) * BKP + lg * 8;
                const float4 t0 = *reinterpret_cast<const float4*>(p);
                const float4 t1 = *reinterpret_cast<const float4*>(p + 4);
                fa[rt][0] = t0.x; fa[rt][1] = t0.y; fa[rt][2] = t0.z; fa[rt][3] = t0.w;
                fa[rt][4] = t1.x; fa[rt][5] = t1.y; fa[rt][6] = t1.z; fa[rt][7] = t1.w;
            }
#pragma unroll
            for (int ct = 0; ct < WN; ++ct) {
                const float* p = Bc + (ct * 16 + li) * BKP + lg * 8;
                const float4 t0 = *reinterpret_cast<const float4*>(p);
                const float4 t1 = *reinterpret_cast<const float4*>(p + 4);
                fb[ct][0] = t0.x; fb[ct][1] = t0.y; fb[ct][2] = t0.z; fb[ct][3] = t0.w;
                fb[ct][4] = t1.x; fb[ct][5] = t1.y; fb[ct][6] = t1.z; fb[ct][7] = t1.w;
            }
        };
        auto mfma_half = [&](int q0, const float (&fa)[RT][8], const float (&fb)[WN][8]) {
#pragma unroll
            for (int q = q0; q < q0 + 4; ++q)
#pragma unroll
                for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[rt][q], fb[ct][q], acc[rt][ct], 0, 0, 0);
        };
        constexpr int N_STAGE_LOADS = (SLOTS_A * (HAS_A2 ? 2 : 1) + SLOTS_B) * NS;
        constexpr int N_STAGE_WRITES = SLOTS_A + SLOTS_B;
        constexpr int N_HALF = 4 * RT * WN;
        auto chunk = [&](auto has_next, auto has_next2, int c, float (&fa)[RT][8], float (&fb)[WN][8], float (&ga)[RT][8],
                         float (&gb)[WN][8]) {
            constexpr bool NEXT = decltype(has_next)::value, NEXT2 = decltype(has_next2)::value;
            if constexpr (NEXT) store_chunk((c + 1) & 1);
            if constexpr (NEXT2) load_chunk(c + 2);
            mfma_half(0, fa, fb);
            if constexpr (NEXT) {
                constexpr int NW = N_STAGE_WRITES < N_HALF ? N_STAGE_WRITES : N_HALF;
                constexpr int NL = NEXT2 ? (N_STAGE_LOADS < N_HALF - NW ? N_STAGE_LOADS : N_HALF - NW) : 0;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT) {
                __syncthreads();
                read_frags(c + 1, ga, gb);
            }
            mfma_half(4, fa, fb);
            __builtin_amdgcn_sched_barrier(0);
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        float f0a[RT][8], f0b[WN][8], f1a[RT][8], f1b[WN][8];
        __syncthreads();  // the previous phase is done with T / the rings
        launder();
        load_chunk(0);
        store_chunk(0);
        if (n_chunks > 1) load_chunk(1);
        __syncthreads();
        read_frags(0, f0a, f0b);
        __builtin_amdgcn_sched_barrier(0);
        int c = 0;
        for (; c + 3 < n_chunks; c += 2) {
            chunk(T_{}, T_{}, c, f0a, f0b, f1a, f1b);
            chunk(T_{}, T_{}, c + 1, f1a, f1b, f0a, f0b);
        }
        const int left = n_chunks - c;
        if (left == 3) {
            chunk(T_{}, T_{}, c, f0a, f0b, f1a, f1b);
            chunk(T_{}, F_{}, c + 1, f1a, f1b, f0a, f0b);
            chunk(F_{}, F_{}, c + 2, f0a, f0b, f1a, f1b);
        } else if (left == 2) {
            chunk(T_{}, F_{}, c, f0a, f0b, f1a, f1b);
            chunk(F_{}, F_{}, c + 1, f1a, f1b, f0a, f0b);
        } else {
            chunk(F_{}, F_{}, c, f0a, f0b, f1a, f1b);
        }
        __syncthreads();  // every wave is done reading T / the rings
    };
    auto init_acc = [&](auto rt_c, f32x4 (&acc)[decltype(rt_c)::value][WN], const float* bias) {
        constexpr int RT = decltype(rt_c)::value;
        const float* bp = bias ? bias : g.W_h;
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) {
            const int col = wave * (16 * WN) + ct * 16 + li;
            const bool okc = col < N && bias != nullptr;
            const float braw = bp[okc ? col : 0];
            const float bv = okc ? braw : 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = f32x4{bv, bv, bv, bv};
        }
    };
    // C/D fragments -> T (col = l&15, row = (l>>4)*4 + reg); columns >= N are never written (stay 0)
    auto frag_to_tile = [&](auto rt_c, const f32x4 (&y)[decltype(rt_c)::value][WN]) {
        constexpr int RT = decltype(rt_c)::value;
        launder();
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) {
            const int col = wave * (16 * WN) + ct * 16 + li;
            if (col < N) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[(rt * 16 + lg * 4 + r) * LDC + col] = y[rt][ct][r];
            }
        }
    };
    // coalesced store of tile rows [0, n_r) to global rows row0.. (16-byte row segments)
    auto tile_to_global = [&](float* dst, long long row0, int ld, int n_r, bool nan_out) {
        launder();
        const float nanv = __int_as_float(0x7fc00000);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = tid + kThreads * j;
            const int r = it / QN, q = it - r * QN;
            if (r < n_r && q < qn) {
                float4 v = *reinterpret_cast<const float4*>(T + r * LDC + 4 * q);
                if (nan_out) v = make_float4(nanv, nanv, nanv, nanv);
                *reinterpret_cast<float4*>(dst + (row0 + r) * ld + 4 * q) = v;
            }
        }
    };
    // message / aggregate from the tile, IN PLACE (results held in registers across one barrier):
    //   last == false:  T[rev(r)] <- S[dst(r)] - T[r]      (mixins.py:11-18), optionally streamed to Mkeep
    //   last == true :  T[a]      <- S[a]                  (base.py:208-211), optionally streamed to Mv
    auto segment_pass = [&](bool last, float* keep, int keep_ld) {
        launder();
        float4 res[ITEMS];
        int dstrow[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = tid + kThreads * j;
            const int r = it / QN, q = it - r * QN;  // r: tile row (message) or tile atom (aggregate)
            dstrow[j] = -1;
            res[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int lim = last ? na : nrows;
            if (r < lim && q < qn) {
                const int a = last ? r : aor[r];
                const int r0 = rp[a], r1 = rp[a + 1];
                float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int rr = r0; rr < r1; ++rr) {  // increasing edge id: the reference's sequential scatter order
                    const float4 y = *reinterpret_cast<const float4*>(T + rr * LDC + 4 * q);
                    if (rr == r0) S = y;
                    else { S.x += y.x; S.y += y.y; S.z += y.z; S.w += y.w; }
                }
                if (last) {
                    res[j] = S;
                    dstrow[j] = r;
                } else {
                    const float4 y = *reinterpret_cast<const float4*>(T + r * LDC + 4 * q);
                    res[j] = make_float4(S.x - y.x, S.y - y.y, S.z - y.z, S.w - y.w);
                    dstrow[j] = revl[r];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = tid + kThreads * j;
            const int q = it % QN;
            if (dstrow[j] >= 0) {
                *reinterpret_cast<float4*>(T + dstrow[j] * LDC + 4 * q) = res[j];
                if (keep) *reinterpret_cast<float4*>(keep + ((long long)(last ? va : rs) + dstrow[j]) * keep_ld + 4 * q) = res[j];
            }
        }
        // (the next phase starts with a barrier)
    };

    using RE = std::integral_constant<int, RT_E>;
    using RA = std::integral_constant<int, RT_A>;
    using G2 = std::integral_constant<int, 2>;
    using G4 = std::integral_constant<int, 4>;
    using T_ = std::true_type;
    using F_ = std::false_type;
    const int T_steps = g.depth;

    // ================= K1: H0 = W_i [V[src] || E] =================
    f32x4 h0[RT_E][WN];
    {
        unsigned offA1[2], offA2[2];
        int i1[2], i2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (tid + kThreads * j) >> 3;
            const bool ok = r < BM && r < nrows;
            i1[j] = g.srcp[ok ? rs + r : 0];
            i2[j] = g.perm[ok ? rs + r : 0];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (tid + kThreads * j) >> 3;
            const bool ok = r < BM && r < nrows;
            offA1[j] = ok ? (unsigned)i1[j] * (unsigned)g.ldv * 4u : kOOB;
            offA2[j] = ok ? (unsigned)i2[j] * (unsigned)g.lde * 4u : kOOB;
        }
        init_acc(RE{}, h0, g.b_i);
        contract(RE{}, G2{}, F_{}, T_{}, h0, g.d_v, g.d_e, gemm::make_rsrc(g.V, g.v_bytes), gemm::make_rsrc(g.E, g.e_bytes),
                 offA1, offA2, g.W_i, g.d_v + g.d_e, (unsigned)(N * (g.d_v + g.d_e)) * 4u);
    }
    // the staging ring overlaid the tile: put its pad columns (they meet zero weights) back to 0
    for (int i = tid; i < BM * (LDC - N); i += kThreads) T[(i / (LDC - N)) * LDC + N + i % (LDC - N)] = 0.f;
    if (g.H0) {  // training: the pre-activation is needed by the backward pass
        frag_to_tile(RE{}, h0);
        __syncthreads();
        tile_to_global(g.H0, rs, g.ldh, nrows, false);
        __syncthreads();
    }
    {
        f32x4 y[RT_E][WN];
#pragma unroll
        for (int rt = 0; rt < RT_E; ++rt)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[rt][ct][r] = tau(h0[rt][ct][r]);
        frag_to_tile(RE{}, y);
    }
    __syncthreads();
    segment_pass(T_steps == 1, T_steps == 1 ? g.Mv : g.Ms, g.ldh);

    // ================= K3 x (depth - 1): H = tau(H0 + W_h M) =================
    for (int step = 1; step < T_steps; ++step) {
        f32x4 acc[RT_E][WN];
        init_acc(RE{}, acc, g.b_h);
        const unsigned dummy[2] = {kOOB, kOOB};
        const rsrc_t rnull = gemm::make_rsrc(g.W_h, 0);
        contract(RE{}, G4{}, T_{}, F_{}, acc, N, 0, rnull, rnull, dummy, dummy, g.W_h, N, (unsigned)(N * N) * 4u);
#pragma unroll
        for (int rt = 0; rt < RT_E; ++rt)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rt][ct][r] = tau(h0[rt][ct][r] + acc[rt][ct][r]);  // H0 + W_h(M): base.py:141
        frag_to_tile(RE{}, acc);
        __syncthreads();
        if (g.Hs) tile_to_global(g.Hs + (long long)(step - 1) * g.slot, rs, g.ldh, nrows, false);
        const bool last = step == T_steps - 1;
        __syncthreads();
        segment_pass(last, last ? g.Mv : (g.Ms ? g.Ms + (long long)step * g.slot : nullptr), g.ldh);
    }

    // ================= K5: out = tau(W_o [V || Mv] + b_o) on the tile's atoms =================
    {
        f32x4 acc[RT_A][WN];
        init_acc(RA{}, acc, g.b_o);
        const unsigned dummy[2] = {kOOB, kOOB};
        const rsrc_t rnull = gemm::make_rsrc(g.W_o, 0);
        const int ldw = g.d_v + N;
        // Mv part first (A = tile rows 0..atoms-1), then the V part (its staging ring overlays the tile)
        contract(RA{}, G2{}, T_{}, F_{}, acc, N, 0, rnull, rnull, dummy, dummy, g.W_o + g.d_v, ldw,
                 (unsigned)((long long)N * ldw - g.d_v) * 4u);
        unsigned offA1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (tid + kThreads * j) >> 3;
            offA1[j] = (r < BA && r < na) ? (unsigned)r * (unsigned)g.ldv * 4u : kOOB;
        }
        contract(RA{}, G2{}, F_{}, F_{}, acc, g.d_v, 0, gemm::make_rsrc(g.V + (long long)va * g.ldv, (unsigned)(na * g.ldv) * 4u), rnull,
                 offA1, dummy, g.W_o, ldw, (unsigned)((long long)N * ldw) * 4u);
#pragma unroll
        for (int rt = 0; rt < RT_A; ++rt)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rt][ct][r] = tau(acc[rt][ct][r]);
        frag_to_tile(RA{}, acc);
        __syncthreads();
        tile_to_global(g.out, va, g.ldout, na, poison);
    }
}

template <int WN>
int launch_mega(const MegaK& g, int n_tiles, hipStream_t s);

#define DMPNN_DEFINE_MEGA(WN)                                                                              \
    template <>                                                                                            \
    int launch_mega<WN>(const MegaK& g, int n_tiles, hipStream_t s) {                                      \
        constexpr size_t lds = lds_bytes<WN>();                                                            \
        static bool attr_set = false;                                                                      \
        if (!attr_set) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mpnn_tile<WN>),            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
            if (e != hipSuccess) {                                                                         \
                set_error("hipFuncSetAttribute(k_mpnn_tile<%d>, %zu B LDS): %s", WN, lds, hipGetErrorString(e)); \
                return DMPNN_EHIP;                                                                         \
            }                                                                                              \
            attr_set = true;                                                                               \
        }                                                                                                  \
        hipLaunchKernelGGL((k_mpnn_tile<WN>), dim3((unsigned)n_tiles), dim3(kThreads), lds, s, g);         \
        DMPNN_CHECK_LAUNCH("k_mpnn_tile");                                                                 \
        return DMPNN_OK;                                                                                   \
    }

}  // namespace mega
}  // namespace dmpnn
