// fp32-MFMA row-panel contraction with optional fused segment epilogue (internal template; the
// instantiation units are dmpnn_gemm_*.hip, the dispatcher is dmpnn_gemm.hip).
//
//   Z[r, :]  = [A1[g1(r), 0:K1] || A2[g2(r), 0:K2]] . W^T + bias + Cadd[r, :]
//   Y[r, :]  = tau(Z[r, :])
//   EPI_PLAIN : store Y (and Z when Zpre is given)
//   EPI_SEG   : rows are directed edges in CSR order (grouped by destination atom, increasing
//               edge id inside an atom), a tile holds WHOLE atoms; additionally
//                   S[v]        = sum_{rows r of v, in order} Y[r]                 (base.py:208-211)
//                   Mout[rev(r)] = S[v] - Y[r]                                      (mixins.py:11-18)
//               i.e. the next depth step's message (or the final per-atom aggregate) is produced
//               by the epilogue of the kernel that produced H, from the LDS-resident output tile:
//               no separate scatter/gather kernel, no [E,h] H round trip through HBM.
//
// gfx950 mapping
//   * v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles/SIMD, 256 FLOP/clk/CU = the fp32 roof).
//     256 threads = 4 waves (one per SIMD); a workgroup owns BM = 16*RT rows x BN = 64*WN columns,
//     wave w owns columns [16*WN*w, 16*WN*(w+1)).  With d_h = 300, WN = 5 -> BN = 320 >= N: the
//     panel holds complete rows, A is read once, the epilogue sees whole rows.
//   * K is walked in 32-wide chunks through a 2-slot LDS ring; global loads run two chunks ahead
//     through staging registers (1 wave/SIMD has the whole 512-entry register file).
//   * k-permutation: MFMA 16x16x4 takes A[i][k] from lane (i = l&15, k = l>>4).  Lane group
//     g = l>>4 reads 8 CONSECUTIVE k (two ds_read_b128) and feeds them to 8 successive MFMAs, so
//     MFMA q of a chunk contracts k = {8g + q}.  A and B use the same assignment: the result is an
//     exact fp32 fmaf chain in a fixed k order (deterministic, batch-invariant).
//   * every global read of an operand is a BUFFER load through a 128-bit resource descriptor:
//     hardware range checking returns 0 for out-of-range rows / columns / k, so tails need no
//     selects on the data and nothing can fault.  Invalid slots use the sentinel offset kOOB.
//   * epilogue: accumulators -> LDS tile (C/D layout is column-per-lane) -> row-major float4 pass
//     (bias + residual + tau, coalesced 16-byte stores) -> optional segment pass from LDS.
#pragma once

#include <type_traits>

#include "dmpnn_common.hpp"

namespace dmpnn {
namespace gemm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

constexpr int kThreads = 256;
constexpr int BK = 32;       // k chunk
constexpr int BKP = BK + 4;  // padded LDS row (floats): 144-byte rows keep ds_read_b128 16-B aligned
constexpr unsigned kOOB = 0x80000000u;  // byte offset beyond every descriptor (num_records < 2^31)
constexpr int kAtomCache = 255;         // atoms of a tile whose row_ptr is cached in LDS per pass (one load per thread)

enum : int { EPI_PLAIN = 0, EPI_SEG = 1 };

struct GemmK {
    int M, N, K1, K2;
    const float* A1; const float* A2; const float* W; const float* bias; const float* Cadd;
    float* C; float* Zpre;
    const int* gather1; const int* gather2;
    int lda1, lda2, ldw, ldcadd, ldc, ldz;
    unsigned a1_bytes, a2_bytes;  // byte extent of the A1 / A2 source tensors
    int act; float slope; const float* slope_ptr;
    const int* tile_row;          // [n_tiles + 1] first row of every tile, or null (tile t = rows t*BM ..)
    // EPI_SEG
    const int* tile_atom;         // [n_tiles + 1] first atom of every tile
    const int* row_ptr;           // [V + 1]
    const int* revp;              // [E] row of the reverse edge
    float* Mout; int ldm;         // [E, ldm] or null
    float* Sout; int lds;         // [V, lds] or null
    unsigned qmagic;              // ceil(2^32 / qn), qn = N / 4 (unused when qn == 1)
    const int* poison_flags;      // plan header (or null): if (*poison_flags & poison_mask) the output is NaN
    int poison_mask;
};

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned clamp_bytes(int64_t b) {
    return b <= 0 ? 0u : (b > 0x7FFFFFFF ? 0x7FFFFFFFu : (unsigned)b);
}

// ---- G-float buffer loads into one quad (4 floats) -------------------------------------------
// off[s] is the byte offset of sub-group s (G floats) or kOOB.
template <int G>
__device__ __forceinline__ u32x4 load_quad(rsrc_t r, const unsigned (&off)[4 / G]) {
    u32x4 v;
    if constexpr (G == 4) {
        v = __builtin_amdgcn_raw_buffer_load_b128(r, off[0], 0, 0);
    } else if constexpr (G == 2) {
        const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, off[0], 0, 0);
        const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(r, off[1], 0, 0);
        v = u32x4{a.x, a.y, b.x, b.y};
    } else {
        v.x = __builtin_amdgcn_raw_buffer_load_b32(r, off[0], 0, 0);
        v.y = __builtin_amdgcn_raw_buffer_load_b32(r, off[1], 0, 0);
        v.z = __builtin_amdgcn_raw_buffer_load_b32(r, off[2], 0, 0);
        v.w = __builtin_amdgcn_raw_buffer_load_b32(r, off[3], 0, 0);
    }
    return v;
}
// row offset + in-row offset; either being the sentinel makes the slot out of range (reads 0)
__device__ __forceinline__ unsigned join_off(unsigned row_off, unsigned k_off) {
    return ((row_off | k_off) & kOOB) ? kOOB : row_off + k_off;
}
__device__ __forceinline__ float4 as_f4(u32x4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Row-major epilogue over the LDS tile: z = acc(+bias) + Cadd, optional Zpre / C stores,
// y = tau(z) written back to LDS (for the segment pass).  Builds with G >= 2 store 16-byte row
// segments (the dispatcher guarantees N % 4 == 0 and aligned outputs); the G == 1 build is the
// any-shape fallback with scalar stores.
// Two passes on purpose: pass A consumes EVERY prefetched residual register before pass B issues the
// first global store.  gfx950 counts stores in vmcnt, so a load result consumed after a store that
// sits under a (uniform) branch makes hipcc wait for the store itself — one store round trip per
// item, serialised.  With no load consumed after the first store, pass B carries no vmcnt wait.
// SIMPLE: tau in {identity, ReLU, LeakyReLU, PReLU} as ONE branch-free formula
//     y = (z > 0 ? z : neg_slope * z) + 0        neg_slope = 1 / 0 / slope
// (the "+ 0" turns the -0 of 0 * negative into the +0 torch.relu returns; NaN still propagates).
// tanh / ELU take the wave-uniform switch of apply_act4.
template <int RT, int WN, int G, int EPI, bool SIMPLE>
__device__ __forceinline__ void epilogue_rows(const GemmK& g, float* Ct, int rs, int nrows, int col0, int tid,
                                              const float4* cadd_pref, float slope, float neg_slope, bool poison) {
    constexpr int BM = 16 * RT, BN = 64 * WN, LDC = BN + 4, QN = BN / 4;
    constexpr int ITEMS = BM * QN / kThreads;  // exact: BM*QN = 16*RT*16*WN
    const bool has_cadd = g.Cadd != nullptr;
    const float nanv = __int_as_float(0x7fc00000);
    float4 z[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int it = tid + kThreads * j;
        const int r = it / QN, q = it - r * QN;  // QN is a compile-time constant
        z[j] = *reinterpret_cast<const float4*>(Ct + r * LDC + 4 * q);
        if (has_cadd) {  // residual added AFTER the contraction: H0 + W_h(M)   (base.py:141)
            const float4 c = cadd_pref[j];
            z[j].x = c.x + z[j].x; z[j].y = c.y + z[j].y; z[j].z = c.z + z[j].z; z[j].w = c.w + z[j].w;
        }
        if (poison) z[j] = make_float4(nanv, nanv, nanv, nanv);
    }
    // pin: every residual register is consumed HERE, before the first store below
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) asm volatile("" : "+v"(z[j].x), "+v"(z[j].y), "+v"(z[j].z), "+v"(z[j].w));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int it = tid + kThreads * j;
        const int r = it / QN, q = it - r * QN;
        const int col = col0 + 4 * q;
        if (r < nrows && col < g.N) {
            const int64_t row = (int64_t)rs + r;
            float4 y;
            if constexpr (SIMPLE) {
                y.x = (z[j].x > 0.f ? z[j].x : neg_slope * z[j].x) + 0.f;
                y.y = (z[j].y > 0.f ? z[j].y : neg_slope * z[j].y) + 0.f;
                y.z = (z[j].z > 0.f ? z[j].z : neg_slope * z[j].z) + 0.f;
                y.w = (z[j].w > 0.f ? z[j].w : neg_slope * z[j].w) + 0.f;
            } else {
                y = apply_act4(z[j], g.act, slope);  // wave-uniform switch
            }
            if constexpr (G >= 2) {
                if (g.Zpre) *reinterpret_cast<float4*>(g.Zpre + row * g.ldz + col) = z[j];
                if (g.C) *reinterpret_cast<float4*>(g.C + row * g.ldc + col) = y;
            } else {
                const float zz[4] = {z[j].x, z[j].y, z[j].z, z[j].w}, yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (col + t < g.N) {
                        if (g.Zpre) g.Zpre[row * g.ldz + col + t] = zz[t];
                        if (g.C) g.C[row * g.ldc + col + t] = yy[t];
                    }
                }
            }
            if (EPI == EPI_SEG) *reinterpret_cast<float4*>(Ct + r * LDC + 4 * q) = y;
        }
    }
}

template <int RT, int WN, int G, bool HAS_A2, int EPI>
__global__ __launch_bounds__(kThreads) void k_gemm(GemmK g) {
    constexpr int BM = 16 * RT, BN = 64 * WN, LDC = BN + 4, QN = BN / 4;
    constexpr int NS = 4 / G;                                          // loads per quad
    constexpr int SLOTS_A = (BM * (BK / 4) + kThreads - 1) / kThreads; // quads of the A chunk per thread
    constexpr int SLOTS_B = BN * (BK / 4) / kThreads;                  // 2 * WN
    constexpr int ITEMS = BM * QN / kThreads;                          // epilogue quads per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BM][BKP]
    float* Bs = smem + 2 * BM * BKP;   // [2][BN][BKP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int kq = tid & 7;  // k-quad of every staging slot of this thread (kThreads % 8 == 0)
    const int t = blockIdx.x;
    int rs, re;
    if (g.tile_row) { rs = g.tile_row[t]; re = g.tile_row[t + 1]; }
    else { rs = t * BM; re = rs + BM < g.M ? rs + BM : g.M; }
    const int nrows = re - rs;
    if (nrows <= 0) {  // trailing tiles of the launch bound; a fused tile may still own zero-degree atoms
        if (EPI != EPI_SEG) return;
        if (g.tile_atom[t] >= g.tile_atom[t + 1]) return;
    }
    if (nrows > BM) return;  // cannot happen with a valid tile table; never index LDS out of range
    const int col0 = blockIdx.y * BN;
    const int K = g.K1 + g.K2;
    const int n_chunks = (K + BK - 1) / BK;

    // ---- descriptors (built from kernel arguments and blockIdx only: wave-uniform) ----
    const bool gat1 = g.gather1 != nullptr, gat2 = g.gather2 != nullptr;
    const rsrc_t rA1 = gat1 ? make_rsrc(g.A1, g.a1_bytes)
                            : make_rsrc(g.A1 + (int64_t)rs * g.lda1, clamp_bytes(((int64_t)nrows * g.lda1) * 4));
    const rsrc_t rA2 = !HAS_A2 ? rA1
                               : (gat2 ? make_rsrc(g.A2, g.a2_bytes)
                                       : make_rsrc(g.A2 + (int64_t)rs * g.lda2, clamp_bytes(((int64_t)nrows * g.lda2) * 4)));
    const rsrc_t rW = make_rsrc(g.W, clamp_bytes(((int64_t)g.N * g.ldw) * 4));

    // ---- per-slot row offsets (chunk-invariant) ----
    // (the index loads are unconditional — a dummy base when there is no gather — so they are issued
    // back to back instead of one load + wait per uniform branch)
    unsigned offA1[SLOTS_A], offA2[SLOTS_A], offB[SLOTS_B];
    {
        int i1[SLOTS_A], i2[SLOTS_A];
#pragma unroll
        for (int j = 0; j < SLOTS_A; ++j) { i1[j] = 0; i2[j] = 0; }
        if (gat1 || gat2) {  // ONE uniform branch around the whole batch of index loads
            const int* gp1 = gat1 ? g.gather1 : g.gather2;
            const int* gp2 = gat2 ? g.gather2 : g.gather1;
#pragma unroll
            for (int j = 0; j < SLOTS_A; ++j) {
                const int r = (tid + kThreads * j) >> 3;
                const bool ok = r < BM && r < nrows;
                i1[j] = gp1[ok ? rs + r : 0];
                i2[j] = HAS_A2 ? gp2[ok ? rs + r : 0] : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < SLOTS_A; ++j) {
            const int r = (tid + kThreads * j) >> 3;
            const bool ok = r < BM && r < nrows;
            const unsigned row1 = gat1 ? (unsigned)i1[j] : (unsigned)r;
            const unsigned row2 = gat2 ? (unsigned)i2[j] : (unsigned)r;
            offA1[j] = ok ? row1 * (unsigned)g.lda1 * 4u : kOOB;
            offA2[j] = (HAS_A2 && ok) ? row2 * (unsigned)g.lda2 * 4u : kOOB;
        }
    }
#pragma unroll
    for (int j = 0; j < SLOTS_B; ++j) {
        const int col = col0 + ((tid + kThreads * j) >> 3);
        offB[j] = col < g.N ? (unsigned)col * (unsigned)g.ldw * 4u : kOOB;
    }

    // segment metadata of this tile, fetched now (consumed after the contraction): the row of the
    // reverse edge of every tile row, and the first kAtomCache+1 row pointers of the tile's atoms
    int seg_va = 0, seg_vb = 0, seg_rev = 0, seg_rp = 0;
    if (EPI == EPI_SEG) {
        seg_va = g.tile_atom[t];
        seg_vb = g.tile_atom[t + 1];
        const int na0 = seg_vb - seg_va < kAtomCache ? seg_vb - seg_va : kAtomCache;
        const int* rvp = g.Mout ? g.revp : g.row_ptr;  // unconditional load, dummy base without Mout
        seg_rev = rvp[(g.Mout && tid < nrows) ? rs + tid : 0];
        seg_rp = g.row_ptr[seg_va + (tid <= na0 ? tid : 0)] - rs;
    }

    u32x4 stA[SLOTS_A], stB[SLOTS_B];
    auto load_chunk = [&](int c) {
        const int kk = c * BK + kq * 4;
        unsigned k1o[NS], k2o[NS], kbo[NS];  // in-row byte offsets (or kOOB) of the NS sub-groups of this quad
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = kk + s * G;
            k1o[s] = k < g.K1 ? (unsigned)k * 4u : kOOB;
            k2o[s] = (k >= g.K1 && k < K) ? (unsigned)(k - g.K1) * 4u : kOOB;
            kbo[s] = k < K ? (unsigned)k * 4u : kOOB;
        }
#pragma unroll
        for (int j = 0; j < SLOTS_A; ++j) {
            unsigned o1[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) o1[s] = join_off(offA1[j], k1o[s]);
            u32x4 v = load_quad<G>(rA1, o1);
            if constexpr (HAS_A2) {
                unsigned o2[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) o2[s] = join_off(offA2[j], k2o[s]);
                const u32x4 w = load_quad<G>(rA2, o2);
                v = v | w;  // at most one of the two is in range (the other reads 0)
            }
            stA[j] = v;
        }
#pragma unroll
        for (int j = 0; j < SLOTS_B; ++j) {
            unsigned ob[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) ob[s] = join_off(offB[j], kbo[s]);
            stB[j] = load_quad<G>(rW, ob);
        }
    };
    auto store_chunk = [&](int slot) {
        float* Ad = As + slot * BM * BKP;
        float* Bd = Bs + slot * BN * BKP;
#pragma unroll
        for (int j = 0; j < SLOTS_A; ++j) {
            const int f = tid + kThreads * j;
            const int r = f >> 3;
            if (r < BM) *reinterpret_cast<float4*>(Ad + r * BKP + kq * 4) = as_f4(stA[j]);
        }
#pragma unroll
        for (int j = 0; j < SLOTS_B; ++j) {
            const int n = (tid + kThreads * j) >> 3;
            *reinterpret_cast<float4*>(Bd + n * BKP + kq * 4) = as_f4(stB[j]);
        }
    };

    // accumulators start from the bias of their column (every row of the C/D fragment shares it);
    // the load is unconditional (dummy base) and lands under the first chunk's staging loads
    f32x4 acc[RT][WN];
    {
        const float* bias_p = g.bias ? g.bias : g.W;
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) {
            const int col = col0 + wave * (16 * WN) + ct * 16 + li;
            const bool okc = col < g.N && g.bias != nullptr;
            const float braw = bias_p[okc ? col : 0];
            const float bv = okc ? braw : 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = f32x4{bv, bv, bv, bv};
        }
    }

    // residual tile prefetch registers (filled while the LAST chunk is contracted)
    float4 cadd_pref[ITEMS];
    const bool has_cadd = g.Cadd != nullptr;
    const rsrc_t rC = has_cadd ? make_rsrc(g.Cadd + (int64_t)rs * g.ldcadd, clamp_bytes(((int64_t)nrows * g.ldcadd) * 4)) : rW;
    auto load_cadd = [&]() {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = tid + kThreads * j;
            const int r = it / QN, q = it - r * QN;
            const int col = col0 + 4 * q;
            const unsigned rowo = r < nrows ? (unsigned)r * (unsigned)g.ldcadd * 4u : kOOB;
            unsigned o[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int cc = col + s * G;  // N % G == 0 is guaranteed by the dispatcher
                o[s] = (rowo != kOOB && cc < g.N) ? rowo + (unsigned)cc * 4u : kOOB;
            }
            cadd_pref[j] = as_f4(load_quad<G>(rC, o));
        }
    };

    // fragments: lane (li, lg) holds k = 8*lg .. 8*lg+7 of row li of every 16-row tile; two register sets,
    // the set of chunk c+1 is read from LDS while the second half of chunk c is contracted
    auto read_frags = [&](int slot, float (&fa)[RT][8], float (&fb)[WN][8]) {
        const float* Ac = As + slot * BM * BKP;
        const float* Bc = Bs + slot * BN * BKP + wave * (16 * WN) * BKP;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float* p = Ac + (rt * 16 + li) * BKP + lg * 8;
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
    // q outermost: RT*WN independent accumulators between two MFMAs on the same one
    // (dependent-accumulator latency of 16x16x4 f32 is 40 cycles vs 32-cycle issue).
    auto mfma_half = [&](int q0, const float (&fa)[RT][8], const float (&fb)[WN][8]) {
#pragma unroll
        for (int q = q0; q < q0 + 4; ++q)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[rt][q], fb[ct][q], acc[rt][ct], 0, 0, 0);
    };

    // One k-chunk.  At entry: (fa, fb) hold chunk c; the staging registers hold chunk c+1 (if any).
    //   [ds_write chunk c+1 -> LDS[(c+1)&1]] [global loads chunk c+2 -> regs]   <- issued between the
    //   first-half MFMAs (one per MFMA: the matrix pipe never waits for the staging traffic)
    //   barrier; ds_read fragments(c+1) -> (ga, gb);  second-half MFMAs cover the LDS latency
    // HAS_NEXT / HAS_NEXT2 are compile-time so each variant is one straight-line basic block.
    constexpr int N_STAGE_LOADS = (SLOTS_A * (HAS_A2 ? 2 : 1) + SLOTS_B) * NS;
    constexpr int N_STAGE_WRITES = SLOTS_A + SLOTS_B;
    constexpr int N_HALF = 4 * RT * WN;
    auto chunk = [&](auto has_next, auto has_next2, auto is_last, int c, float (&fa)[RT][8], float (&fb)[WN][8],
                     float (&ga)[RT][8], float (&gb)[WN][8]) {
        constexpr bool NEXT = decltype(has_next)::value, NEXT2 = decltype(has_next2)::value, LAST = decltype(is_last)::value;
        if constexpr (NEXT) store_chunk((c + 1) & 1);
        if constexpr (NEXT2) load_chunk(c + 2);
        if constexpr (LAST) {
            if (has_cadd) load_cadd();
        }
        mfma_half(0, fa, fb);
        if constexpr (NEXT) {
            constexpr int NW = N_STAGE_WRITES < N_HALF ? N_STAGE_WRITES : N_HALF;
            constexpr int NL = NEXT2 ? (N_STAGE_LOADS < N_HALF - NW ? N_STAGE_LOADS : N_HALF - NW) : 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 1 DS write
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT) {
            __syncthreads();  // chunk c+1 is in LDS for every wave
            read_frags((c + 1) & 1, ga, gb);
        }
        mfma_half(4, fa, fb);
        __builtin_amdgcn_sched_barrier(0);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    // ---- pipeline prologue ----
    float f0a[RT][8], f0b[WN][8], f1a[RT][8], f1b[WN][8];
    load_chunk(0);
    store_chunk(0);
    if (n_chunks > 1) load_chunk(1);
    __syncthreads();
    read_frags(0, f0a, f0b);
    __builtin_amdgcn_sched_barrier(0);

    // steady state: chunks with two successors, two per trip (static fragment-set indices)
    int c = 0;
    for (; c + 3 < n_chunks; c += 2) {
        chunk(T_{}, T_{}, F_{}, c, f0a, f0b, f1a, f1b);
        chunk(T_{}, T_{}, F_{}, c + 1, f1a, f1b, f0a, f0b);
    }
    // tail: c is even here, 1..3 chunks left
    const int left = n_chunks - c;
    if (left == 3) {
        chunk(T_{}, T_{}, F_{}, c, f0a, f0b, f1a, f1b);
        chunk(T_{}, F_{}, F_{}, c + 1, f1a, f1b, f0a, f0b);
        chunk(F_{}, F_{}, T_{}, c + 2, f0a, f0b, f1a, f1b);
    } else if (left == 2) {
        chunk(T_{}, F_{}, F_{}, c, f0a, f0b, f1a, f1b);
        chunk(F_{}, F_{}, T_{}, c + 1, f1a, f1b, f0a, f0b);
    } else {
        chunk(F_{}, F_{}, T_{}, c, f0a, f0b, f1a, f1b);
    }
    __syncthreads();  // every wave is done with the staging ring before it is reused as the output tile

    // ---- epilogue: accumulators -> LDS tile (reuses the staging ring; the loop's last barrier
    // guarantees every fragment read is done) ----
    float* Ct = smem;  // [BM][LDC]
    {
        // C/D layout of 16x16x4: col = l&15, row = (l>>4)*4 + reg
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ct[(rt * 16 + lg * 4 + r) * LDC + wave * (16 * WN) + ct * 16 + li] = acc[rt][ct][r];
    }
    // segment metadata of this tile (EPI_SEG): reverse rows of the tile's rows
    int* meta = reinterpret_cast<int*>(smem + BM * LDC);  // [BM] revp, then [kAtomCache + 1] row_ptr
    if (EPI == EPI_SEG) {
        if (tid < BM) meta[tid] = seg_rev;
    }
    __syncthreads();

    const float slope = g.slope_ptr ? *g.slope_ptr : g.slope;
    const bool poison = g.poison_flags && (g.poison_flags[0] & g.poison_mask);
    if (g.act == DMPNN_ACT_TANH || g.act == DMPNN_ACT_ELU) {
        epilogue_rows<RT, WN, G, EPI, false>(g, Ct, rs, nrows, col0, tid, cadd_pref, slope, 0.f, poison);
    } else {
        const float neg_slope = g.act == DMPNN_ACT_NONE ? 1.f : (g.act == DMPNN_ACT_RELU ? 0.f : slope);
        epilogue_rows<RT, WN, G, EPI, true>(g, Ct, rs, nrows, col0, tid, cadd_pref, slope, neg_slope, poison);
    }

    if (EPI == EPI_SEG) {
        // ---- segment pass: one (atom, column-quad) item per thread and step ----
        int* rp = meta + BM;
        const int va = seg_va, vb = seg_vb;
        const int qn = (g.N + 3) >> 2;  // live column quads (N % 4 == 0 on this path)
        for (int a0 = va; a0 < vb; a0 += kAtomCache) {
            const int na = vb - a0 < kAtomCache ? vb - a0 : kAtomCache;
            __syncthreads();  // tile (first pass) / previous row_ptr cache (later passes) settled
            if (tid <= na) rp[tid] = a0 == va ? seg_rp : g.row_ptr[a0 + tid] - rs;
            __syncthreads();
            const int n_items = na * qn;
            for (int it = tid; it < n_items; it += kThreads) {
                const int al = qn == 1 ? it : (int)__umulhi((unsigned)it, g.qmagic);  // it / qn
                const int q = it - al * qn;
                const int r0 = rp[al], r1 = rp[al + 1];
                float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int r = r0; r < r1; ++r) {  // increasing edge id: the reference's sequential scatter order
                    const float4 y = *reinterpret_cast<const float4*>(Ct + r * LDC + 4 * q);
                    if (r == r0) S = y;
                    else { S.x += y.x; S.y += y.y; S.z += y.z; S.w += y.w; }
                }
                if (g.Sout) *reinterpret_cast<float4*>(g.Sout + (int64_t)(a0 + al) * g.lds + 4 * q) = S;
                if (g.Mout) {
                    for (int r = r0; r < r1; ++r) {
                        const float4 y = *reinterpret_cast<const float4*>(Ct + r * LDC + 4 * q);
                        *reinterpret_cast<float4*>(g.Mout + (int64_t)meta[r] * g.ldm + 4 * q) =
                            make_float4(S.x - y.x, S.y - y.y, S.z - y.z, S.w - y.w);
                    }
                }
            }
        }
    }
}

template <int RT, int WN>
constexpr size_t lds_bytes(int epi) {
    constexpr int BM = 16 * RT, BN = 64 * WN;
    size_t ring = (size_t)2 * (BM + BN) * BKP * sizeof(float);
    size_t tile = (size_t)BM * (BN + 4) * sizeof(float);
    if (epi == EPI_SEG) tile += (size_t)(BM + kAtomCache + 1) * sizeof(int);
    return ring > tile ? ring : tile;
}

// One launcher per instantiation (defined in the dmpnn_gemm_*.hip units).
template <int RT, int WN, int G, bool HAS_A2, int EPI>
int launch_gemm(const GemmK& g, int n_tiles, hipStream_t s);

#define DMPNN_DEFINE_GEMM(RT, WN, G, A2, EPI)                                                              \
    template <>                                                                                            \
    int launch_gemm<RT, WN, G, A2, EPI>(const GemmK& g, int n_tiles, hipStream_t s) {                      \
        constexpr size_t lds = lds_bytes<RT, WN>(EPI);                                                     \
        static bool attr_set = false;                                                                      \
        if (!attr_set) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<RT, WN, G, A2, EPI>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
            if (e != hipSuccess) {                                                                         \
                set_error("hipFuncSetAttribute(k_gemm<%d,%d,%d>, %zu B LDS): %s", RT, WN, G, lds,          \
                          hipGetErrorString(e));                                                           \
                return DMPNN_EHIP;                                                                         \
            }                                                                                              \
            attr_set = true;                                                                               \
        }                                                                                                  \
        const dim3 grid((unsigned)n_tiles, (unsigned)((g.N + 64 * WN - 1) / (64 * WN)));                   \
        hipLaunchKernelGGL((k_gemm<RT, WN, G, A2, EPI>), grid, dim3(kThreads), lds, s, g);                 \
        DMPNN_CHECK_LAUNCH("k_gemm");                                                                      \
        return DMPNN_OK;                                                                                   \
    }

}  // namespace gemm
}  // namespace dmpnn
