// Per-step contraction on the f16 matrix pipe with the exact 3-term split (see dmpnn_mega16_impl.hpp):
//     C = act([A1[gather] || A2] . W^T + bias + Cadd)        one 48-row tile x up to 512 output columns per workgroup
// The building blocks are those of the whole-forward tile kernel — operand rows gathered ONCE through buffer
// descriptors into registers, 128 columns per group, tile maximum -> exact power-of-two scale, split into the
// LDS tile; weight fragments straight from L2 in the fragment-major pre-split layout; barrier-free MFMA loop;
// groups beyond the first rescale the accumulator by the exact ratio of their scales — followed by a row-major
// epilogue through an fp32 LDS tile (coalesced residual reads and stores).  Serves the per-step routes (large
// batches, d_h up to 512 per column block, wider in several blocks) and the data gradients of the backward pass.
#pragma once

#include <type_traits>

#include "dmpnn_mega16_impl.hpp"
#include "dmpnn_seg16.hpp"

namespace dmpnn {
namespace rows16 {

using gemm::BK;
using gemm::f32x4;
using gemm::kOOB;
using gemm::kThreads;
using gemm::rsrc_t;
using gemm::u32x2;
using gemm::u32x4;
using mega16::h2;
using mega16::h4;
using mega16::h8;
using mega16::scale_for;
using mega16::SplitW;

constexpr int RT = 3, BM = 16 * RT;

struct Rows16K {
    int M, N, K1, K2;
    const float* A1; int lda1; const int* gather1; unsigned a1_bytes;  // a1_bytes: extent of a gathered A1 (else per tile)
    const float* A2; int lda2;
    SplitW W;                 // pre-split weights of ALL N columns (fragment-major), nc = ceil((K1 + K2) / 32)
    const float* bias;        // [N] or null
    const float* Cadd; int ldcadd;
    float* C; int ldc;
    float* Zpre; int ldz;
    int act; float slope; const float* slope_ptr;
    const int* poison_flags; int poison_mask;
    int vec_out;              // C / Zpre / Cadd rows are 16-byte aligned and N % 4 == 0: float4 epilogue
    // SEG (the K1 launch of the per-step fused route on the f16 pipe, dmpnn_step16_impl.hpp): rows are CSR-ordered
    // edges, a tile holds whole destination atoms (plan tile tables), A2 is gathered too, and the epilogue forms the
    // tile's segment sums and writes the first message in split rows (or Mv)
    const int* gather2; unsigned a2_bytes;
    const int* tile_row; const int* tile_atom; const int* row_ptr; const int* revp;
    unsigned char* Mout; int ts; float* Sout; int lds; unsigned qmagic;
    int half_out;  // Mout in half storage (DMPNN_F_STORE16)
    float* M32; int ldm32;  // training: the first message also as fp32 rows (or null)
};

// GC: k-chunks per operand group.  4 (128 columns, 24 operand registers, two workgroups per CU) streams large batches;
// 12 (384 columns: d_h = 300 in ONE group — one maximum, one barrier pair, one 10-chunk barrier-free contraction)
// is the latency-optimal shape when there is about one tile per CU (the data gradients of a 512-molecule batch).
template <int WN, int GC>
constexpr size_t tile_bytes() {  // fp32 epilogue tile and the split operand tile share one region
    return (size_t)BM * (64 * WN + 4) * 4 > (size_t)BM * (GC * 128 + 16) ? (size_t)BM * (64 * WN + 4) * 4 : (size_t)BM * (GC * 128 + 16);
}
template <int WN, int GC, bool SEG = false>
constexpr size_t lds_bytes() {
    return tile_bytes<WN, GC>() + 64 + (SEG ? (size_t)(BM + gemm::kAtomCache + 1) * sizeof(int) : 0);  // + scale words (+ segment metadata)
}

template <int WN, int GC, bool SEG = false>
__global__ __launch_bounds__(kThreads, GC == 4 ? 2 : 1) void k_rows16(Rows16K g) {
    constexpr int BN = 64 * WN, LDC = BN + 4, QN = BN / 4;
    constexpr int ITEMS = BM * QN / kThreads;  // 3 WN
    constexpr int TSG = GC * 128 + 16;         // bytes of one row of the split operand tile (GC chunks)
    constexpr int J = 4 * RT;                  // operand rows per thread
    constexpr int NP = GC / 4;                 // 64-pair passes over a row of the group
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* T = reinterpret_cast<float*>(lds);                      // [BM][LDC] fp32 epilogue tile
    unsigned char* Ag = lds;                                       // [BM][TSG] split operand tile (overlays T)
    unsigned* maxbits = reinterpret_cast<unsigned*>(lds + tile_bytes<WN, GC>());  // [4] rotating tile maxima
    int* meta = reinterpret_cast<int*>(lds + tile_bytes<WN, GC>() + 64);          // SEG: [BM] reverse rows | [kAtomCache + 1] row pointers

    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int li = lane & 15, lg = lane >> 4;
    auto launder = [&]() {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wave = tid >> 6; li = lane & 15; lg = lane >> 4;
    };
    int row0 = blockIdx.x * BM;
    int nrows = g.M - row0 < BM ? g.M - row0 : BM;
    int seg_va = 0, seg_vb = 0;
    if constexpr (SEG) {
        row0 = g.tile_row[blockIdx.x];
        nrows = g.tile_row[blockIdx.x + 1] - row0;
        seg_va = g.tile_atom[blockIdx.x]; seg_vb = g.tile_atom[blockIdx.x + 1];
        if (nrows <= 0 && seg_va >= seg_vb) return;  // trailing slots of the launch bound
        if (nrows < 0 || nrows > BM) return;
    }
    const int col0 = blockIdx.y * BN;            // column block of this workgroup
    const int ncols = g.N - col0 < BN ? g.N - col0 : BN;
    const int K = g.K1 + g.K2;
    const bool poison = g.poison_flags && (g.poison_flags[0] & g.poison_mask);
    const float slope = g.slope_ptr ? *g.slope_ptr : g.slope;

    if (tid < 4) maxbits[tid] = 0u;
    __syncthreads();
    // ---- operand rows of the tile: row wave + 4 j, column pair `lane` of a 128-column group ----
    const bool gathered = g.gather1 != nullptr;
    const rsrc_t rA1 = gathered ? gemm::make_rsrc(g.A1, g.a1_bytes)
                                : gemm::make_rsrc(g.A1 + (long long)row0 * g.lda1, (unsigned)(nrows * g.lda1) * 4u);
    const bool gathered2 = SEG && g.gather2 != nullptr;
    const rsrc_t rA2 = gathered2 ? gemm::make_rsrc(g.A2, g.a2_bytes)
                                 : gemm::make_rsrc(g.A2 ? g.A2 + (long long)row0 * g.lda2 : g.A1, g.A2 ? (unsigned)(nrows * g.lda2) * 4u : 0u);
    unsigned ro1[J], ro2[J];
    {
        int idx[J], idx2[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int r = wave + 4 * j;
            const int rr = (r < nrows && nrows > 0) ? row0 + r : (nrows > 0 ? row0 : 0);
            idx[j] = gathered ? g.gather1[rr] : r;
            idx2[j] = gathered2 ? g.gather2[rr] : r;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool ok = wave + 4 * j < nrows;
            ro1[j] = ok ? (unsigned)idx[j] * (unsigned)g.lda1 * 4u : kOOB;
            ro2[j] = ok ? (unsigned)idx2[j] * (unsigned)g.lda2 * 4u : kOOB;
        }
    }
    // SEG: segment metadata of the tile, fetched now, consumed by the epilogue
    int seg_rev = 0, seg_rp = 0;
    if constexpr (SEG) {
        const int na0 = seg_vb - seg_va < gemm::kAtomCache ? seg_vb - seg_va : gemm::kAtomCache;
        const int* rvp = g.Mout ? g.revp : g.row_ptr;
        seg_rev = rvp[(g.Mout && tid < nrows) ? row0 + tid : 0];
        seg_rp = g.row_ptr[seg_va + (tid <= na0 ? tid : 0)] - row0;
    }
    auto ga_load = [&](int grp, u32x2 (&v)[NP][J]) {
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            const int k = grp * (GC * 32) + (pp * 64 + lane) * 2;
            const unsigned k1o = k < g.K1 ? (unsigned)k * 4u : kOOB;
            const unsigned k2o = (k >= g.K1 && k < K) ? (unsigned)(k - g.K1) * 4u : kOOB;
#pragma unroll
            for (int j = 0; j < J; ++j)
                v[pp][j] = __builtin_amdgcn_raw_buffer_load_b64(rA1, gemm::join_off(ro1[j], k1o), 0, 0) |
                           __builtin_amdgcn_raw_buffer_load_b64(rA2, gemm::join_off(ro2[j], k2o), 0, 0);
        }
    };
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
    auto tile_scale = [&](float local_max) -> float {  // (see dmpnn_mega16_impl.hpp: rotating slots, one barrier)
        const int slot = scale_phase & 3;
        local_max = wave_max(local_max);
        if (lane == 0) atomicMax(&maxbits[slot], __float_as_uint(local_max));
        __syncthreads();
        const float mx = __uint_as_float(maxbits[slot]);
        if (tid == 0) maxbits[(slot + 2) & 3] = 0u;
        ++scale_phase;
        return scale_for(mx);
    };

    f32x4 acc[RT][WN];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- weight fragments of this wave's column tiles: [column tile][chunk][hi|lo][lane][16 B] ----
    const int NT = (g.N + 15) / 16;
    const rsrc_t rW = gemm::make_rsrc(g.W.p, (unsigned)(NT * g.W.nc * 2048));
    unsigned offB[WN];
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int tile = col0 / 16 + wave * WN + ct;
        offB[ct] = tile < NT ? (unsigned)tile * (unsigned)(g.W.nc * 2048) + (unsigned)lane * 16u : kOOB;
    }
    auto load_bfrags = [&](int c, h8 (&bh)[WN], h8 (&bl)[WN]) {
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) {
            const unsigned o = offB[ct] == kOOB ? kOOB : offB[ct] + (unsigned)c * 2048u;
            bh[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o, 0, 0));
            bl[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o == kOOB ? kOOB : o + 1024u, 0, 0));
        }
    };
    // one group of n_chunks (<= GC) k-chunks, weight chunks wc0 ..: barrier-free MFMA loop on the split tile Ag.
    // One compact loop body (two chunks, ping-pong fragment registers): a workgroup runs one tile, so every
    // instruction is fetched cold — code size is latency.  Fragments of chunk c+1 (A, from LDS) and c+2 (weights,
    // from L2) are fetched under the MFMAs of chunk c; reads past the group are clamped / out of range (0).
    auto contract = [&](int n_chunks, int wc0) {
        h8 ah[2][RT], al[2][RT], bh[2][WN], bl[2][WN];
        auto read_afrags = [&](int c, h8 (&xh)[RT], h8 (&xl)[RT]) {
            const int cc = c < n_chunks ? c : n_chunks - 1;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const unsigned char* p = Ag + (rt * 16 + li) * TSG + cc * 128 + lg * 16;
                xh[rt] = *reinterpret_cast<const h8*>(p);
                xl[rt] = *reinterpret_cast<const h8*>(p + 64);
            }
        };
        auto ring = [&](int c, h8 (&xh)[RT], h8 (&xl)[RT], h8 (&nh)[RT], h8 (&nl)[RT]) {
            const bool more = c + 1 < n_chunks;
#pragma unroll
            for (int ct = 0; ct < WN; ++ct) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[rt], bh[0][ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[rt], bl[0][ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[rt], bh[0][ct], acc[rt][ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned o = (more && offB[ct] != kOOB) ? offB[ct] + (unsigned)(wc0 + c + 1) * 2048u : kOOB;
                bh[0][ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o, 0, 0));
                bl[0][ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o == kOOB ? kOOB : o + 1024u, 0, 0));
                if (ct == (WN > 1 ? WN - 2 : 0)) read_afrags(c + 1, nh, nl);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        load_bfrags(wc0, bh[0], bl[0]);
        __syncthreads();  // the split tile is complete
        launder();
        read_afrags(0, ah[0], al[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma nounroll
        for (int c = 0; c < n_chunks; c += 2) {
            ring(c, ah[0], al[0], ah[1], al[1]);
            if (c + 1 < n_chunks) ring(c + 1, ah[1], al[1], ah[0], al[0]);
        }
    };

    // ---- groups of 128 operand columns: registers -> maximum -> scale -> split tile -> MFMAs ----
    const int n_groups = (g.W.nc + GC - 1) / GC;
    u32x2 v[NP][J];
    ga_load(0, v);
    float s_prev = 0.f;
#pragma nounroll
    for (int grp = 0; grp < n_groups; ++grp) {
        float mx = 0.f;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int j = 0; j < J; ++j) mx = fmaxf(mx, fmaxf(fabsf(__uint_as_float(v[pp][j].x)), fabsf(__uint_as_float(v[pp][j].y))));
        const float s = tile_scale(mx);  // (barrier: every wave is past its reads of the LDS tile)
        if (s_prev != 0.f && s_prev != s) {
            const float f = s / s_prev;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[rt][ct][r] *= f;
        }
        s_prev = s;
        launder();
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                unsigned hi, lo;
                mega16::split2(__uint_as_float(v[pp][j].x), __uint_as_float(v[pp][j].y), s, hi, lo);
                unsigned char* p = Ag + (wave + 4 * j) * TSG + (pp * 4 + (lane >> 4)) * 128 + (lane & 15) * 4;
                *reinterpret_cast<unsigned*>(p) = hi;
                *reinterpret_cast<unsigned*>(p + 64) = lo;
            }
        if (grp + 1 < n_groups) ga_load(grp + 1, v);  // the next group's rows are in flight under this group's MFMAs
        const int ncg = g.W.nc - grp * GC < GC ? g.W.nc - grp * GC : GC;
        contract(ncg, grp * GC);
    }

    // ---- epilogue: split domain -> fp32, bias; row-major pass through the LDS tile ----
    launder();
    const float inv_sA = 1.f / s_prev;
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int col = col0 + wave * (16 * WN) + ct * 16 + li;
        const bool okc = col < g.N;
        const float isw = g.W.inv_scale[okc ? col : 0] * inv_sA;
        const float bv = (okc && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rt][ct][r] = acc[rt][ct][r] * isw + bv;
    }
    __syncthreads();  // every wave is done with the operand tile the epilogue tile overlays
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int cl = wave * (16 * WN) + ct * 16 + li;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(rt * 16 + lg * 4 + r) * LDC + cl] = acc[rt][ct][r];
    }
    __syncthreads();
    const bool vec_out = g.vec_out != 0;
    const float nanv = __int_as_float(0x7fc00000);
    if (vec_out) {
        // two passes: every residual quad is requested first (buffer loads, out of range -> 0: no load sits under a
        // branch, so they are all in flight together), then the arithmetic and the stores
        const rsrc_t rC = gemm::make_rsrc(g.Cadd ? g.Cadd + (long long)row0 * g.ldcadd + col0 : g.A1,
                                          g.Cadd ? (unsigned)(((nrows - 1) * g.ldcadd + ncols) * 4) : 0u);
        float4 res[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = tid + kThreads * j;
            const int r = it / QN, q = it - r * QN;
            const unsigned off = (r < nrows && 4 * q < ncols) ? (unsigned)(r * g.ldcadd + 4 * q) * 4u : kOOB;
            res[j] = gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(rC, off, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = tid + kThreads * j;
            const int r = it / QN, q = it - r * QN;
            const int c = 4 * q;
            float4 z = *reinterpret_cast<const float4*>(T + r * LDC + c);
            z.x += res[j].x; z.y += res[j].y; z.z += res[j].z; z.w += res[j].w;
            if (poison) z = make_float4(nanv, nanv, nanv, nanv);
            if (r < nrows && c < ncols) {
                const long long row = row0 + r;
                if (g.Zpre) *reinterpret_cast<float4*>(g.Zpre + row * g.ldz + col0 + c) = z;
                const float4 y = apply_act4(z, g.act, slope);
                if (g.C) *reinterpret_cast<float4*>(g.C + row * g.ldc + col0 + c) = y;
                if constexpr (SEG) *reinterpret_cast<float4*>(T + r * LDC + c) = y;  // tau(z): what the segment sums run over
            }
        }
        if constexpr (SEG) {
            if (tid < BM) meta[tid] = seg_rev;
            step16::SegOut o;
            o.row_ptr = g.row_ptr; o.revp = g.revp; o.Mout = g.Mout; o.ts = g.ts; o.Sout = g.Sout; o.lds = g.lds; o.N = g.N; o.half = g.half_out; o.SoutS = nullptr; o.tss = 0; o.M32 = g.M32; o.ldm32 = g.ldm32;
            step16::seg_epilogue<LDC, BN / 4, kThreads>(o, T, meta, row0, nrows, seg_va, seg_vb, seg_rp, poison, g.qmagic, tile_scale);
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = tid + kThreads * j;
            const int r = it / QN, q = it - r * QN;
            const int c = 4 * q;
            if (r < nrows && c < ncols) {
                const long long row = row0 + r;
                const float4 z4 = *reinterpret_cast<const float4*>(T + r * LDC + c);
                const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (c + t < ncols) {
                        float v = zz[t] + (g.Cadd ? g.Cadd[row * g.ldcadd + col0 + c + t] : 0.f);
                        if (poison) v = nanv;
                        if (g.Zpre) g.Zpre[row * g.ldz + col0 + c + t] = v;
                        if (g.C) g.C[row * g.ldc + col0 + c + t] = apply_act(v, g.act, slope);
                    }
                }
            }
        }
    }
}

template <int WN, int GC, bool SEG = false>
int launch_rows16(const Rows16K& g, int row_tiles, int col_blocks, hipStream_t s);

#define DMPNN_DEFINE_ROWS16(WN, GC) DMPNN_DEFINE_ROWS16_X(WN, GC, false)
#define DMPNN_DEFINE_ROWS16_X(WN, GC, SEG)                                                                 \
    template <>                                                                                            \
    int launch_rows16<WN, GC, SEG>(const Rows16K& g, int row_tiles, int col_blocks, hipStream_t s) {       \
        constexpr size_t lds = lds_bytes<WN, GC, SEG>();                                                   \
        static bool attr_set = false;                                                                      \
        if (!attr_set) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows16<WN, GC, SEG>),      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
            if (e != hipSuccess) {                                                                         \
                set_error("hipFuncSetAttribute(k_rows16<%d>, %zu B LDS): %s", WN, lds, hipGetErrorString(e)); \
                return DMPNN_EHIP;                                                                         \
            }                                                                                              \
            attr_set = true;                                                                               \
        }                                                                                                  \
        hipLaunchKernelGGL((k_rows16<WN, GC, SEG>), dim3((unsigned)row_tiles, (unsigned)col_blocks), dim3(kThreads), lds, s, g); \
        DMPNN_CHECK_LAUNCH("k_rows16");                                                                    \
        return DMPNN_OK;                                                                                   \
    }

}  // namespace rows16
}  // namespace dmpnn
