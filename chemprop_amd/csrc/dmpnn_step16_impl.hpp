// Per-step FUSED depth update on the f16 matrix pipe (exact 3-term operand split, fp32 accumulate) for batches of ANY
// molecule size — the route BASELINE configs 3-5 take (ZINC-sized, 40-atom and reaction graphs do not fit the tiles of
// the whole-forward kernel):
//
//     H' = tau(H0 + W_h M (+ b_h))                    base.py:135-141      one 48-row tile of WHOLE destination atoms
//     M_next[rev r] = S[dst r] - H'[r]   or   Mv = S   mixins.py:11-18 / base.py:208-211   (S: the tile's own segment sums)
//
// in ONE launch per depth step and with 3 row passes of HBM traffic per step (read M, read H0, write M_next) instead of
// the 5 of the per-step general route (contraction + stand-alone segment kernel).  Edge tensors live in the plan's
// CSR-row order (rows of a destination atom contiguous), as in the fp32 fused route (dmpnn_gemm_impl.hpp, EPI_SEG).
//
// What makes it stream (the general route's k_rows16 sat at 0.3 of HBM: operand rows went global -> registers ->
// maximum -> barrier -> split -> LDS, three dependent phases per 128 columns and little in flight):
//   * the message tensor is kept BETWEEN steps in the split form the matrix pipe wants: row = nc chunks of
//     [hi 32 halfs | lo 32 halfs] + a 16-byte tail holding the row's power-of-two scale (x s = hi + lo) — 4 bytes per
//     element like fp32, written by the PRODUCER's epilogue (which has the values in registers anyway), the scale being
//     the maximum of the producer's tile;
//   * that row format IS the LDS layout of the contraction's A tile (row stride 128 nc + 16: conflict-free ds_read_b128),
//     and a tile's rows are contiguous in memory: the consumer fetches its whole operand tile with LDS-DMA
//     (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction), no register staging, no VALU, one barrier;
//   * the residual H0 is requested at kernel entry straight into the accumulator fragments (acc = H0 s_r s_W, exact
//     power-of-two scalings), so the whole tile input (2 x 60 KB) is in flight before the first MFMA and two workgroups
//     per CU keep ~240 KB per CU outstanding;
//   * rows carry their own scale (s_r of the producing tile), the MFMA result of row r is in scale s_r s_W[col]:
//     one multiply per element in the epilogue.
#pragma once

#include "dmpnn_rows16_impl.hpp"
#include "dmpnn_seg16.hpp"

namespace dmpnn {
namespace step16 {

using gemm::f32x4;
using gemm::kAtomCache;
using gemm::kOOB;
using gemm::kThreads;
using gemm::rsrc_t;
using mega16::h4;
using mega16::h8;
using mega16::scale_for;
using mega16::split4;
using mega16::SplitW;

constexpr int RT = 3;
static_assert(BM == 16 * RT, "48-row tiles");

// bytes of one row of a split edge tensor of d_h columns (the A-tile row of the kernels: 64-column granules + 16 B tail)
__host__ __device__ constexpr int split_row_bytes(int d_h) { return ((d_h + 63) / 64) * 256 + 16; }

struct Step16K {
    int M, N;
    const int* tile_row; const int* tile_atom; const int* row_ptr; const int* revp;
    const unsigned char* A; int ts;       // split operand rows [M][ts]
    SplitW W; const float* bias;
    const float* Cadd; int ldcadd;        // residual H0 [M][ldcadd] fp32
    unsigned char* Mout; float* Sout; int lds;
    int act; float slope; const float* slope_ptr;
    const int* poison_flags; int poison_mask;
    unsigned qmagic;
};

template <int WN>
constexpr size_t lds_bytes() {
    return (size_t)BM * (64 * WN * 4 + 16) + (size_t)(BM + kAtomCache + 1) * sizeof(int) + 64;
}

template <int WN>
__global__ __launch_bounds__(kThreads, 2) void k_step16(Step16K g) {
    constexpr int BN = 64 * WN, LDC = BN + 4, TS = BN * 4 + 16;
    static_assert(LDC * 4 == TS, "the fp32 epilogue tile overlays the split operand tile row for row");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* Ag = lds;                                    // [BM][TS] split operand tile (DMA target)
    float* T = reinterpret_cast<float*>(lds);                   // [BM][LDC] fp32 epilogue tile (overlays it)
    int* meta = reinterpret_cast<int*>(lds + (size_t)BM * TS);  // [BM] reverse rows | [kAtomCache + 1] row pointers
    unsigned* maxbits = reinterpret_cast<unsigned*>(meta + BM + kAtomCache + 1);

    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int li = lane & 15, lg = lane >> 4;
    auto launder = [&]() {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wave = tid >> 6; li = lane & 15; lg = lane >> 4;
    };
    const int t = blockIdx.x;
    const int rs = g.tile_row[t], re = g.tile_row[t + 1];
    const int va = g.tile_atom[t], vb = g.tile_atom[t + 1];
    const int nrows = re - rs;
    if (nrows <= 0 && va >= vb) return;   // trailing slots of the launch bound
    if (nrows < 0 || nrows > BM) return;  // (cannot happen with a valid tile table)
    const bool poison = g.poison_flags && (g.poison_flags[0] & g.poison_mask);
    if (tid < 4) maxbits[tid] = 0u;

    // ---- everything the tile needs is requested now: operand rows by LDS-DMA, residual into the accumulators ----
    {
        const unsigned nbytes = (unsigned)(nrows * g.ts);
        const rsrc_t rA = gemm::make_rsrc(g.A + (long long)rs * g.ts, nbytes);
        const int n_inst = (int)((nbytes + 1023u) >> 10);
#ifndef DMPNN_X_NOLOAD
        for (int i = wave; i < n_inst; i += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(Ag + i * 1024), 16,
                                                     (unsigned)(i * 1024 + lane * 16), 0, 0, 0);
#endif
    }
    f32x4 acc[RT][WN];
    {
        const rsrc_t rC = gemm::make_rsrc(g.Cadd ? g.Cadd + (long long)rs * g.ldcadd : reinterpret_cast<const float*>(g.A),
                                          (g.Cadd && nrows > 0) ? (unsigned)(((nrows - 1) * g.ldcadd + g.N) * 4) : 0u);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rt * 16 + lg * 4 + r, col = wave * (16 * WN) + ct * 16 + li;
#ifdef DMPNN_X_NOLOAD
                    const unsigned off = kOOB;
#else
                    const unsigned off = (row < nrows && col < g.N) ? (unsigned)(row * g.ldcadd + col) * 4u : kOOB;
#endif
                    acc[rt][ct][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rC, off, 0, 0));
                }
    }
    // segment metadata (consumed by the epilogue)
    const int na0 = vb - va < kAtomCache ? vb - va : kAtomCache;
    const int* rvp = g.Mout ? g.revp : g.row_ptr;
    const int seg_rev = rvp[(g.Mout && tid < nrows) ? rs + tid : 0];
    const int seg_rp = g.row_ptr[va + (tid <= na0 ? tid : 0)] - rs;
    const float slope = g.slope_ptr ? *g.slope_ptr : g.slope;

    // ---- weight fragments: [column tile][chunk][hi|lo][lane][16 B], straight from L2 ----
    const int NT = (g.N + 15) / 16;
    const rsrc_t rW = gemm::make_rsrc(g.W.p, (unsigned)(NT * g.W.nc * 2048));
    unsigned offB[WN];
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int tile = wave * WN + ct;
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
    h8 ah[2][RT], al[2][RT], bh[2][WN], bl[2][WN];
    load_bfrags(0, bh[0], bl[0]);
    load_bfrags(1, bh[1], bl[1]);
    float isw[WN], bv[WN];
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int col = wave * (16 * WN) + ct * 16 + li;
        const bool okc = col < g.N;
        isw[ct] = g.W.inv_scale[okc ? col : 0];
        bv[ct] = (okc && g.bias) ? g.bias[col] : 0.f;
    }
    __syncthreads();  // the operand tile has landed (the barrier's release waits for the DMA: vmcnt(0))
    launder();
    // the rows' own scales (tail of every operand row); rows beyond the tile: 1
    float sr[RT][4], isr[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + lg * 4 + r;
            const float v = *reinterpret_cast<const float*>(Ag + row * TS + BN * 4);
            sr[rt][r] = (row < nrows && v > 0.f && v < 3.0e38f) ? v : 1.f;
            isr[rt][r] = 1.f / sr[rt][r];
        }
    // residual into the split domain of its row and column: acc = H0 s_r s_W (powers of two: exact)
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const float sw = 1.f / isw[ct];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rt][ct][r] *= sr[rt][r] * sw;
    }
    // ---- barrier-free MFMA loop on the static operand tile ----
    const int n_chunks = g.W.nc;
    auto read_afrags = [&](int c, h8 (&xh)[RT], h8 (&xl)[RT]) {
        const int cc = c < n_chunks ? c : n_chunks - 1;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned char* p = Ag + (rt * 16 + li) * TS + cc * 128 + lg * 16;
            xh[rt] = *reinterpret_cast<const h8*>(p);
            xl[rt] = *reinterpret_cast<const h8*>(p + 64);
        }
    };
    auto step = [&](int c, h8 (&xh)[RT], h8 (&xl)[RT], h8 (&yh)[WN], h8 (&yl)[WN], h8 (&nh)[RT], h8 (&nl)[RT]) {
#pragma unroll
        for (int ct = 0; ct < WN; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[rt], yh[ct], acc[rt][ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_afrags(c + 1, nh, nl);
#pragma unroll
        for (int ct = 0; ct < WN; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[rt], yl[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < WN; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[rt], yh[ct], acc[rt][ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_bfrags(c + 2, yh, yl);
    };
    read_afrags(0, ah[0], al[0]);
    __builtin_amdgcn_sched_barrier(0);
#ifndef DMPNN_X_NOMFMA
#pragma nounroll
    for (int c = 0; c < n_chunks; c += 2) {
        step(c, ah[0], al[0], bh[0], bl[0], ah[1], al[1]);
        if (c + 1 < n_chunks) step(c + 1, ah[1], al[1], bh[1], bl[1], ah[0], al[0]);
    }
#endif

    // ---- epilogue: split domain -> fp32, tau; tile -> segment sums -> next message (split rows) / Mv ----
    launder();
#pragma unroll
    for (int ct = 0; ct < WN; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = acc[rt][ct][r] * (isw[ct] * isr[rt][r]) + bv[ct];
                acc[rt][ct][r] = apply_act(z, g.act, slope);
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
    if (tid < BM) meta[tid] = seg_rev;
    int scale_phase = 0;
    auto tile_scale = [&](float local_max) -> float {
        int u = (int)__float_as_uint(local_max);
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0xB1, 0xf, 0xf, true));
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0x4E, 0xf, 0xf, true));
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0x141, 0xf, 0xf, true));
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0x140, 0xf, 0xf, true));
        const int m = max(max(__builtin_amdgcn_readlane(u, 0), __builtin_amdgcn_readlane(u, 16)),
                          max(__builtin_amdgcn_readlane(u, 32), __builtin_amdgcn_readlane(u, 48)));
        if ((threadIdx.x & 63) == 0) atomicMax(&maxbits[scale_phase & 3], (unsigned)m);
        __syncthreads();
        const float mxv = __uint_as_float(maxbits[scale_phase & 3]);
        ++scale_phase;
        return scale_for(mxv);
    };
    SegOut o;
    o.row_ptr = g.row_ptr; o.revp = g.revp; o.Mout = g.Mout; o.ts = g.ts; o.Sout = g.Sout; o.lds = g.lds;
    o.N = g.N;
#ifndef DMPNN_X_NOSEG
    seg_epilogue<LDC, BN / 4>(o, T, meta, rs, nrows, va, vb, seg_rp, poison, g.qmagic, tile_scale);
#endif
}

template <int WN>
int launch_step16(const Step16K& g, int n_tiles, hipStream_t s);

#define DMPNN_DEFINE_STEP16(WN)                                                                            \
    template <>                                                                                            \
    int launch_step16<WN>(const Step16K& g, int n_tiles, hipStream_t s) {                                  \
        constexpr size_t lds = lds_bytes<WN>();                                                            \
        static bool attr_set = false;                                                                      \
        if (!attr_set) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step16<WN>),               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
            if (e != hipSuccess) {                                                                         \
                set_error("hipFuncSetAttribute(k_step16<%d>, %zu B LDS): %s", WN, lds, hipGetErrorString(e)); \
                return DMPNN_EHIP;                                                                         \
            }                                                                                              \
            attr_set = true;                                                                               \
        }                                                                                                  \
        hipLaunchKernelGGL((k_step16<WN>), dim3((unsigned)n_tiles), dim3(kThreads), lds, s, g);            \
        DMPNN_CHECK_LAUNCH("k_step16");                                                                    \
        return DMPNN_OK;                                                                                   \
    }

}  // namespace step16
}  // namespace dmpnn
