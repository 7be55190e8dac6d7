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

#include <stdlib.h>

#include "dmpnn_rows16_impl.hpp"
#include "dmpnn_seg16.hpp"

namespace dmpnn {
namespace step16 {

using gemm::f32x4;
using gemm::kAtomCache;
using gemm::kOOB;
using gemm::rsrc_t;
using mega16::h4;
using mega16::h8;
using mega16::scale_for;
using mega16::split4;
using mega16::SplitW;

constexpr int RT = 3;
static_assert(BM == 16 * RT, "48-row tiles");

// 1 / s for a power of two s (every scale of the split format is one: scale_for, k_split_weights) — EXACT, one integer subtraction.
// (round 5: the epilogue formed acc * (isw / s_r) per element: 60 IEEE divisions of ~10 VALU instructions each per lane and tile — a
//  third of the kernel's VALU instructions, found in the instruction histogram of the ISA)
#if defined(DMPNN_STEP16_IEEE_DIV)   // (A/B build: the divisions as rounds 2-4 had them)
__device__ __forceinline__ float rcp_pow2(float s) { return 1.f / s; }
#else
__device__ __forceinline__ float rcp_pow2(float s) { return __uint_as_float(0x7F000000u - __float_as_uint(s)); }
#endif
constexpr int kXChunks = 8;  // most chunks (of 32 columns) of the second operand: d_v + d_e <= 256

// output columns a workgroup covers for d_h: 4 waves x 16 WN (d_h <= 320) or 8 waves x 16 WN (d_h <= 640)
__host__ __device__ constexpr int block_cols(int d_h) { return d_h <= 320 ? ((d_h + 63) / 64) * 64 : ((d_h + 127) / 128) * 128; }
// bytes of one row of a split edge tensor of d_h columns (the A-tile row of the kernels: whole chunk pairs + 16 B tail)
__host__ __device__ constexpr int split_row_bytes(int d_h) { return block_cols(d_h) * 4 + 16; }
// ... in half storage (DMPNN_F_STORE16): the hi halfs alone, 2 bytes per element + the 16-byte tail
__host__ __device__ constexpr int half_row_bytes(int d_h) { return block_cols(d_h) * 2 + 16; }
// ... of a split operand of K columns that is only ever READ as an operand (the gathered K1 input): whole 32-column chunks
__host__ __device__ constexpr int split_operand_bytes(int K) { return ((K + 31) / 32) * 128 + 16; }

struct Step16K {
    int M, N;
    const int* tile_row; const int* tile_atom; const int* row_ptr; const int* revp;
    const unsigned char* A; int ts;       // split operand rows [M][ts]: W.nc chunks (+ padding) + the 16-byte tail at ts - 16
    SplitW W; const float* bias;
    const float* Cadd; int ldcadd;        // residual H0 [M][ldcadd] fp32 (or null)
    // second operand (or null): the K1 operand [V[src] || E] of the same rows, exactly split once per forward (k_split_rows):
    // z = W2 x (+ bias2) + W A (+ bias) — the residual H0 = W_i x + b_i is RECOMPUTED per step from 400-byte rows instead of
    // being written once and read back every step as 1 200-byte fp32 rows through 60 scattered 4-byte loads per lane
    // The x tile does not go through LDS (62 KB operand tile + 19 KB would push two workgroups past the CU's 160 KB): every
    // lane fetches its own A fragments of the <= kXChunks chunks straight from the rows at kernel entry (16-byte loads).
    const unsigned char* A2; int ts2;     // rows [M][ts2]
    SplitW W2; const float* bias2;
    float* Zpre; int ldz;                 // pre-activation rows [M][ldz] fp32 to store (K1: H0), or null
    // round 5: H0 kept in the layout of the accumulator fragments — row QUADS [quad][BN columns][4 rows] fp32, the quads of tile t from
    // ((first row + 3) >> 2) + t on (every tile's range is its own: ceil(nrows / 4) quads, at most one quad of padding per tile) —
    // written by K1 straight from its registers (H0q_out), read back by every depth step as RT * WN coalesced 16-byte loads per lane
    // (H0q_in: four 256-byte segments per wave instruction) instead of recomputing W_i x per step or gathering fp32 rows by the word
    float* H0q_out; const float* H0q_in;
    unsigned char* Mout; float* Sout; int lds;
    unsigned char* SoutS;                 // the per-atom sums as split rows [V][TSO] instead of fp32 Sout (or null)
    int uniform;                          // 1: no tile table — tile t = rows 48 t .. (the finalize over atoms: no segments)
    float* Yout; int ldy;                 // 1: plain epilogue — tau(z) rows [M][ldy] fp32, nothing else (or null)
    float* Hout; int ldho;                // training: H' = tau(z) rows [M][ldho] fp32 kept for the backward pass (or null)
    float* M32; int ldm32;                // training: the next message also as fp32 rows (the weight gradients' operand) (or null)
    unsigned char* bits; int bstride;     // lean training (ReLU-class activation): [tau(z) > 0] of every element, one bit each, rows [M][bstride]
                                          // bytes (bit c & 7 of byte c >> 3) — all the backward step kernel needs of H' (or null)
    int act; float slope; const float* slope_ptr;
    const int* poison_flags; int poison_mask;
    unsigned qmagic;
    int half_out;                         // Mout in half storage (DMPNN_F_STORE16); the operand's own format is the template parameter HIN
    int tile_bytes;                       // bytes of the LDS region shared by the operand tile and the fp32 epilogue tile
    long long* dbg;                       // optional [16] cycle stamps of one workgroup (dmpnn_debug_timestamps), else null
};

template <int WN, int NW>
__host__ __device__ constexpr size_t meta_bytes() { return (size_t)(BM + kAtomCache + 1) * sizeof(int) + 64; }

// HIN: the operand rows are in half storage ([hi 32 halfs] chunks; two MFMA passes a_hi (b_hi + b_lo) instead of three)
// XP:  the second operand x is there (g.A2) and the fp32 residual is not (g.Cadd): separate instantiations, so that neither
//      path carries the other's registers (the kernel sits at the 256-register limit of two workgroups per CU)
template <int WN, int NW, bool HIN, bool XP>
__global__ __launch_bounds__(64 * NW, 2) void k_step16(Step16K g) {
    constexpr int NT = 64 * NW;
    constexpr int BN = 16 * WN * NW, LDC = BN + 4, TSO = BN * 4 + 16;
    static_assert(LDC * 4 == TSO, "rows of the fp32 epilogue tile and of the split output tensor have the same stride");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* Ag = lds;                                    // [BM][ts] split operand tile (DMA target)
    float* T = reinterpret_cast<float*>(lds);                   // [BM][LDC] fp32 epilogue tile (overlays it)
    int* meta = reinterpret_cast<int*>(lds + g.tile_bytes);     // [BM] reverse rows | [kAtomCache + 1] row pointers
    unsigned* maxbits = reinterpret_cast<unsigned*>(meta + BM + kAtomCache + 1);

    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int li = lane & 15, lg = lane >> 4;
    auto launder = [&]() {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wave = tid >> 6; li = lane & 15; lg = lane >> 4;
    };
    int n_stamp = 0;
    auto stamp = [&]() {
        if (g.dbg && blockIdx.x == 37 && threadIdx.x == 0 && n_stamp < 16) g.dbg[n_stamp] = (long long)__builtin_readcyclecounter();
        ++n_stamp;
    };
    stamp();  // 0 entry
    const int t = blockIdx.x;
    const int rs = g.uniform ? BM * t : g.tile_row[t], re = g.uniform ? (BM * t + BM < g.M ? BM * t + BM : g.M) : g.tile_row[t + 1];
    const int va = g.uniform ? 0 : g.tile_atom[t], vb = g.uniform ? 0 : g.tile_atom[t + 1];
    const int nrows = re - rs;
    if (nrows <= 0 && va >= vb) return;   // trailing slots of the launch bound
    if (nrows < 0 || nrows > BM) return;  // (cannot happen with a valid tile table)
    const bool poison = g.poison_flags && (g.poison_flags[0] & g.poison_mask);
    if (tid < 4) maxbits[tid] = 0u;
    const int TS = g.ts;

    // ---- everything the tile needs is requested now: operand rows by LDS-DMA, residual into the accumulators ----
    const int TS2 = g.ts2;
    // (DMPNN_STEP16_STAMPS2, DMPNN_STEP16_DIAG_*: measurement builds of round 5 — scripts/probe_stamps_step16b.py, scripts/gpu_r5_diag.sh;
    //  what they found is in DESIGN.md section 4 "k_step16: where a tile's 31 k cycles go")
#if defined(DMPNN_STEP16_STAMPS2)
    stamp();  // a: tile table read, nothing requested yet
#endif
    if (g.A) {
        const unsigned nbytes = (unsigned)(nrows * TS);
        const rsrc_t rA = gemm::make_rsrc(g.A + (long long)rs * TS, nbytes);
        const int n_inst = (int)((nbytes + 1023u) >> 10);
        for (int i = wave; i < n_inst; i += NW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(Ag + i * 1024), 16,
                                                     (unsigned)(i * 1024 + lane * 16), 0, 0, 0);
    }
#if defined(DMPNN_STEP16_STAMPS2)
    stamp();  // b: DMA issued
#endif
    // second operand: this lane's A fragments (row rt * 16 + li, 8 reduction columns from lg * 8 of a chunk) of chunks 0 and 1,
    // and the rows' scales; chunk c + 2 is fetched into the registers of chunk c behind its MFMAs
    h8 xh[2][RT], xl[2][RT];
    float s2 = 1.f;  // (k_split_rows scales a whole tile at once: every row's tail holds the same value)
    const rsrc_t rX = gemm::make_rsrc(XP ? g.A2 + (long long)rs * TS2 : g.A, XP ? (unsigned)(nrows * TS2) : 0u);
    auto load_xfrags = [&](int c, h8 (&h)[RT], h8 (&l)[RT]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = rt * 16 + li;
            const unsigned off = (c < g.W2.nc && row < nrows) ? (unsigned)(row * TS2 + c * 128 + lg * 16) : kOOB;
            h[rt] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rX, off, 0, 0));
            l[rt] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rX, off == kOOB ? kOOB : off + 64u, 0, 0));
        }
    };
    if constexpr (XP) {
        load_xfrags(0, xh[0], xl[0]);
        load_xfrags(1, xh[1], xl[1]);
        const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, nrows > 0 ? (unsigned)(TS2 - 16) : kOOB, 0, 0));
        s2 = (v2 > 0.f && v2 < 3.0e38f && (__float_as_uint(v2) & 0x007FFFFFu) == 0u) ? v2 : 1.f;
    }
    f32x4 acc[RT][WN];
    if (!XP && g.H0q_in) {
        const rsrc_t rQ = gemm::make_rsrc(g.H0q_in + ((long long)((rs + 3) >> 2) + t) * (BN * 4), (unsigned)(((nrows + 3) >> 2) * BN * 16));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct) {   // (a quad past the tile's rows: out of range, zeros)
                const unsigned off = (unsigned)((rt * 4 + lg) * (BN * 16) + (wave * (16 * WN) + ct * 16 + li) * 16);
                acc[rt][ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rQ, off, 0, 0));
            }
    } else if (!XP && g.Cadd) {
        const rsrc_t rC = gemm::make_rsrc(g.Cadd + (long long)rs * g.ldcadd, nrows > 0 ? (unsigned)(((nrows - 1) * g.ldcadd + g.N) * 4) : 0u);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rt * 16 + lg * 4 + r, col = wave * (16 * WN) + ct * 16 + li;
                    const unsigned off = (row < nrows && col < g.N) ? (unsigned)(row * g.ldcadd + col) * 4u : kOOB;
                    acc[rt][ct][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rC, off, 0, 0));
                }
    } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // segment metadata (consumed by the epilogue)
    const int na0 = vb - va < kAtomCache ? vb - va : kAtomCache;
    const int* rvp = g.Mout ? g.revp : g.row_ptr;
    const int seg_rev = g.uniform ? 0 : rvp[(g.Mout && tid < nrows) ? rs + tid : 0];
    const int seg_rp = g.uniform ? 0 : g.row_ptr[va + (tid <= na0 ? tid : 0)] - rs;
    const float slope = g.slope_ptr ? *g.slope_ptr : g.slope;

    // ---- weight fragments: [column tile][chunk][hi|lo][lane][16 B], straight from L2 ----
    const int NTL = (g.N + 15) / 16;
    const rsrc_t rW = gemm::make_rsrc(g.W.p, g.A ? (unsigned)(NTL * g.W.nc * 2048) : 0u);
    const rsrc_t rW2 = gemm::make_rsrc(XP ? g.W2.p : g.W.p, XP ? (unsigned)(NTL * g.W2.nc * 2048) : 0u);
    unsigned offB[WN], offB2[WN];
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int tile = wave * WN + ct;
        offB[ct] = tile < NTL ? (unsigned)tile * (unsigned)(g.W.nc * 2048) + (unsigned)lane * 16u : kOOB;
        offB2[ct] = (tile < NTL && XP) ? (unsigned)tile * (unsigned)(g.W2.nc * 2048) + (unsigned)lane * 16u : kOOB;
    }
    auto load_bfrags = [&](const rsrc_t& rw, const unsigned (&off)[WN], int c, h8 (&bh)[WN], h8 (&bl)[WN]) {
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) {
            const unsigned o = off[ct] == kOOB ? kOOB : off[ct] + (unsigned)c * 2048u;
            bh[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rw, o, 0, 0));
            bl[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rw, o == kOOB ? kOOB : o + 1024u, 0, 0));
        }
    };
    h8 ah[2][RT], al[2][RT], bh[2][WN], bl[2][WN];
    // (the first contraction's first two weight chunks: in flight under the DMA)
    if (XP) { load_bfrags(rW2, offB2, 0, bh[0], bl[0]); load_bfrags(rW2, offB2, 1, bh[1], bl[1]); }
    else load_bfrags(rW, offB, 0, bh[0], bl[0]);   // (chunk 0 of the main contraction: the ring takes it from there)
    float isw[WN], isw2[WN], bv[WN];
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int col = wave * (16 * WN) + ct * 16 + li;
        const bool okc = col < g.N;
        isw[ct] = g.A ? g.W.inv_scale[okc ? col : 0] : 1.f;
        isw2[ct] = XP ? g.W2.inv_scale[okc ? col : 0] : 1.f;
        bv[ct] = ((okc && g.bias) ? g.bias[col] : 0.f) + ((okc && g.bias2) ? g.bias2[col] : 0.f);
    }
#if defined(DMPNN_STEP16_STAMPS2)
    stamp();  // c: everything requested
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();  // d: everything requested has LANDED (measurement build only: the x contraction below then overlaps nothing)
#endif
    if constexpr (XP) {
        // z = W2 x first, in the scale s_r2 s_W2 of its own rows and columns (no LDS involved: runs while the DMA of the main operand is still landing;
        // weight chunks 0 and 1 were requested with them, chunk c + 2 goes out behind chunk c) ...
        auto xstep = [&](int c, h8 (&h)[RT], h8 (&l)[RT], h8 (&yh)[WN], h8 (&yl)[WN]) {
#pragma unroll
            for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[rt], yh[ct], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[rt], yl[ct], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l[rt], yh[ct], acc[rt][ct], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < g.W2.nc) { load_xfrags(c + 2, h, l); load_bfrags(rW2, offB2, c + 2, yh, yl); }
        };
#if !defined(DMPNN_STEP16_DIAG_NOX)     // (timing diagnostic, WRONG numbers: no x contraction)
#pragma nounroll
        for (int c = 0; c < g.W2.nc; c += 2) {
            xstep(c, xh[0], xl[0], bh[0], bl[0]);
            if (c + 1 < g.W2.nc) xstep(c + 1, xh[1], xl[1], bh[1], bl[1]);
        }
#endif
    }
    stamp();  // 1 everything requested
    __syncthreads();  // the operand tiles have landed (the barrier's release waits for the DMA: vmcnt(0))
    stamp();  // 2 landed
    launder();
    // ---- barrier-free MFMA loops on the static operand tiles ----
    auto contract = [&](auto hin_c, const unsigned char* Ab, int ts, const rsrc_t& rw, const unsigned (&off)[WN], int n_chunks, bool prefetched) {
        constexpr bool H = decltype(hin_c)::value;
        auto read_afrags = [&](int c, h8 (&xh)[RT], h8 (&xl)[RT]) {
            const int cc = c < n_chunks ? c : n_chunks - 1;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (H) {
                    xh[rt] = *reinterpret_cast<const h8*>(Ab + (rt * 16 + li) * ts + cc * 64 + lg * 16);
                } else {
                    const unsigned char* p = Ab + (rt * 16 + li) * ts + cc * 128 + lg * 16;
                    xh[rt] = *reinterpret_cast<const h8*>(p);
                    xl[rt] = *reinterpret_cast<const h8*>(p + 64);
                }
            }
        };
        // the tile kernels' contraction (dmpnn_mega16_impl.hpp): ONE set of weight fragments as a ring over the column tiles — column
        // tile ct's three products together, its fragments of chunk c + 1 requested right behind them (+1.5 .. 3 % on BASELINE configs 2-4)
        auto ring = [&](int c, h8 (&xh)[RT], h8 (&xl)[RT], h8 (&nh)[RT], h8 (&nl)[RT]) {
            const bool more = c + 1 < n_chunks;
#pragma unroll
            for (int ct = 0; ct < WN; ++ct) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[rt], bh[0][ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[rt], bl[0][ct], acc[rt][ct], 0, 0, 0);
                if constexpr (!H) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[rt], bh[0][ct], acc[rt][ct], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#if defined(DMPNN_STEP16_DIAG_WSAME)   // (timing diagnostic, WRONG numbers: every weight fragment from the same 2 KiB — what does the weight stream cost?)
                const unsigned o = more ? (unsigned)lane * 16u : kOOB;
#else
                const unsigned o = (more && off[ct] != kOOB) ? off[ct] + (unsigned)(c + 1) * 2048u : kOOB;
#endif
                bh[0][ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rw, o, 0, 0));
                bl[0][ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rw, o == kOOB ? kOOB : o + 1024u, 0, 0));
                if (ct == (WN > 1 ? WN - 2 : 0)) read_afrags(c + 1, nh, nl);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (!prefetched) load_bfrags(rw, off, 0, bh[0], bl[0]);
        read_afrags(0, ah[0], al[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma nounroll
        for (int c = 0; c < n_chunks; c += 2) {
            ring(c, ah[0], al[0], ah[1], al[1]);
            if (c + 1 < n_chunks) ring(c + 1, ah[1], al[1], ah[0], al[0]);
        }
    };
    // the rows' own scales (tail of every operand row of the main operand; K1 without one: the scales of x); rows beyond the tile: 1
    float sr[RT][4], isr[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + lg * 4 + r;
            const float v = g.A ? *reinterpret_cast<const float*>(Ag + row * TS + (TS - 16)) : s2;
            // (a scale is a power of two by the format's contract; anything else — a corrupted tail — counts as 1)
            sr[rt][r] = (row < nrows && v > 0.f && v < 3.0e38f && (__float_as_uint(v) & 0x007FFFFFu) == 0u) ? v : 1.f;
            isr[rt][r] = rcp_pow2(sr[rt][r]);
        }
    if (XP && g.A) {
        // ... then into the scale of the main operand (exact: powers of two)
        const float is2 = rcp_pow2(s2);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float f = sr[rt][r] * is2;
#pragma unroll
                for (int ct = 0; ct < WN; ++ct) acc[rt][ct][r] *= f * (isw2[ct] * rcp_pow2(isw[ct]));
            }
    }
    if (XP && !g.A) {  // (K1: the scale of x IS the final one)
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) isw[ct] = isw2[ct];
    }
    if (!XP && (g.Cadd || g.H0q_in)) {
        // residual into the split domain of its row and column: acc = H0 s_r s_W (powers of two: exact)
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) {
            const float sw = rcp_pow2(isw[ct]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rt][ct][r] *= sr[rt][r] * sw;
        }
    }
    if (g.A) contract(std::integral_constant<bool, HIN>{}, Ag, TS, rW, offB, g.W.nc, !XP);

    stamp();  // 3 MFMA loop issued
#if defined(DMPNN_STEP16_DIAG_NOEPI)   // (timing diagnostic, WRONG numbers: no epilogue — the accumulators are summed into one store per lane)
    {
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < WN; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) sum += acc[rt][ct][0] + acc[rt][ct][1] + acc[rt][ct][2] + acc[rt][ct][3];
        if (g.Mout && nrows > 0) reinterpret_cast<float*>(g.Mout + (long long)rs * (g.half_out ? BN * 2 + 16 : TSO))[tid] = sum;
        return;
    }
#endif
    // ---- epilogue: split domain -> fp32 (+ bias); [pre-activation rows out]; tau; tile -> segment sums -> message / Mv ----
    launder();
#pragma unroll
    for (int ct = 0; ct < WN; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
#if defined(DMPNN_STEP16_IEEE_DIV)
            for (int r = 0; r < 4; ++r) acc[rt][ct][r] = acc[rt][ct][r] * (isw[ct] / sr[rt][r]) + bv[ct];
#else
            for (int r = 0; r < 4; ++r) acc[rt][ct][r] = acc[rt][ct][r] * (isw[ct] * isr[rt][r]) + bv[ct];
#endif
    stamp();  // 4 unscaled
    if (g.H0q_out) {  // (uniform; K1) the pre-activation H0 = W_i x + b_i leaves as row quads, straight from the fragments
        float* qb = g.H0q_out + ((long long)((rs + 3) >> 2) + t) * (BN * 4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            if (rt * 16 + lg * 4 < nrows) {
#pragma unroll
                for (int ct = 0; ct < WN; ++ct)
                    *reinterpret_cast<float4*>(qb + (rt * 4 + lg) * (BN * 4) + (wave * (16 * WN) + ct * 16 + li) * 4) =
                        make_float4(acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]);
            }
    }
    __syncthreads();  // every wave is done with the operand tile the epilogue tile overlays
    stamp();  // 5 all waves through the contraction
    // ReLU-class activations: a compare + select on the fragments.  tanh / ELU (and K1, whose pre-activation H0 leaves as
    // rows) go through ONE row-major pass over the tile instead: no transcendental code unrolled 60 times (code size is
    // latency for a workgroup that runs its code once).
    const bool zpre = g.Zpre != nullptr;
    const bool tpass = zpre || g.act == DMPNN_ACT_TANH || g.act == DMPNN_ACT_ELU;
    const float neg = tpass ? 1.f : (g.act == DMPNN_ACT_NONE ? 1.f : (g.act == DMPNN_ACT_RELU ? 0.f : slope));
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int cl = wave * (16 * WN) + ct * 16 + li;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = acc[rt][ct][r];
                T[(rt * 16 + lg * 4 + r) * LDC + cl] = (z > 0.f ? z : neg * z) + 0.f;
            }
    }
    if (tid < BM) meta[tid] = seg_rev;
    if (tpass) {  // (K1: the pre-activation H0 leaves as coalesced rows;) the tile becomes tau(z)
        __syncthreads();
        const int qn = g.N >> 2;
        const float nanv = __int_as_float(0x7fc00000);
        for (int it = tid; it < nrows * qn; it += NT) {
            const int r = qn == 1 ? it : (int)__umulhi((unsigned)it, g.qmagic), q = it - r * qn;
            float4* cell = reinterpret_cast<float4*>(T + r * LDC + 4 * q);
            float4 z = *cell;
            if (poison) z = make_float4(nanv, nanv, nanv, nanv);
            if (zpre) *reinterpret_cast<float4*>(g.Zpre + (long long)(rs + r) * g.ldz + 4 * q) = z;
            *cell = apply_act4(z, g.act, slope);
        }
    }
    stamp();  // 6 tile written (+ pre-activation rows)
    if (g.Yout) {  // (uniform) plain epilogue: the rows of tau(z), coalesced — the finalize over atoms
        __syncthreads();
        const int qn = g.N >> 2;
        const float nanv = __int_as_float(0x7fc00000);
        for (int it = tid; it < nrows * qn; it += NT) {
            const int r = qn == 1 ? it : (int)__umulhi((unsigned)it, g.qmagic), q = it - r * qn;
            float4 z = *reinterpret_cast<const float4*>(T + r * LDC + 4 * q);
            if (poison) z = make_float4(nanv, nanv, nanv, nanv);
            *reinterpret_cast<float4*>(g.Yout + (long long)(rs + r) * g.ldy + 4 * q) = z;
        }
        return;
    }
    if (g.Hout) {  // (uniform) training: the rows of H' = tau(z) leave coalesced before the segment pass turns the tile into the message
        __syncthreads();
        const int qn = g.N >> 2;
        const float nanv = __int_as_float(0x7fc00000);
        for (int it = tid; it < nrows * qn; it += NT) {
            const int r = qn == 1 ? it : (int)__umulhi((unsigned)it, g.qmagic), q = it - r * qn;
            float4 z = *reinterpret_cast<const float4*>(T + r * LDC + 4 * q);
            if (poison) z = make_float4(nanv, nanv, nanv, nanv);
            *reinterpret_cast<float4*>(g.Hout + (long long)(rs + r) * g.ldho + 4 * q) = z;
        }
    }
    if (g.bits) {  // (uniform) lean training: the sign of tau(z) — one byte per (row, 8 columns) — instead of the fp32 rows
        __syncthreads();
        constexpr int G8B = BN / 8;
        for (int it = tid; it < nrows * G8B; it += NT) {
            const int r = it / G8B, g8 = it - r * G8B;
            const float4 m0 = *reinterpret_cast<const float4*>(T + r * LDC + 8 * g8);
            const float4 m1 = *reinterpret_cast<const float4*>(T + r * LDC + 8 * g8 + 4);
            const unsigned b = (m0.x > 0.f ? 1u : 0u) | (m0.y > 0.f ? 2u : 0u) | (m0.z > 0.f ? 4u : 0u) | (m0.w > 0.f ? 8u : 0u) |
                               (m1.x > 0.f ? 16u : 0u) | (m1.y > 0.f ? 32u : 0u) | (m1.z > 0.f ? 64u : 0u) | (m1.w > 0.f ? 128u : 0u);
            g.bits[(long long)(rs + r) * g.bstride + g8] = (unsigned char)b;
        }
    }
    int scale_phase = 0;
    auto tile_scale = [&](float local_max) -> float {
        stamp();  // 7 pass 1 done
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
        if (tid == 0) maxbits[(scale_phase + 2) & 3] = 0u;  // (re-armed two calls ahead: last read before the previous call's barrier)
        ++scale_phase;
        return scale_for(mxv);
    };
    SegOut o;
    o.row_ptr = g.row_ptr; o.revp = g.revp; o.Mout = g.Mout; o.ts = g.half_out ? BN * 2 + 16 : TSO; o.Sout = g.Sout; o.lds = g.lds;
    o.N = g.N; o.half = g.half_out; o.SoutS = g.SoutS; o.tss = TSO; o.M32 = g.M32; o.ldm32 = g.ldm32;
    seg_epilogue<LDC, BN / 4, NT>(o, T, meta, rs, nrows, va, vb, seg_rp, poison, g.qmagic, tile_scale);
    stamp();  // 8 (7 without a message) end
}

// ---- the K1 operand [A1[g1[r]] || A2[g2[r]]] (fp32, gathered) as split rows: one workgroup per row tile of the plan ----
// (used where K1 runs on k_step16 itself: d_h > 320; narrower blocks take k_rows16<.., SEG>, which splits on the fly)
struct SplitRowsK {
    const int* tile_row; int n_tiles; int n_rows;   // (tile_row null: uniform 48-row tiles over n_rows rows)
    const float* A1; int lda1; const int* g1; int K1; unsigned a1_bytes;   // (g1 / g2 null: rows in place)
    const float* A2; int lda2; const int* g2; int K2; unsigned a2_bytes;
    unsigned char* out; int ts;
};
__device__ __forceinline__ void split_rows_body(const SplitRowsK& g) {
    __shared__ unsigned maxbits;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (no tile table: uniform 48-row tiles over n_rows rows; no gather arrays: the rows in place)
    const int rs = g.tile_row ? g.tile_row[blockIdx.x] : BM * (int)blockIdx.x;
    const int nrows = g.tile_row ? g.tile_row[blockIdx.x + 1] - rs : (rs + BM < g.n_rows ? BM : g.n_rows - rs);
    if (nrows <= 0 || nrows > BM) return;
    if (tid == 0) maxbits = 0u;
    __syncthreads();
    const int K = g.K1 + g.K2, nc = (g.ts - 16) / 128;
    const rsrc_t r1 = gemm::make_rsrc(g.A1, g.a1_bytes), r2 = gemm::make_rsrc(g.A2 ? g.A2 : g.A1, g.A2 ? g.a2_bytes : 0u);
    constexpr int J = BM / 4;
    auto fetch = [&](unsigned o1, unsigned o2, int k) -> gemm::u32x2 {
        const unsigned k1o = k < g.K1 ? (unsigned)k * 4u : kOOB;
        const unsigned k2o = (k >= g.K1 && k < K) ? (unsigned)(k - g.K1) * 4u : kOOB;
        return __builtin_amdgcn_raw_buffer_load_b64(r1, gemm::join_off(o1, k1o), 0, 0) |
               __builtin_amdgcn_raw_buffer_load_b64(r2, gemm::join_off(o2, k2o), 0, 0);
    };
    auto put = [&](int r, int k, float x, float y, float s) {
        unsigned hi, lo;
        mega16::split2(x, y, s, hi, lo);
        unsigned char* p = g.out + (long long)(rs + r) * g.ts + (k >> 5) * 128 + (k & 31) * 2;
        *reinterpret_cast<unsigned*>(p) = hi;
        *reinterpret_cast<unsigned*>(p + 64) = lo;
    };
    float mx = 0.f;
    if (nc * 32 <= 256) {
        // one pass: the tile's gathered operand (<= 48 rows x 256 columns) is held in registers between the maximum and the split
        gemm::u32x2 v[J][2];
        unsigned o1[J], o2[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int r = wave + 4 * j;
            const bool ok = r < nrows;
            o1[j] = ok ? (unsigned)(g.g1 ? g.g1[rs + r] : rs + r) * (unsigned)g.lda1 * 4u : kOOB;
            o2[j] = (ok && g.A2) ? (unsigned)(g.g2 ? g.g2[rs + r] : rs + r) * (unsigned)g.lda2 * 4u : kOOB;
        }
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = lane * 2 + 128 * i;
                v[j][i] = (k < nc * 32) ? fetch(o1[j], o2[j], k) : gemm::u32x2{0u, 0u};
                mx = fmaxf(mx, fmaxf(fabsf(__uint_as_float(v[j][i].x)), fabsf(__uint_as_float(v[j][i].y))));
            }
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (lane == 0) atomicMax(&maxbits, __float_as_uint(mx));
        __syncthreads();
        const float s = scale_for(__uint_as_float(maxbits));
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int r = wave + 4 * j;
            if (r >= nrows) continue;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = lane * 2 + 128 * i;
                if (k < nc * 32) put(r, k, __uint_as_float(v[j][i].x), __uint_as_float(v[j][i].y), s);
            }
            if (lane == 0) *reinterpret_cast<float4*>(g.out + (long long)(rs + r) * g.ts + (g.ts - 16)) = make_float4(s, 0.f, 0.f, 0.f);
        }
        return;
    }
    // wider operands: two passes over the (L2-resident) rows
    for (int pass = 0; pass < 2; ++pass) {
        float s = 1.f;
        if (pass == 1) {
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            if (lane == 0) atomicMax(&maxbits, __float_as_uint(mx));
            __syncthreads();
            s = scale_for(__uint_as_float(maxbits));
        }
        for (int j = 0; j < J; ++j) {
            const int r = wave + 4 * j;
            if (r >= nrows) continue;
            const unsigned o1 = (unsigned)(g.g1 ? g.g1[rs + r] : rs + r) * (unsigned)g.lda1 * 4u;
            const unsigned o2 = g.A2 ? (unsigned)(g.g2 ? g.g2[rs + r] : rs + r) * (unsigned)g.lda2 * 4u : kOOB;
            for (int k = lane * 2; k < nc * 32; k += 128) {
                const gemm::u32x2 v = fetch(o1, o2, k);
                const float x = __uint_as_float(v.x), y = __uint_as_float(v.y);
                if (pass == 0) mx = fmaxf(mx, fmaxf(fabsf(x), fabsf(y)));
                else put(r, k, x, y, s);
            }
            if (pass == 1 && lane == 0) *reinterpret_cast<float4*>(g.out + (long long)(rs + r) * g.ts + (g.ts - 16)) = make_float4(s, 0.f, 0.f, 0.f);
        }
    }
}

static __global__ __launch_bounds__(256) void k_split_rows(SplitRowsK g) { split_rows_body(g); }
// ... with the pre-split of the forward's weights riding in the same launch (workgroups beyond the row tiles: 4 waves each, one wave
// per matrix row): this launch does not read the weights, the next one does — the split costs no launch of its own
static __global__ __launch_bounds__(256) void k_split_rows_w(SplitRowsK g, mega16::SplitArgs sp, int n_row_blocks) {
    if ((int)blockIdx.x < n_row_blocks) split_rows_body(g);
    else mega16::split_weights_wave(sp, ((int)blockIdx.x - n_row_blocks) * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
}

template <int WN, int NW, bool HIN, bool XP>
int launch_step16(const Step16K& g, int n_tiles, hipStream_t s);

#define DMPNN_DEFINE_STEP16_H(WN, NW, HIN, XP)                                                             \
    template <>                                                                                            \
    int launch_step16<WN, NW, HIN, XP>(const Step16K& g, int n_tiles, hipStream_t s) {                     \
        const size_t lds = (size_t)g.tile_bytes + meta_bytes<WN, NW>();                                    \
        static size_t attr_set = 0;                                                                        \
        if (attr_set < lds) {                                                                              \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step16<WN, NW, HIN, XP>),  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
            if (e != hipSuccess) {                                                                         \
                set_error("hipFuncSetAttribute(k_step16<%d,%d>, %zu B LDS): %s", WN, NW, lds, hipGetErrorString(e)); \
                return DMPNN_EHIP;                                                                         \
            }                                                                                              \
            attr_set = lds;                                                                                \
        }                                                                                                  \
        hipLaunchKernelGGL((k_step16<WN, NW, HIN, XP>), dim3((unsigned)n_tiles), dim3(64 * NW), lds, s, g); \
        DMPNN_CHECK_LAUNCH("k_step16");                                                                    \
        return DMPNN_OK;                                                                                   \
    }
#define DMPNN_DEFINE_STEP16(WN, NW)                                                                        \
    DMPNN_DEFINE_STEP16_H(WN, NW, false, false) DMPNN_DEFINE_STEP16_H(WN, NW, true, false)                 \
    DMPNN_DEFINE_STEP16_H(WN, NW, false, true) DMPNN_DEFINE_STEP16_H(WN, NW, true, true)

}  // namespace step16
}  // namespace dmpnn
