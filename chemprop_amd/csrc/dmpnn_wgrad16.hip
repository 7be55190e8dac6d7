// Weight gradients on the f16 matrix pipe (round 2):  gW[n][k] = sum_m gZ[m][n] * Acat[m][k],  Acat = [A1[g1(m)] || A2[g2(m)] || 1].
//
// The reduction index m is the ROW index of both operands; v_mfma_f32_16x16x32_f16 wants 8 consecutive reduction elements per
// lane, i.e. both operands TRANSPOSED, and the exact 3-term split (x s = hi + lo, s a power of two) needs a scale that is
// uniform along the reduction.  Doing transpose + split inside the product kernel was built first and is a wash: every
// operand element is read — and then split — by each of the N/64 (or K/64) workgroups that need it, and the VALU cost of the
// splits replaces the matrix-pipe time saved (30.5 us per launch against 30.6 us, branch exp/mol-tiles).  So the operands are
// split ONCE:
//
//   k_wsplit16   one launch for all operands of a backward pass.  Block (column tile ct of 64 outputs, chunk c of 32 rows)
//                of an operand becomes 64 rows x [hi 32 halfs | lo 32 halfs] = 8 KB, contiguous, in exactly the byte order
//                the product kernel's LDS reads want (the 16-byte pieces of a row are XOR-swizzled with (row >> 1) & 7: a
//                plain 128-byte row stride would put the 16 rows of a ds_read_b128 on the same banks, and an LDS-DMA cannot
//                pad), + one power-of-two scale per block (block maximum).  A workgroup does 4 chunks of one column tile:
//                four coalesced row loads per thread and chunk pair, register transpose (4 consecutive reduction rows of
//                one output = one 8-byte store), gather and concatenation of [A1[g1] || A2[g2] || 1] on the fly.
//   k_wgrad16    (output tile 64 x 64, row split) per 2-chunk stage: both operand blocks by LDS-DMA (16 KB + 16 KB, no
//                registers, no VALU), barrier, 24 MFMAs per wave into fresh accumulators, fp32 sums += product / (s_z s_a)
//                (exact inverse: scales never mix).  Wave w owns output rows 16 w .. of the tile.  Slab per split, reduced
//                by k_wgrad_reduce as before (deterministic: no atomics).
#include <type_traits>
#include <stdlib.h>
#include <string.h>

#include "dmpnn_common.hpp"
#include "dmpnn_gemm_impl.hpp"

namespace dmpnn {
namespace wg16 {

using gemm::f32x4;
using gemm::rsrc_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int kBlk = 8192;  // bytes of one (64 outputs x 32 reduction rows) block

__device__ __forceinline__ float scale_for(float maxabs) {  // exact power of two that puts maxabs at [2^13, 2^14); 1 for 0 / inf / nan
    if (!(maxabs > 0.f) || !(maxabs < 3.0e38f)) return 1.f;
    int e;
    frexpf(maxabs, &e);
    return ldexpf(1.f, 14 - e);
}

// k_wsplit16 ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wsplit16(WSplitArgs a) {
    __shared__ unsigned mx[4];
    int j = 0;
    while (j + 1 < a.n_jobs && (int)blockIdx.x >= a.job[j + 1].wg0) ++j;
    const WSplitJob& J = a.job[j];
    const int local = (int)blockIdx.x - J.wg0;
    const int n_cg = (J.n_chunks + 3) >> 2;
    const int ct = local / n_cg, cg = local - ct * n_cg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mb = tid >> 4, cb = tid & 15;  // rows 4 mb .. + 3 of a 64-row half, outputs 4 cb .. + 3 of the column tile
    const int K = J.K1 + J.K2;
    if (tid < 4) mx[tid] = 0u;
    __syncthreads();
    // the two column pairs of this thread: where they come from is fixed for the whole workgroup
    const float* src[2];
    int64_t ld[2];
    const int* gth[2];
    const long long* gth64[2];
    int kind[2];  // 0: read, 1: (1, 0) — the bias column of ones, 2: zeros
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = 64 * ct + 4 * cb + 2 * h;
        gth64[h] = nullptr;
        if (c < J.K1) { src[h] = J.A1 + c; ld[h] = J.lda1; gth[h] = J.g1; gth64[h] = J.g1 ? nullptr : J.g1_64; kind[h] = 0; }
        else if (c < K) { src[h] = J.A2 + (c - J.K1); ld[h] = J.lda2; gth[h] = J.g2; kind[h] = 0; }
        else { src[h] = J.A1; ld[h] = 0; gth[h] = nullptr; kind[h] = (J.ones && c == K) ? 1 : 2; }
    }
    float2 v[2][4][2];  // [half of 64 rows][row][column pair]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t m = (int64_t)128 * cg + 64 * q + 4 * mb + r;
            const int64_t mc = m < J.M ? m : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int64_t row = gth[h] ? (int64_t)gth[h][mc] : mc;
                if (gth64[h]) { row = (int64_t)gth64[h][mc]; row = row < 0 ? 0 : (row >= J.g1_rows ? J.g1_rows - 1 : row); }
                v[q][r][h] = *reinterpret_cast<const float2*>(src[h] + row * ld[h]);
            }
        }
    float mloc[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool mok = (int64_t)128 * cg + 64 * q + 4 * mb + r < J.M;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float2 x = v[q][r][h];
                if (!mok || kind[h] == 2) x = make_float2(0.f, 0.f);
                else if (kind[h] == 1) x = make_float2(1.f, 0.f);
                v[q][r][h] = x;
                mloc[q] = fmaxf(mloc[q], fmaxf(fabsf(x.x), fabsf(x.y)));
            }
        }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float m = mloc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) atomicMax(&mx[2 * q + (wave >> 1)], __float_as_uint(m));  // (waves 0, 1 hold rows 0..31 of a half, waves 2, 3 rows 32..63)
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int chunk = 4 * cg + 2 * q + (wave >> 1);
        if (chunk >= J.n_chunks) continue;
        const float s = scale_for(__uint_as_float(mx[2 * q + (wave >> 1)]));
        unsigned char* blk = J.out + ((int64_t)ct * J.n_chunks + chunk) * kBlk;
        const int p = (mb & 7) >> 1, sub = (mb & 1) * 8;  // 16-byte piece of the 4 reduction rows, and the half of it
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {  // the register transpose: 4 consecutive reduction rows of one output
            const int n = 4 * cb + cc, key = (n >> 1) & 7;
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = ((cc & 1) ? v[q][r][cc >> 1].y : v[q][r][cc >> 1].x) * s;
            const h4 hi = h4{(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
            const h4 lo = h4{(_Float16)(x[0] - (float)hi[0]), (_Float16)(x[1] - (float)hi[1]), (_Float16)(x[2] - (float)hi[2]), (_Float16)(x[3] - (float)hi[3])};
            *reinterpret_cast<h4*>(blk + n * 128 + ((p ^ key) << 4) + sub) = hi;
            *reinterpret_cast<h4*>(blk + n * 128 + (((p + 4) ^ key) << 4) + sub) = lo;
        }
        if ((tid & 127) == 0) J.scales[(int64_t)ct * J.n_chunks + chunk] = s;
    }
}

// k_wgrad16 -------------------------------------------------------------------------------------------------------------
constexpr int kStage = 2;  // chunks per stage

__global__ __launch_bounds__(256) void k_wgrad16(WProdJobs jobs) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kStage * kBlk];  // [Z blocks of the stage | A blocks of the stage]
    int j = 0;
    while (j + 1 < jobs.n_jobs && (int)blockIdx.x >= jobs.wg0[j + 1]) ++j;
    const WProdArgs& a = jobs.job[j];
    const int local = (int)blockIdx.x - jobs.wg0[j];  // (wg0 is a multiple of 8: local & 7 is still the XCD)
    const int tiles = a.n_nt * a.n_kt;
    const int per = (jobs.wg0[j + 1] - jobs.wg0[j]) >> 3;  // XCD-aware order of the (split, tile) ranks: the workgroups of one row split share an L2
    const int rank = (local & 7) * per + (local >> 3);
    if (rank >= tiles * a.splits) return;
    const int split = rank / tiles, tile = rank - split * tiles;
    const int nt = tile / a.n_kt, kt0 = tile - nt * a.n_kt;
    const int c_lo = split * a.chunks_per_split;
    const int c_hi = c_lo + a.chunks_per_split < a.n_chunks ? c_lo + a.chunks_per_split : a.n_chunks;
    if (c_lo >= c_hi) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned char* Zb = a.Z + (int64_t)nt * a.n_chunks * kBlk;
    const unsigned char* Ab = a.A + (int64_t)kt0 * a.n_chunks * kBlk;
    const float* sz = a.sZ + (int64_t)nt * a.n_chunks;
    const float* sa = a.sA + (int64_t)kt0 * a.n_chunks;
    const rsrc_t rZ = gemm::make_rsrc(Zb, (unsigned)(a.n_chunks * kBlk)), rA = gemm::make_rsrc(Ab, (unsigned)(a.n_chunks * kBlk));
    f32x4 acc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment addresses inside a block: row (16 w + li | 16 kt + li), pieces lg (hi) and lg + 4 (lo), swizzled with (row >> 1) & 7
    const int zrow = 16 * wave + li, zkey = (zrow >> 1) & 7;
    const int zoff_h = zrow * 128 + ((lg ^ zkey) << 4), zoff_l = zrow * 128 + (((lg + 4) ^ zkey) << 4);
    int aoff_h[4], aoff_l[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const int arow = 16 * kt + li, akey = (arow >> 1) & 7;
        aoff_h[kt] = arow * 128 + ((lg ^ akey) << 4);
        aoff_l[kt] = arow * 128 + (((lg + 4) ^ akey) << 4);
    }
    for (int c = c_lo; c < c_hi; c += kStage) {
        const int nst = c_hi - c < kStage ? c_hi - c : kStage;
        // both operands' blocks of the stage: contiguous in memory, 1 KiB per wave instruction
        for (int i = wave; i < nst * 8; i += 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rZ, (__attribute__((address_space(3))) void*)(lds + i * 1024), 16,
                                                     (unsigned)(c * kBlk + i * 1024 + lane * 16), 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(lds + kStage * kBlk + i * 1024), 16,
                                                     (unsigned)(c * kBlk + i * 1024 + lane * 16), 0, 0, 0);
        }
        float inv[kStage];
#pragma unroll
        for (int s = 0; s < kStage; ++s) inv[s] = s < nst ? 1.f / (sz[c + s] * sa[c + s]) : 0.f;
        __syncthreads();  // (the barrier's release waits for the DMA)
#pragma unroll
        for (int s = 0; s < kStage; ++s) {
            if (s < nst) {
                const unsigned char* zb = lds + s * kBlk;
                const unsigned char* ab = lds + kStage * kBlk + s * kBlk;
                const h8 ah = *reinterpret_cast<const h8*>(zb + zoff_h), al = *reinterpret_cast<const h8*>(zb + zoff_l);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const h8 bh = *reinterpret_cast<const h8*>(ab + aoff_h[kt]), bl = *reinterpret_cast<const h8*>(ab + aoff_l[kt]);
                    f32x4 p = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    p = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, p, 0, 0, 0);
                    p = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, p, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[kt][r] = fmaf(p[r], inv[s], acc[kt][r]);
                }
            }
        }
        __syncthreads();  // (every wave is done with the stage before the next DMA overwrites it)
    }
    // D fragment: lane (li, lg) holds rows 16 wave + 4 lg + r, column 16 kt + li of the tile
    float* slab = a.slab + (int64_t)split * a.slab_stride;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 64 * nt + 16 * wave + 4 * lg + r, k = 64 * kt0 + 16 * kt + li;
            if (n < a.N && k < a.Kt) slab[(int64_t)n * a.ldk + k] = acc[kt][r];
        }
}

// k_rows2sr ---------------------------------------------------------------------------------------------------------------
// fp32 rows [A1[g1(m)] || A2[g2(m)]] -> split rows (whole 32-column chunks of [hi | lo] + the tail with the row's own power-of-two
// scale): the operands of k_wgrad16r that no kernel already holds split — the K1 operand [V[src] || E], the finalize's [V || Mv], a
// rider's plain rows.  One wave per row, several operands per launch.
struct SRJobs { SRJob job[4]; int wg0[5]; int n_jobs; };
__global__ __launch_bounds__(256) void k_rows2sr(SRJobs a) {
    int j = 0;
    while (j + 1 < a.n_jobs && (int)blockIdx.x >= a.wg0[j + 1]) ++j;
    const SRJob& J = a.job[j];
    const int lane = threadIdx.x & 63;
    const int64_t m = ((int64_t)blockIdx.x - a.wg0[j]) * 4 + (threadIdx.x >> 6);
    if (m >= J.M) return;
    int64_t r1 = J.g1 ? (int64_t)J.g1[m] : (J.g1_64 ? (int64_t)J.g1_64[m] : m);
    if (J.g1 || J.g1_64) r1 = r1 < 0 ? 0 : (r1 >= J.g1_rows ? J.g1_rows - 1 : r1);   // (an index out of range: the plan's verdict poisons the result)
    const int64_t r2 = J.g2 ? (int64_t)J.g2[m] : m;
    const float* x1 = J.A1 + r1 * J.lda1;
    const float* x2 = J.K2 ? J.A2 + r2 * J.lda2 : x1;
    const int K = J.K1 + J.K2;
    float v[8];
    float mx = 0.f;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int c = lane + 64 * jj;
        v[jj] = c < J.K1 ? x1[c] : (c < K ? x2[c - J.K1] : 0.f);
        mx = fmaxf(mx, fabsf(v[jj]));
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const float sc = scale_for(mx);
    unsigned char* o = J.out + m * J.ts;
    const int cols = ((K + 31) >> 5) << 5;   // whole chunks: the padding columns are written (zeros)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int c = lane + 64 * jj;
        if (c < cols) {
            const float y = v[jj] * sc;
            const _Float16 hi = (_Float16)y;
            *reinterpret_cast<_Float16*>(o + (c >> 5) * 128 + (c & 31) * 2) = hi;
            *reinterpret_cast<_Float16*>(o + (c >> 5) * 128 + 64 + (c & 31) * 2) = (_Float16)(y - (float)hi);
        }
    }
    if (lane == 0) *reinterpret_cast<float4*>(o + (J.ts - 16)) = make_float4(sc, mx > 0.f ? 0.f : 1.f, 0.f, 0.f);   // (scale, zero-row flag)
}

// k_wgrad16r ------------------------------------------------------------------------------------------------------------
// The product over operands in SPLIT-ROW form (round 4): rows of [hi 32 halfs | lo 32 halfs] chunks + a 16-byte tail with the row's
// power-of-two scale — the form the step kernels keep their message rows in and contract them from (dmpnn_step16_impl.hpp), row-major,
// nothing transposed or re-blocked by anyone.  v_mfma_f32_16x16x32_f16 wants 8 consecutive REDUCTION elements (rows) per lane; gfx950's
// LDS transpose read (ds_read_b64_tr_b16: lane i of a 16-lane group receives column i of the 4 x 16 block whose sixteen 8-byte pieces
// the group's lanes address, lane 4 row + column quad) delivers exactly that from a row-major image, so both operands go from memory
// to LDS by DMA as they are — 32 rows per stage — and leave it as fragments.
//   LDS image of a stage: chunk c of row r (128 bytes = 8 pieces of 16) at c 4096 + r 128, piece p at slot p ^ swz(r),
//   swz(r) = 2 ((r >> 1) & 1) + 4 ((r >> 3) & 1).  The DMA builds it: lane l of unit u (rows 8 u .. 8 u + 7) fetches piece
//   (l & 7) ^ swz(r) of row r = 8 u + (l >> 3) — whole 128-byte lines — and lands at the unit's byte 16 l.
// Output tile: ALL n (<= 320 rows: every 16-row tile of Z's columns) x 128 k per workgroup — wave w owns the tiles w, w + 4, .. of n (<= 5)
// and all eight 16-column tiles of k: 120 MFMAs per wave and 32-row stage from 52 transpose reads; Z is streamed once per 128
// columns of k (the 64 x 64 kernel above reads 10 KB of LDS per 12 MFMAs and is bound by the LDS, not by the matrix pipe).
// Scales are per ROW: instead of fresh accumulators and one multiply-add per stage and element, the Z fragments (8 reduction rows per
// lane) are scaled DOWN by the exact powers of two rho = F / (s_Z s_A) of their rows (an h8 per lane group from LDS, four
// v_pk_mul_f16), F = the smallest s_Z s_A of the workgroup's whole row range: the products of all stages then share the factor F and
// accumulate in the matrix pipe (a row whose rho leaves the f16 range holds values negligible beside the range's largest).
struct WProdR {
    const unsigned char* Z; const unsigned char* A;   // split rows [M][tsz], [M][tsa];  A null: the COLUMN SUMS of Z (a bias gradient) — one
                                                      // k tile whose first column is all ones, rho = F / s_Z
    float* slab;                                      // this job's slabs [splits][N][ldk]
    float* slab_b;                                    // (or null) the column sums of Z ride in the job's k group 0: slabs [splits][N][4], column 0
    long long M; int N, K;                            // (column sums: K = 1)
    int tsz, ncz, tsa, nca;                           // row bytes and live chunks of Z (ceil(N / 32)) and A (ceil(K / 32))
    int n_kg, splits, rows_per_split, per8;           // per8: workgroups of the job / 8 (launch order)
    int even_kt;                                      // 1: the k column groups are the job's 16-column tiles dealt evenly (each within four chunks of A)
    int ldk; long long slab_stride;
    int wg0;                                          // the job's first workgroup
};
struct WProdRJobs { WProdR job[kWProdRMaxJobs]; int n_jobs; long long* dbg; };   // dbg: optional [64] cycle stamps of workgroup 0 (dmpnn_debug_timestamps)
}  // namespace wg16

namespace wg16 {
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 f16x4_t;
using gemm::u32x4;
using gemm::u32x2;
constexpr int kRChunk = 4096;   // bytes of one chunk of a stage: 32 rows x 128

__device__ __forceinline__ h8 tr_pair(const unsigned char* base, int off) {   // rows 8 g .. 8 g + 7 of one 16-column tile: two transpose reads
    const f16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) f16x4_t*)(base + off));
    const f16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) f16x4_t*)(base + off + 512));
    const h4 x = __builtin_bit_cast(h4, a), y = __builtin_bit_cast(h4, b);
    return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Round 6: ONE 512-thread workgroup per CU, two stage images in LDS, filled by PLAIN global loads -> registers -> ds_write_b128.
// The round-4 form (256 threads, two workgroups per CU, one image filled by LDS-DMA) spent 4.7 k of a stage's 9.8 k cycles ISSUING
// the 56 DMA pieces — a CU takes ~12 bytes per cycle through that path however the pieces are issued, and a wave that waits for it
// issues no MFMA (profiles/r04_wgrad16r_stage_stamps.txt) — and hipcc orders a wave's LDS reads behind all of its outstanding DMA,
// so a second image bought nothing.  A stage's 56 KB as 7 x 16 bytes per thread through the L1 path are requested at the top of the
// stage, land in registers under the stage's MFMAs, and go to the OTHER image behind them: one barrier per stage, no DMA.  (A second
// register set — stage s + 2 in flight under stage s + 1 — does not fit beside the 96 accumulators: 44 of them spilled and the launch
// took 170 us instead of 50, profiles/r06_wgrad16r_notes.txt; and the launch is bound by the bytes it moves — ~10 B/clk per CU when
// every CU streams from HBM — not by one round trip per stage.)
// Wave w owns the 16-row tiles w, w + 8, w + 16 of n (<= 3) and all (<= 8) 16-column tiles of the workgroup's k column group; the k
// column groups of a job are whole 32-column chunks of A, as even as they come (300 columns: 3 + 3 + 4 chunks, not 4 + 4 + 2).
// The COLUMN SUMS of Z (the bias gradient) ride in the k group 0 workgroups of a product (slab_b): the Z fragments are there — one more
// scaling (rho_c = F_c / s_Z) and two MFMAs per tile against a fragment of ones, instead of a job that reads all of Z again.
template <int I, int N, class Fn>
__device__ __forceinline__ void wg_static_for(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wg_static_for<I + 1, N>(f);
    }
}
constexpr int kRTW8 = 3;   // most 16-row tiles of n per wave
constexpr int kRPiecesZ = 5, kRPiecesA = 2;   // pieces per thread and stage: chunks lc0, lc0 + 2, .. of Z (<= 10) and of the k column group's A (<= 4)
struct RSet { u32x4 Z[kRPiecesZ]; u32x4 A[kRPiecesA]; u32x2 tz, ta; };
// NTW: 16-row tiles of n per wave (1 .. 3; the launch's widest job: ceil(ceil(N / 16) / 8)) — a template parameter and the SAME for every
// wave, so that no product sits under a branch: tested per product (wave + 8 r < n_nt), every MFMA was wrapped in s_and_saveexec /
// s_cbranch_execz and issued every ~30 cycles; as per-wave variants of the stage inside one kernel hipcc shuffled the accumulators between
// them and spilled.  A wave's tiles beyond the job's n are PHANTOMS: they multiply whatever the image holds there (always inside the image)
// into accumulators nobody stores — 24 tiles' products for the 19 of d_h = 300.
template <int NTW>
__global__ __launch_bounds__(512, 2) void k_wgrad16r(WProdRJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // rho16[2][32] | rhoc16[2][32] | red | image 0 | image 1  (image: ncz chunks of Z | <= 4 chunks of A)
    int j = 0;
    while (j + 1 < jobs.n_jobs && (int)blockIdx.x >= jobs.job[j + 1].wg0) ++j;
    const WProdR& P = jobs.job[j];
    const WProdR& a = P;
    const int local = (int)blockIdx.x - P.wg0;
    const int rank = (local & 7) * P.per8 + (local >> 3);   // XCD-aware: the k columns of one row split share an L2 (they stream the same Z rows)
    if (rank >= P.n_kg * P.splits) return;
    const int split = rank / P.n_kg, kg = rank - split * P.n_kg;
    const bool colsum = P.A == nullptr;
    const bool ride_b = !colsum && P.slab_b != nullptr && kg == 0;   // (uniform)
    const long long m_lo = (long long)split * P.rows_per_split;
    const long long m_hi = m_lo + P.rows_per_split < P.M ? m_lo + P.rows_per_split : P.M;
    const int n_rows = (int)(m_hi - m_lo);   // (> 0: the host sizes `splits` so that every split holds rows)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: what depends on it alone is a scalar branch)
    const int li = lane & 15, lg = lane >> 4;
    int n_stamp = 0;
    auto stamp = [&]() {
        if (jobs.dbg && blockIdx.x == 0 && threadIdx.x == 0 && n_stamp < 64) jobs.dbg[n_stamp] = (long long)__builtin_readcyclecounter();
        ++n_stamp;
    };
    stamp();  // 0 entry
    _Float16* rho16 = reinterpret_cast<_Float16*>(lds);                       // [2][32]
    _Float16* rhoc16 = reinterpret_cast<_Float16*>(lds + 128);                // [2][32]
    unsigned* red = reinterpret_cast<unsigned*>(lds + 256);
    // (the job's fields once, as scalars: `jobs.job[j]` with a computed j is kernarg MEMORY to hipcc — it re-read tsz / tsa per piece)
    const int tsz = __builtin_amdgcn_readfirstlane(P.tsz), tsa = __builtin_amdgcn_readfirstlane(P.tsa), ncz = __builtin_amdgcn_readfirstlane(P.ncz);
    const int img_bytes = (ncz + 4) * kRChunk;
    unsigned char* img0 = lds + 512;
    // the workgroup's rows of both operands as buffers: a row past the range is out of range of its descriptor and reads as ZEROS —
    // no address selects, no zeroing of dead pieces (the host holds rows_per_split x row bytes below 2^31)
    const rsrc_t rZ = gemm::make_rsrc(a.Z + m_lo * tsz, (unsigned)(n_rows * tsz));
    const rsrc_t rA = gemm::make_rsrc(colsum ? a.Z : a.A + m_lo * tsa, colsum ? 0u : (unsigned)(n_rows * tsa));
    // this workgroup's 16-column tiles of k: [kt0, kt0 + kt_live), the job's ceil(K / 16) tiles dealt to its n_kg column groups as evenly as
    // they come (300 columns: 7 + 6 + 6, not 8 + 8 + 3); its image holds the 32-column chunks [ca0, ca0 + ca_live) of A those tiles
    // lie in (<= 4: a group of 7 or 8 tiles), koff = 1 when the group starts in the second half of chunk ca0
    // (even_kt 0 — a group would then span five chunks, e.g. 23 tiles as 7 + 8 + 8 —: whole chunks per group, the tiles of the last one short)
    const int nkt = (P.K + 15) >> 4;
    const int kt0 = colsum ? 0 : (P.even_kt ? (kg * nkt) / P.n_kg : 2 * ((kg * P.nca) / P.n_kg));
    const int kt1 = P.even_kt ? ((kg + 1) * nkt) / P.n_kg : (2 * (((kg + 1) * P.nca) / P.n_kg) < nkt ? 2 * (((kg + 1) * P.nca) / P.n_kg) : nkt);
    const int kt_live = colsum ? 0 : kt1 - kt0;
    const int ca0 = kt0 >> 1, koff = kt0 & 1;
    const int ca_live = colsum ? 0 : ((kt0 + kt_live + 1) >> 1) - ca0;
    // ---- the loader's thread: row lr of the stage, LDS slot lq of its 128-byte line <- piece lq ^ swz(lr), chunks lc0, lc0 + 2, ..
    const int lr = (tid & 255) >> 3, lq = tid & 7, lc0 = tid >> 8;
    const int lpiece = lq ^ (2 * ((lr >> 1) & 1) + 4 * ((lr >> 3) & 1));
    // two 32-bit offsets per thread (its row and first piece in Z and in A; piece jj: + 256 jj, the instruction's immediate) and one for the
    // rows' tails (thread t < 32 owns row t of the stage)
    const unsigned offz = (unsigned)(lr * tsz + lpiece * 16 + lc0 * 128);
    const unsigned offa = (unsigned)(lr * tsa + lpiece * 16 + (ca0 + lc0) * 128);
    const unsigned offtz = tid < 32 ? (unsigned)(tid * tsz + (tsz - 16)) : gemm::kOOB;
    const unsigned offta = (tid < 32 && !colsum) ? (unsigned)(tid * tsa + (tsa - 16)) : gemm::kOOB;
    RSet S0, S1;
    // requests of the stage at rows m0 ..: the two tails first (loads return in order: the rows' rho is wanted first), then the pieces.
    // Piece i of a stage (0 .. 6: Z chunks lc0, lc0 + 2, .., then A's): issued one by one BETWEEN the k tiles' products of the stage
    // before — all nine at the top of a stage took 1.5 k cycles to go out (the CU's request queue), with every wave waiting in front of
    // its first fragment read
    auto load_tails = [&](int m0, RSet& S) {
        S.tz = __builtin_amdgcn_raw_buffer_load_b64(rZ, offtz == gemm::kOOB ? gemm::kOOB : (unsigned)(m0 * tsz) + offtz, 0, 0);
        S.ta = __builtin_amdgcn_raw_buffer_load_b64(rA, offta == gemm::kOOB ? gemm::kOOB : (unsigned)(m0 * tsa) + offta, 0, 0);
    };
    auto load_piece = [&](auto ic, int m0, RSet& S) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < kRPiecesZ)
            S.Z[i] = __builtin_amdgcn_raw_buffer_load_b128(rZ, lc0 + 2 * i < ncz ? (unsigned)(m0 * tsz) + offz + 256u * i : gemm::kOOB, 0, 0);
        else
            S.A[i - kRPiecesZ] = __builtin_amdgcn_raw_buffer_load_b128(rA, lc0 + 2 * (i - kRPiecesZ) < ca_live ? (unsigned)(m0 * tsa) + offa + 256u * (i - kRPiecesZ) : gemm::kOOB, 0, 0);
    };
    auto load_stage = [&](int m0, RSet& S) {
        load_tails(m0, S);
        wg_static_for<0, kRPiecesZ + kRPiecesA>([&](auto ic) { load_piece(ic, m0, S); });
    };
    float F = 0.f, Fc = 0.f;
    // the stage in register set S into image b: the rows' rho = F / (s_Z s_A) (rho_c = F_c / s_Z) from its tails, and its pieces one by one
    auto put_rho = [&](RSet& S, int b) {
        if (tid < 32) {
            // tail = (scale, 1 if every element of the row's scaling unit is ZERO); a row past the range read zeros: rho = 0
            const float szx = __uint_as_float(S.tz.x), szy = __uint_as_float(S.tz.y), sax = __uint_as_float(S.ta.x), say = __uint_as_float(S.ta.y);
            const float sz = szy == 0.f ? szx : 0.f;
            const float sa = colsum ? 1.f : (say == 0.f ? sax : 0.f);
            const float fh = sz * sa;
            rho16[b * 32 + tid] = (_Float16)(fh > 0.f ? F / fh : 0.f);
            rhoc16[b * 32 + tid] = (_Float16)(sz > 0.f ? Fc / sz : 0.f);
        }
    };
    // EVERY piece is stored — a dead one (a chunk the job does not have) into the trash chunk behind the images: a store under a
    // condition leaves its load unwaited-for on the other path, and hipcc then guards the NEXT request into the same registers
    // with s_waitcnt vmcnt(2) — three requests in flight instead of nine (ISA, round 6)
    unsigned char* const trash = img0 + 2 * img_bytes + (tid & 255) * 16;
    auto put_piece = [&](auto ic, RSet& S, int b) {
        constexpr int i = decltype(ic)::value;
        unsigned char* dst = img0 + b * img_bytes + (tid & 255) * 16 + lc0 * kRChunk;
        if constexpr (i < kRPiecesZ)
            *reinterpret_cast<u32x4*>(lc0 + 2 * i < ncz ? dst + 2 * i * kRChunk : trash) = S.Z[i];
        else
            *reinterpret_cast<u32x4*>(lc0 + 2 * (i - kRPiecesZ) < ca_live ? dst + (ncz + 2 * (i - kRPiecesZ)) * kRChunk : trash) = S.A[i - kRPiecesZ];
    };
    auto put_stage = [&](RSet& S, int b) {
        put_rho(S, b);
        wg_static_for<0, kRPiecesZ + kRPiecesA>([&](auto ic) { put_piece(ic, S, b); });
    };
    // ---- F: the smallest s_Z s_A over the rows of this workgroup's range (the rows' tails); F_c: the smallest s_Z ----
    // (requested BEFORE stage 0: behind its 56 KB they arrived 5 k cycles later — loads return in order)
    {
        if (tid == 0) { red[0] = 0x7f7fffffu; red[1] = 0x7f7fffffu; }
        float f = 3.0e38f, fc = 3.0e38f;
        u32x2 tzv[2], tav[2];   // (<= 1 024 rows in one go; longer ranges loop)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = tid + 512 * i;
            tzv[i] = __builtin_amdgcn_raw_buffer_load_b64(rZ, r < n_rows ? (unsigned)(r * tsz + (tsz - 16)) : gemm::kOOB, 0, 0);
            tav[i] = __builtin_amdgcn_raw_buffer_load_b64(rA, (r < n_rows && !colsum) ? (unsigned)(r * tsa + (tsa - 16)) : gemm::kOOB, 0, 0);
        }
        stamp();  // 1 tails requested
        load_stage(0, S0);    // (in flight under the pass over the tails)
        load_stage(32, S1);
        stamp();  // 2 stages 0 and 1 requested
        auto take = [&](u32x2 tz, u32x2 ta, bool live) {
            // tail = (scale, 1 if every element of the row's scaling unit is ZERO): such a row takes no part — its conventional scale 1
            // would otherwise drag F down by the scale of the rows that do hold values (2^30 for gradients of 1e-5) and flush them
            const float zx = __uint_as_float(tz.x), zy = __uint_as_float(tz.y), ax = colsum ? 1.f : __uint_as_float(ta.x), ay = colsum ? 0.f : __uint_as_float(ta.y);
            if (live && zy == 0.f && ay == 0.f && zx * ax > 0.f) f = fminf(f, zx * ax);
            if (live && zy == 0.f && zx > 0.f) fc = fminf(fc, zx);
        };
        take(tzv[0], tav[0], tid < n_rows);
        take(tzv[1], tav[1], tid + 512 < n_rows);
        for (int r = tid + 1024; r < n_rows; r += 512) {
            const u32x2 tz = __builtin_amdgcn_raw_buffer_load_b64(rZ, (unsigned)(r * tsz + (tsz - 16)), 0, 0);
            const u32x2 ta = __builtin_amdgcn_raw_buffer_load_b64(rA, colsum ? gemm::kOOB : (unsigned)(r * tsa + (tsa - 16)), 0, 0);
            take(tz, ta, true);
        }
        for (int off = 32; off > 0; off >>= 1) { f = fminf(f, __shfl_xor(f, off)); fc = fminf(fc, __shfl_xor(fc, off)); }
        __syncthreads();   // (red is armed)
        if (lane == 0 && f < 3.0e38f) atomicMin(&red[0], __float_as_uint(f));   // (positive floats order like their bit patterns)
        if (lane == 0 && fc < 3.0e38f) atomicMin(&red[1], __float_as_uint(fc));
    }
    __syncthreads();
    F = __uint_as_float(red[0]);
    Fc = __uint_as_float(red[1]);
    stamp();  // 3 F known
    put_stage(S0, 0);
    stamp();  // 4 stage 0 in its image
    f32x4 acc[NTW][8], accb[NTW];
#pragma unroll
    for (int r = 0; r < NTW; ++r) {
        accb[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) acc[r][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // transpose-read addresses inside a chunk image: rows 8 lg + (li >> 2) (+ 4 for the second read: + 512 bytes), piece
    // (part 4 + h2 2 + ((li >> 1) & 1)) ^ swz, 8-byte half li & 1;  swz = 4 (lg & 1) + 2 ((li >> 3) & 1) for all of them
    const int rowb = (8 * lg + (li >> 2)) * 128 + (li & 1) * 8;
    int px[2][2];   // [part: hi | lo][h2: columns 0..15 | 16..31 of the chunk]
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) px[part][h2] = rowb + ((((part ^ (lg & 1)) << 2) + ((h2 ^ ((li >> 3) & 1)) << 1) + ((li >> 1) & 1)) << 4);
    const int n_nt = (P.N + 15) >> 4;                       // 16-row tiles of n
    const int k_lo = 16 * kt0;                              // first k column of this workgroup
    // column sums: the one fragment — row 0 (lanes li = 0) all ones, exact in f16; no lo part
    const _Float16 one = (_Float16)(li == 0 ? 1.f : 0.f);
    const h8 ones8 = h8{one, one, one, one, one, one, one, one};
    // The products are formed TRANSPOSED — the k tile of A is the matrix pipe's first operand, the n tile of Z its second — so that a
    // lane's four accumulator registers are four CONSECUTIVE k columns of one row n of the slab: one 16-byte store per tile in the
    // epilogue (as rows of n per lane they were 96 scattered 4-byte stores per wave, 16 k of the workgroup's 86 k cycles).
    // One stage: the products over image b, and BETWEEN the k tiles' products, piece by piece: the stage in register set X (requested a
    // stage ago) goes to the other image (last read before the previous barrier), the stage after it is requested into set Y (whose
    // pieces went to this image a stage ago) — two stages in flight, no store phase of its own (it was 1.2 k of a stage's 4.4 k cycles).
    auto compute = [&](int b, RSet& X, RSet& Y, int m0_next) {
        put_rho(X, b ^ 1);
        load_tails(m0_next, Y);
        const unsigned char* Zt = img0 + b * img_bytes;
        const unsigned char* At = Zt + ncz * kRChunk;
        const h8 rho = *reinterpret_cast<const h8*>(rho16 + b * 32 + 8 * lg);
        h8 zh[NTW], zl[NTW];
        h8 bh, bl;
        if (kt_live > 0) {   // (uniform; a column-sum job has no products)
            bh = tr_pair(At, px[0][koff]); bl = tr_pair(At, px[1][koff]);   // (the first k tile's fragments with the Z fragments)
#pragma unroll
            for (int r = 0; r < NTW; ++r) {
                const int nt = wave + 8 * r;
                const unsigned char* zc = Zt + (nt >> 1) * kRChunk;
                zh[r] = tr_pair(zc, px[0][nt & 1]) * rho;
                zl[r] = tr_pair(zc, px[1][nt & 1]) * rho;
            }
        }
        wg_static_for<0, 8>([&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            // the next k tile's fragments are requested in front of this tile's products (hipcc left to itself hoists all eight tiles'
            // reads — 64 registers — or none)
            h8 nbh = bh, nbl = bl;
            if (kt + 1 < kt_live) {   // (uniform)
                const unsigned char* ac = At + ((kt + 1 + koff) >> 1) * kRChunk;
                nbh = tr_pair(ac, px[0][(kt + 1 + koff) & 1]); nbl = tr_pair(ac, px[1][(kt + 1 + koff) & 1]);
            }
            // (piece kt moves whether or not k tile kt is live — NO memory instruction under a condition: hipcc then counts the loads in
            //  flight exactly, s_waitcnt vmcnt(8) in front of every store, instead of the smallest count over the paths)
            if constexpr (kt < kRPiecesZ + kRPiecesA) { put_piece(ktc, X, b ^ 1); load_piece(ktc, m0_next, Y); }
            __builtin_amdgcn_sched_barrier(0);
            if (kt < kt_live) {   // (uniform)
                // (the three products of a tile are a dependent chain on its accumulator: the tiles' chains interleaved)
#pragma unroll
                for (int r = 0; r < NTW; ++r) acc[r][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, zh[r], acc[r][kt], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < NTW; ++r) acc[r][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, zh[r], acc[r][kt], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < NTW; ++r) acc[r][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, zl[r], acc[r][kt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            bh = nbh; bl = nbl;
        });
        if (colsum || ride_b) {   // (uniform) the column sums: the Z fragments once more (the products' are dead by now), scaled by rho_c
            const h8 rhoc = colsum ? rho : *reinterpret_cast<const h8*>(rhoc16 + b * 32 + 8 * lg);
#pragma unroll
            for (int r = 0; r < NTW; ++r) {
                const int nt = wave + 8 * r;
                const unsigned char* zc = Zt + (nt >> 1) * kRChunk;
                const h8 rh = tr_pair(zc, px[0][nt & 1]) * rhoc, rl = tr_pair(zc, px[1][nt & 1]) * rhoc;
                // (a column-sum job's kt_live is 0: its accb is the one accumulator set it uses — no select between two register arrays, which
                //  would put both in scratch memory)
                accb[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, rh, accb[r], 0, 0, 0);
                accb[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, rl, accb[r], 0, 0, 0);
            }
        }
    };
    __syncthreads();   // image 0 and its rho are in place
    // stage s computes from image s & 1; S1 holds stage s + 1 at even s, S0 at odd s
    const int n_st = (n_rows + 31) >> 5;
    int st = 0;
    for (; st + 1 < n_st; st += 2) {
        stamp();  // 5 + 3 st
        compute(0, S1, S0, 32 * (st + 2));
        stamp();  //   products, stores and requests issued
        __syncthreads();
        stamp();  //   barrier
        compute(1, S0, S1, 32 * (st + 3));
        stamp();
        __syncthreads();
        stamp();
    }
    if (st < n_st) {   // (an odd count's last stage: what it stores and requests is past the range — zeros, no memory traffic)
        stamp();
        compute(0, S1, S0, 32 * (st + 2));
        stamp();
    }
    // D fragment (transposed product): lane (li, lg) holds row n = 16 nt + li, columns k_lo + 16 kt + 4 lg .. + 3 of this workgroup's k column group
    const float iF = (F > 0.f && F < 3.0e38f) ? 1.f / F : 0.f;
    float* slab = P.slab + (long long)split * P.slab_stride;
    if (colsum) {   // (uniform) column 0 of slabs [N][4]: row 0 of the transposed tile — lanes lg = 0, register 0
#pragma unroll
        for (int r = 0; r < NTW; ++r) {
            const int n = 16 * (wave + 8 * r) + li;
            if (wave + 8 * r < n_nt && lg == 0 && n < P.N) slab[(long long)n * P.ldk] = accb[r][0] * iF;
        }
    } else {
#pragma unroll
        for (int r = 0; r < NTW; ++r) {
            const int nt = wave + 8 * r, n = 16 * nt + li;
            if (nt >= n_nt || n >= P.N) continue;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                const int k = k_lo + 16 * kt + 4 * lg;   // (a quad that starts below K ends below ldk = K rounded up to 4: the padding is never read)
                if (kt < kt_live && k < P.K)
                    *reinterpret_cast<float4*>(slab + (long long)n * P.ldk + k) = make_float4(acc[r][kt][0] * iF, acc[r][kt][1] * iF, acc[r][kt][2] * iF, acc[r][kt][3] * iF);
            }
        }
    }
    stamp();  // slabs stored (issued)
    if (ride_b && lg == 0) {   // the column sums: column 0 of slabs [N][4]
        const float iFc = (Fc > 0.f && Fc < 3.0e38f) ? 1.f / Fc : 0.f;
        float* sb = P.slab_b + (long long)split * P.N * 4;
#pragma unroll
        for (int r = 0; r < NTW; ++r) {
            const int n = 16 * (wave + 8 * r) + li;
            if (wave + 8 * r < n_nt && n < P.N) sb[n * 4] = accb[r][0] * iFc;
        }
    }
}

}  // namespace wg16

// ---- host side ------------------------------------------------------------------------------------------------------
extern thread_local long long* g_debug_stamps;
size_t wsplit16_bytes(int64_t M, int64_t C) {  // one operand: blocks + scales
    const int64_t n_ct = (C + 63) / 64, n_chunks = (M + 31) / 32;
    return (size_t)(n_ct * n_chunks) * wg16::kBlk + (((size_t)(n_ct * n_chunks) * 4 + 255) & ~size_t(255));
}

void wsplit16_job(WSplitJob* j, int64_t M, int C, const float* A1, int64_t lda1, const int* g1, int K1, const float* A2, int64_t lda2,
                  const int* g2, int K2, int ones, void* ws) {
    memset(j, 0, sizeof(*j));
    j->M = M; j->C = C;
    j->A1 = A1; j->lda1 = lda1; j->g1 = g1; j->K1 = K1;
    j->A2 = K2 ? A2 : A1; j->lda2 = K2 ? lda2 : lda1; j->g2 = K2 ? g2 : g1; j->K2 = K2;
    j->ones = ones;
    j->n_ct = (C + 63) / 64; j->n_chunks = (int)((M + 31) / 32);
    j->out = static_cast<unsigned char*>(ws);
    j->scales = reinterpret_cast<float*>(j->out + (size_t)j->n_ct * j->n_chunks * wg16::kBlk);
}

int launch_wsplit16(WSplitArgs& a, hipStream_t s) {
    int total = 0;
    for (int i = 0; i < a.n_jobs; ++i) {
        a.job[i].wg0 = total;
        total += a.job[i].n_ct * ((a.job[i].n_chunks + 3) / 4);
    }
    if (total == 0) return DMPNN_OK;
    hipLaunchKernelGGL(wg16::k_wsplit16, dim3((unsigned)total), dim3(256), 0, s, a);
    DMPNN_CHECK_LAUNCH("k_wsplit16");
    return DMPNN_OK;
}

// rows of both operands must be 8-byte loadable as pairs of columns: even column counts and leading dimensions, 8-byte aligned bases
bool wgrad16_operand_ok(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2) {
    auto ok8 = [](const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0 && ld % 2 == 0; };
    if (K1 <= 0 || K1 % 2 || !ok8(A1, lda1)) return false;
    if (K2 > 0 && (K2 % 2 || !ok8(A2, lda2))) return false;
    return true;
}

// the product of two split operands (Z: [M][N], A: [M][Kt]) into `splits` slabs; chunks_per_split a multiple of the stage
WProdPlan plan_wgrad16(int64_t M, int N, int Kt) {
    WProdPlan p;
    p.n_nt = (N + 63) / 64; p.n_kt = (Kt + 63) / 64; p.n_chunks = (int)((M + 31) / 32);
    // workgroups of a product launch: 3 per CU for long reductions (measured best of {256 .. 1536} at 36 000+ rows); 2 per CU up to
    // 768 chunks (24 576 rows: the 512-molecule training step) — fewer row splits are fewer slabs for the reduce kernel, which is the
    // larger effect there (block step 185.6 -> 180.2 us, model step 240.2 -> 234.7; at 355 702 rows 512 would cost 3 %: scripts/r3_ab_train.sh)
    const int target = p.n_chunks <= 768 ? 512 : 768;
    int splits = target / (p.n_nt * p.n_kt);
    if (splits < 1) splits = 1;
    int cps = (p.n_chunks + splits - 1) / splits;
    cps = (cps + wg16::kStage - 1) / wg16::kStage * wg16::kStage;
    if (cps < wg16::kStage) cps = wg16::kStage;
    p.chunks_per_split = cps;
    p.splits = (p.n_chunks + cps - 1) / cps;
    if (p.splits < 1) p.splits = 1;
    p.ldk = (Kt + 3) / 4 * 4;
    p.slab_stride = (int64_t)N * p.ldk;
    return p;
}

void wgrad16_add(WProdJobs* jobs, const WSplitJob& Z, const WSplitJob& A, const WProdPlan& p, int N, int Kt, float* slab) {
    WProdArgs& a = jobs->job[jobs->n_jobs];
    memset(&a, 0, sizeof(a));
    a.Z = Z.out; a.sZ = Z.scales; a.A = A.out; a.sA = A.scales;
    a.n_nt = p.n_nt; a.n_kt = p.n_kt; a.n_chunks = p.n_chunks; a.chunks_per_split = p.chunks_per_split; a.splits = p.splits;
    a.N = N; a.Kt = Kt; a.slab = slab; a.ldk = p.ldk; a.slab_stride = p.slab_stride;
    const int total = p.n_nt * p.n_kt * p.splits;
    jobs->wg0[jobs->n_jobs + 1] = jobs->wg0[jobs->n_jobs] + (total + 7) / 8 * 8;
    ++jobs->n_jobs;
}

// fp32 rows -> split rows, up to 4 operands per launch (K1 + K2 <= 512 columns each; ts >= ceil(K / 32) * 128 + 16)
int launch_rows2sr(const SRJob* J, int n, hipStream_t s) {
    wg16::SRJobs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) {
        if (J[i].M <= 0) continue;
        const int K = J[i].K1 + J[i].K2;
        if (a.n_jobs >= 4 || K <= 0 || K > 512 || J[i].ts < ((K + 31) / 32) * 128 + 16) { set_error("rows2sr: at most 4 operands of <= 512 columns per launch"); return DMPNN_EINVAL; }
        a.job[a.n_jobs] = J[i];
        a.wg0[a.n_jobs + 1] = a.wg0[a.n_jobs] + (int)((J[i].M + 3) / 4);
        ++a.n_jobs;
    }
    if (a.n_jobs == 0) return DMPNN_OK;
    hipLaunchKernelGGL(wg16::k_rows2sr, dim3((unsigned)a.wg0[a.n_jobs]), dim3(256), 0, s, a);
    DMPNN_CHECK_LAUNCH("k_rows2sr");
    return DMPNN_OK;
}

// products over split-row operands (k_wgrad16r).  plan: the row splits of one job — every split is a slab the reduce kernel reads
static int wgradr_target_wgs() {
    static const int wgs = [] {
        const char* e = getenv("DMPNN_WGRADR_WGS");   // (A/B runs: workgroups of a product launch)
        int v = e ? atoi(e) : 0;
        if (v > 0) return v;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) return cus;
        return 256;
    }();
    return wgs;
}

// the k column groups of a job as its 16-column tiles dealt evenly — allowed when every group lies within four 32-column chunks of A
static bool wgradr_even_kt(int K) {
    const int nkt = (K + 15) / 16, n_kg = (K + 127) / 128;
    if (n_kg < 1) return false;
    for (int g = 0; g < n_kg; ++g) {
        const int t0 = g * nkt / n_kg, t1 = (g + 1) * nkt / n_kg;
        if (((t1 + 1) >> 1) - (t0 >> 1) > 4) return false;
    }
    return true;
}
// relative cost of one 32-row stage of a workgroup of a job with K columns of A (0: a column-sum job): its largest k column group's
// 16-column tiles (<= 8; 24 MFMAs per tile and wave), not below the stage's loads (a column-sum job: ~3 tiles' worth)
static int wgradr_stage_cost(int K) {
    if (K <= 1) return 3;
    const int nkt = (K + 15) / 16, nca = (K + 31) / 32, n_kg = (K + 127) / 128;
    const int kt = wgradr_even_kt(K) ? (nkt + n_kg - 1) / n_kg : 2 * ((nca + n_kg - 1) / n_kg);
    return kt < 3 ? 3 : (kt > 8 ? 8 : kt);
}

void wgrad16r_plan_launch(const int64_t* M, const int* K, int n, int reserve, int* rows_out) {
    const int cus = wgradr_target_wgs();
    double units = 0;   // rows of a cost-8 job the launch amounts to
    int64_t m_max = 0;
    for (int j = 0; j < n; ++j) {
        rows_out[j] = 32;
        if (M[j] <= 0) continue;
        units += (double)M[j] * ((K[j] + 127) / 128 > 0 ? (K[j] + 127) / 128 : 1) * wgradr_stage_cost(K[j]) / 8.0;
        m_max = M[j] > m_max ? M[j] : m_max;
    }
    if (units <= 0) return;
    // one workgroup per CU while that keeps a workgroup under ~1 024 rows; beyond, whole rounds of workgroups of >= 512 rows (the CUs
    // balance the jobs' different stage costs among themselves; the slabs stay a small fraction of the operand bytes)
    int64_t rounds = (int64_t)(units / (512.0 * cus));
    if (rounds < 1) rounds = 1;
    int64_t budget = rounds * cus - reserve;
    if (budget < 8) budget = 8;
    int64_t R0 = (int64_t)(units / (double)budget);
    R0 = (R0 + 31) / 32 * 32;
    if (R0 < 32) R0 = 32;
    const int64_t R_max = ((m_max + 31) / 32 * 32) * 8;
    for (;; R0 += 32) {
        int64_t wg = 0;
        for (int j = 0; j < n; ++j) {
            if (M[j] <= 0) continue;
            int64_t Rj = (R0 * 8 / wgradr_stage_cost(K[j]) + 31) / 32 * 32;
            if (Rj > ((int64_t)1 << 20)) Rj = (int64_t)1 << 20;
            rows_out[j] = (int)Rj;
            const int n_kg = (K[j] + 127) / 128 > 0 ? (K[j] + 127) / 128 : 1;
            wg += (n_kg * ((M[j] + Rj - 1) / Rj) + 7) / 8 * 8;   // (every job's workgroups are padded to a multiple of 8: the XCD-aware order)
        }
        if (wg <= budget || R0 >= R_max) break;
    }
}

WProdRPlan plan_wgrad16r(int64_t M, int N, int K, int rows_per_split) {
    WProdRPlan p;
    p.n_kg = (K + 127) / 128;
    int64_t rps;
    if (rows_per_split > 0) {
        rps = (rows_per_split + 31) / 32 * 32;
    } else {
        int splits = 256 / p.n_kg;                  // ~256 workgroups per product, two or more per CU over the jobs of a launch
        if (splits < 1) splits = 1;
        rps = (M + splits - 1) / splits;
        rps = (rps + 31) / 32 * 32;
    }
    if (rps < 32) rps = 32;
    p.rows_per_split = (int)rps;
    p.splits = (int)((M + rps - 1) / rps);
    if (p.splits < 1) p.splits = 1;
    p.ldk = (K + 3) / 4 * 4;
    p.slab_stride = (int64_t)N * p.ldk;
    return p;
}

// `n` jobs in one launch (<= kWProdRMaxJobs; more: several launches).  A job with A == nullptr: the column sums of Z (K = 1).
int launch_wgrad16r(const WProdRJob* J, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += kWProdRMaxJobs) {
        const int nn = n - i0 < kWProdRMaxJobs ? n - i0 : kWProdRMaxJobs;
        wg16::WProdRJobs P;
        memset(&P, 0, sizeof(P));
        int wg = 0, ncz_max = 0, nj = 0, nt_max = 1;
        for (int i = 0; i < nn; ++i) {
            const WProdRJob& q = J[i0 + i];
            if (q.M <= 0) continue;   // (its reduce job sums zero slabs: the caller zero-fills)
            const int K = q.A ? q.K : 1;
            const int ncz = (q.N + 31) / 32, nca = (K + 31) / 32;
            const WProdRPlan& p = q.plan;
            if ((q.N + 15) / 16 > 8 * wg16::kRTW8 || ncz > 10 || q.tsz < ncz * 128 + 16 || (q.A && q.tsa < nca * 128 + 16) ||
                (int64_t)p.rows_per_split * (q.tsz > q.tsa ? q.tsz : q.tsa) > ((int64_t)1 << 31)) {
                set_error("wgrad16r: d_h <= 320, whole chunks + tail per operand row");
                return DMPNN_EINVAL;
            }
            wg16::WProdR& a = P.job[nj++];
            a.Z = q.Z; a.A = q.A; a.slab = q.slab; a.slab_b = q.A ? q.slab_b : nullptr; a.M = q.M; a.N = q.N; a.K = K;
            a.tsz = q.tsz; a.ncz = ncz; a.tsa = q.A ? q.tsa : q.tsz; a.nca = nca;
            a.n_kg = p.n_kg; a.splits = p.splits; a.rows_per_split = p.rows_per_split; a.per8 = (p.n_kg * p.splits + 7) / 8;
            a.even_kt = wgradr_even_kt(K) ? 1 : 0;
            a.ldk = p.ldk; a.slab_stride = p.slab_stride;
            a.wg0 = wg; wg += a.per8 * 8;
            if (ncz > ncz_max) ncz_max = ncz;
            if ((q.N + 15) / 16 > nt_max) nt_max = (q.N + 15) / 16;
        }
        if (nj == 0) continue;
        P.n_jobs = nj;
        const size_t lds = 512 + 2 * (size_t)(ncz_max + 4) * wg16::kRChunk + wg16::kRChunk;   // rho, rho_c, red | two stage images | the trash chunk
        P.dbg = g_debug_stamps;
        const int ntw = (nt_max + 7) / 8;   // 16-row tiles of n per wave: the launch's widest job (1 .. 3, checked above)
        auto go = [&](auto kern, size_t& attr_set) -> int {
            if (attr_set < lds) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) { set_error("hipFuncSetAttribute(k_wgrad16r, %zu B LDS): %s", lds, hipGetErrorString(e)); return DMPNN_EHIP; }
                attr_set = lds;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)wg), dim3(512), lds, s, P);
            return DMPNN_OK;
        };
        static size_t attr1 = 0, attr2 = 0, attr3 = 0;
        if (ntw <= 1) DMPNN_TRY(go(&wg16::k_wgrad16r<1>, attr1));
        else if (ntw == 2) DMPNN_TRY(go(&wg16::k_wgrad16r<2>, attr2));
        else DMPNN_TRY(go(&wg16::k_wgrad16r<3>, attr3));
        DMPNN_CHECK_LAUNCH("k_wgrad16r");
    }
    return DMPNN_OK;
}

int launch_wgrad16(const WProdJobs& jobs, hipStream_t s) {
    if (jobs.n_jobs == 0) return DMPNN_OK;
    hipLaunchKernelGGL(wg16::k_wgrad16, dim3((unsigned)jobs.wg0[jobs.n_jobs]), dim3(256), 0, s, jobs);
    DMPNN_CHECK_LAUNCH("k_wgrad16");
    return DMPNN_OK;
}

}  // namespace dmpnn
