// K6 for molecules beyond the whole-forward tile (round 4): the backward pass of the per-step FUSED route on the f16 matrix pipe
// (dmpnn_step16_impl.hpp) as step kernels of the same shape — one launch per depth step over the plan's 48-row tiles of whole
// destination atoms — reading what the LEAN training forward kept (dmpnn_step16.hip: fused16_lean): the split message rows
// M^(t) of every step, the split K1 operand x = [V[src] || E], one sign bit per element of H0 / H^(t), the per-atom sums Mv.
//
//   with T[r] = gM^(t+1)[rev r] scattered there by the previous launch (mixins.py:11-18 transposed: the message backward is the
//   forward's segment form with read row rev r):
//       gH^(t)[r] = S'[dst r] - T[r],   S'[v] = sum of T over the rows of v          (first launch: gH^(T-1)[r] = gMv[dst r])
//       gZ^(t)    = gH^(t) * tau'(z^(t))                                              (tau' from the kept sign bits: ReLU class)
//       gM^(t)    = gZ^(t) W_h            -> T_next[rev r] = gM^(t)[r]                (base.py:135-141 transposed; f16 pipe, exact split)
//   and gZ^(t) leaves as the OPERAND of the weight-gradient products (dmpnn_wgrad16.hip, k_wgrad16r) — as split rows, the very A tile
//   of the contraction copied out row by row:
//       gW_h = sum_t gZ^(t)^T M^(t),   gW_i = (sum_t gZ^(t))^T x = sum_t gZ^(t)^T x   (no gH0 tensor: products over the steps)
//
// It replaces, per depth step of the per-step general route, one contraction launch (k_rows16: read gZ, write gM), one
// k_edge_bwd launch (read gM, H, gH0; write gZ, gH0) and the gZ half of k_wsplit16 (read gZ, write blocks) by ONE launch that
// reads T once and writes T_next and the gZ rows once.  The products' other operands are the forward's kept split rows (M^(t), x) as
// they are: nothing is re-blocked (an earlier form of this file wrote tile-packed 8 KB blocks and converted the kept rows to them:
// 0.57 ms per step of 40-atom x 4 096 for k_rows2blk alone, profiles/r04_train40_kernel_stats.txt).
#include <stdlib.h>
#include <string.h>

#include "dmpnn_step16_impl.hpp"

namespace dmpnn {
namespace bstep16 {

using gemm::f32x4;
using gemm::kAtomCache;
using gemm::kOOB;
using gemm::rsrc_t;
using mega16::h4;
using mega16::h8;
using mega16::scale_for;
using mega16::SplitW;
using step16::BM;

constexpr int RT = 3;
struct BStepK {
    int M, N;                                  // directed edges (rows), d_h
    const int* hdr;                            // plan header: flags, the number of row tiles actually used
    int n_atoms;
    const int* tile_row; const int* tile_atom; const int* row_ptr; const int* revp; const int* dstp;
    const float* Tin; int ldt;                 // [M][ldt] fp32: Tin[r] = gM_next[rev r] (message mode), or null:
    const float* gMv; int ldg;                 // [V][ldg] fp32: gH[r] = gMv[dst r]      (gather mode: the first launch)
    const unsigned char* bits; int bstride;    // [tau(z) > 0] of this site's rows, [M][bstride] bytes (null: tau' = 1)
    float neg;                                 // tau' where the bit is 0: 0 (ReLU), the slope (LeakyReLU)
    unsigned char* Zrows; int tsz;             // gZ as a product operand (k_wgrad16r): split ROWS [M][tsz] — the contraction's own A tile, row by row, the tile's scale in the tails
    SplitW W;                                  // pre-split W_h^T (fragment-major); p null: no contraction (the last site: gZ^(0) only)
    float* Tout; int ldo;                      // Tout[revp[r]] = gM[r]
    unsigned qmagic;                           // ceil(2^32 / (N / 4))
    int poison_mask;
    int tile_bytes;
};

// One tile of whole destination atoms.  4 waves x WN column tiles of 16 (d_h <= 320), two workgroups per CU.
template <int WN>
__global__ __launch_bounds__(256, 2) void k_bstep16(BStepK g) {
    constexpr int NT = 256, NW = 4;
    constexpr int BN = 16 * WN * NW, LDC = BN + 4, TS = BN * 4 + 16;
    static_assert(LDC * 4 == TS, "fp32 tile rows and split tile rows share one stride");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* T = reinterpret_cast<float*>(lds);                      // [BM][LDC] fp32: input rows -> gH -> gZ; later the fp32 result tile
    unsigned char* Ag = lds;                                       // [BM][TS] split A tile of the contraction (overlays it)
    unsigned char* Bt = lds + g.tile_bytes;                        // [BM][BN / 8] sign bytes of the tile's rows
    int* meta = reinterpret_cast<int*>(Bt + BM * (BN / 8));        // [BM] reverse rows | [kAtomCache + 1] row pointers
    int* rp = meta + BM;
    unsigned* maxbits = reinterpret_cast<unsigned*>(rp + kAtomCache + 1);

    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int li = lane & 15, lg = lane >> 4;
    const int t = blockIdx.x;
    const int n_tiles = g.hdr[DMPNN_HDR_NTILES];
    if (t >= n_tiles) return;                                      // (launch bound: the products read the same count)
    const int rs = g.tile_row[t], re = g.tile_row[t + 1];
    const int va = g.tile_atom[t], vb = g.tile_atom[t + 1];
    const int nrows = re - rs;
    if (nrows < 0 || nrows > BM) return;                           // (cannot happen with a valid tile table)
    const bool poison = (g.hdr[DMPNN_HDR_FLAGS] & g.poison_mask) != 0;
    const int qn = g.N >> 2;
    const bool gather = g.Tin == nullptr;
    if (tid == 0) maxbits[0] = 0u;
    // ---- metadata ----
    if (tid < nrows) meta[tid] = gather ? g.dstp[rs + tid] : (g.Tout ? g.revp[rs + tid] : 0);
    {   // sign bytes of the tile's rows (rows are contiguous: nrows * bstride bytes)
        constexpr int BB = BN / 8;
        for (int it = tid; it < nrows * BB; it += NT) Bt[it] = g.bits ? g.bits[(long long)rs * g.bstride + it] : (unsigned char)0xFF;
    }
    // ---- the tile's input rows by LDS-DMA (no registers, one round trip): fp32 rows at stride LDF = N in the tile region.  Message
    // mode: the rows are contiguous in memory; gather mode: lane-linear in LDS, every lane from its own atom row ----
    const int LDF = g.N;                                           // (ldt == N: d_h % 4 == 0 on this route, so ldh = d_h)
    {
        const unsigned row_b = (unsigned)g.N * 4u;
        const unsigned nbytes = (unsigned)nrows * row_b;
        const int n_inst = (int)((nbytes + 1023u) >> 10);
        if (!gather) {
            const rsrc_t rT = gemm::make_rsrc(g.Tin + (long long)rs * g.ldt, nbytes);
            for (int i = wave; i < n_inst; i += NW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rT, (__attribute__((address_space(3))) void*)(lds + i * 1024), 16,
                                                         (unsigned)(i * 1024 + lane * 16), 0, 0, 0);
        } else {
            __syncthreads();                                       // meta = the rows' destination atoms
            const rsrc_t rG = gemm::make_rsrc(g.gMv, gemm::clamp_bytes((long long)g.n_atoms * g.ldg * 4));
            for (int i = wave; i < n_inst; i += NW) {
                const unsigned o = (unsigned)(i * 1024 + lane * 16);
                const unsigned r = __umulhi(o >> 4, g.qmagic);     // o / row_b  (row_b = 16 qn)
                const unsigned cb = o - r * row_b;
                const unsigned off = r < (unsigned)nrows ? (unsigned)meta[r] * (unsigned)g.ldg * 4u + cb : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rG, (__attribute__((address_space(3))) void*)(lds + i * 1024), 16, off, 0, 0, 0);
            }
        }
    }
    if (gather) {
        // meta now becomes the reverse rows (the scatter of the result): every wave has issued its loads
        __syncthreads();
        if (tid < nrows) meta[tid] = g.Tout ? g.revp[rs + tid] : 0;
    }
    __syncthreads();   // (the barrier's release waits for the DMA)
    // ---- gH, gZ in place ----
    const float nanv = __int_as_float(0x7fc00000);
    float mx = 0.f;
    auto mask4 = [&](float4 gh, int r, int q) -> float4 {
        const unsigned b = (unsigned)Bt[r * (BN / 8) + (q >> 1)] >> ((q & 1) * 4);
        float4 z;
        z.x = (b & 1u) ? gh.x : g.neg * gh.x;
        z.y = (b & 2u) ? gh.y : g.neg * gh.y;
        z.z = (b & 4u) ? gh.z : g.neg * gh.z;
        z.w = (b & 8u) ? gh.w : g.neg * gh.w;
        if (poison) z = make_float4(nanv, nanv, nanv, nanv);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(z.x), fabsf(z.y)), fmaxf(fabsf(z.z), fabsf(z.w))));
        return z;
    };
    if (gather) {
        for (int it = tid; it < nrows * qn; it += NT) {
            const int r = qn == 1 ? it : (int)__umulhi((unsigned)it, g.qmagic), q = it - r * qn;
            float4* cell = reinterpret_cast<float4*>(T + r * LDF + 4 * q);
            *cell = mask4(*cell, r, q);
        }
    } else {
        for (int a0 = va; a0 < vb; a0 += kAtomCache) {
            const int na = vb - a0 < kAtomCache ? vb - a0 : kAtomCache;
            __syncthreads();
            if (tid <= na) rp[tid] = g.row_ptr[a0 + tid] - rs;
            __syncthreads();
            for (int it = tid; it < na * qn; it += NT) {
                const int al = qn == 1 ? it : (int)__umulhi((unsigned)it, g.qmagic), q = it - al * qn;
                const int r0 = rp[al], r1 = rp[al + 1];
                if (r1 - r0 <= 4) {   // an atom of a molecule: its rows requested together, summed in increasing row order
                    float4 y[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = *reinterpret_cast<const float4*>(T + (r0 + i < r1 ? r0 + i : (r1 > r0 ? r0 : 0)) * LDF + 4 * q);
                    float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (r1 > r0) S = y[0];
#pragma unroll
                    for (int i = 1; i < 4; ++i)
                        if (r0 + i < r1) { S.x += y[i].x; S.y += y[i].y; S.z += y[i].z; S.w += y[i].w; }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (r0 + i < r1)
                            *reinterpret_cast<float4*>(T + (r0 + i) * LDF + 4 * q) =
                                mask4(make_float4(S.x - y[i].x, S.y - y[i].y, S.z - y[i].z, S.w - y[i].w), r0 + i, q);
                    continue;
                }
                float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int r = r0; r < r1; ++r) {
                    const float4 y = *reinterpret_cast<const float4*>(T + r * LDF + 4 * q);
                    S.x += y.x; S.y += y.y; S.z += y.z; S.w += y.w;
                }
                for (int r = r0; r < r1; ++r) {
                    float4* cell = reinterpret_cast<float4*>(T + r * LDF + 4 * q);
                    const float4 y = *cell;
                    *cell = mask4(make_float4(S.x - y.x, S.y - y.y, S.z - y.z, S.w - y.w), r, q);
                }
            }
        }
    }
    // ---- tile scale ----
    {
        int u = (int)__float_as_uint(mx);
        for (int off = 32; off > 0; off >>= 1) u = max(u, __shfl_xor(u, off));
        if (lane == 0) atomicMax(&maxbits[0], (unsigned)u);
    }
    __syncthreads();
    const float s = poison ? 1.f : scale_for(__uint_as_float(maxbits[0]));
    const bool contract_on = g.W.p != nullptr;
    if (!contract_on && !g.Zrows) return;
    // ---- the split A tile of the contraction.  Item = (8 rows, 4 columns): read as fp32 BEFORE anyone overwrites the region with
    // the split tile (the two share the LDS, at different row strides) ----
    constexpr int NQ = BN / 4;                        // column quads of the padded row
    constexpr int ITEMS = (BM / 8) * NQ;              // 6 row groups x NQ
    constexpr int IPT = (ITEMS + NT - 1) / NT;        // items per thread (2 at BN = 320)
    float4 v[IPT][8];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int it = tid + k * NT;
        const int gq = it / NQ, q = it - gq * NQ;     // row group (8 rows), column quad
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int r = 8 * gq + jj;
            const bool ok = it < ITEMS && r < nrows && 4 * q < g.N;
            const float4 x = *reinterpret_cast<const float4*>(T + (ok ? r * LDF + 4 * q : 0));
            v[k][jj] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int it = tid + k * NT;
        if (it >= ITEMS) continue;
        const int gq = it / NQ, q = it - gq * NQ;
        const int col4 = 4 * q;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            h4 hi, lo;
            mega16::split4(v[k][jj], s, hi, lo);
            unsigned char* p = Ag + (8 * gq + jj) * TS + (col4 >> 5) * 128 + (col4 & 31) * 2;
            *reinterpret_cast<h4*>(p) = hi;
            *reinterpret_cast<h4*>(p + 64) = lo;
        }
    }
    __syncthreads();   // the split A tile is complete
    if (g.Zrows) {
        // gZ as a product operand = this tile, row by row (k_wgrad16r reads split rows as they are): 16-byte pieces, whole 128-byte lines
        const int npc = ((g.N + 31) >> 5) * 8;   // pieces of a row's live chunks
        for (int it = tid; it < nrows * npc; it += NT) {
            const int r = it / npc, pc = it - r * npc;
            *reinterpret_cast<uint4*>(g.Zrows + (long long)(rs + r) * g.tsz + pc * 16) = *reinterpret_cast<const uint4*>(Ag + r * TS + pc * 16);
        }
        // (tail: the tile's scale, and 1 where the whole tile is zero — such rows take no part in the product's common factor)
        if (tid < nrows) *reinterpret_cast<float4*>(g.Zrows + (long long)(rs + tid) * g.tsz + (g.tsz - 16)) =
            make_float4(s, (poison || __uint_as_float(maxbits[0]) > 0.f) ? 0.f : 1.f, 0.f, 0.f);
    }
    if (!contract_on) return;

    // ---- gM = gZ W_h: barrier-free MFMA loop, A fragments from the LDS tile, weight fragments from L2 as a ring (k_step16's loop) ----
    asm volatile("" : "+v"(tid));
    lane = tid & 63; wave = tid >> 6; li = lane & 15; lg = lane >> 4;
    const int NTL = (g.N + 15) / 16;
    const rsrc_t rW = gemm::make_rsrc(g.W.p, (unsigned)(NTL * g.W.nc * 2048));
    unsigned offB[WN];
    float isw[WN];
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int tile = wave * WN + ct;
        offB[ct] = tile < NTL ? (unsigned)tile * (unsigned)(g.W.nc * 2048) + (unsigned)lane * 16u : kOOB;
        const int col = wave * (16 * WN) + ct * 16 + li;
        isw[ct] = g.W.inv_scale[col < g.N ? col : 0];
    }
    f32x4 acc[RT][WN];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    h8 ah[2][RT], al[2][RT], bh[WN], bl[WN];
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
    auto ring = [&](int c, h8 (&xh)[RT], h8 (&xl)[RT], h8 (&nh)[RT], h8 (&nl)[RT]) {
        const bool more = c + 1 < n_chunks;
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[rt], bh[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[rt], bl[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[rt], bh[ct], acc[rt][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned o = (more && offB[ct] != kOOB) ? offB[ct] + (unsigned)(c + 1) * 2048u : kOOB;
            bh[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o, 0, 0));
            bl[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o == kOOB ? kOOB : o + 1024u, 0, 0));
            if (ct == (WN > 1 ? WN - 2 : 0)) read_afrags(c + 1, nh, nl);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        bh[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, offB[ct], 0, 0));
        bl[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, offB[ct] == kOOB ? kOOB : offB[ct] + 1024u, 0, 0));
    }
    read_afrags(0, ah[0], al[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma nounroll
    for (int c = 0; c < n_chunks; c += 2) {
        ring(c, ah[0], al[0], ah[1], al[1]);
        if (c + 1 < n_chunks) ring(c + 1, ah[1], al[1], ah[0], al[0]);
    }
    // ---- result: split domain -> fp32 tile -> rows of T_next at the reverse rows ----
    const float is = 1.f / s;
    __syncthreads();   // every wave is done with the A tile the fp32 tile overlays
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int cl = wave * (16 * WN) + ct * 16 + li;
        const float f = isw[ct] * is;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(rt * 16 + lg * 4 + r) * LDC + cl] = acc[rt][ct][r] * f;
    }
    __syncthreads();
    for (int it = tid; it < nrows * qn; it += NT) {
        const int r = qn == 1 ? it : (int)__umulhi((unsigned)it, g.qmagic), q = it - r * qn;
        float4 z = *reinterpret_cast<const float4*>(T + r * LDC + 4 * q);
        if (poison) z = make_float4(nanv, nanv, nanv, nanv);
        *reinterpret_cast<float4*>(g.Tout + (long long)meta[r] * g.ldo + 4 * q) = z;
    }
}

template <int WN>
static int launch_bstep(const BStepK& g0, int n_tiles, hipStream_t s) {
    BStepK g = g0;
    constexpr int BN = 64 * WN;
    g.tile_bytes = BM * (BN * 4 + 16);
    const size_t lds = (size_t)g.tile_bytes + (size_t)BM * (BN / 8) + (size_t)(BM + kAtomCache + 1) * sizeof(int) + 64;
    static size_t attr_set = 0;
    if (attr_set < lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bstep16<WN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(k_bstep16<%d>, %zu B LDS): %s", WN, lds, hipGetErrorString(e)); return DMPNN_EHIP; }
        attr_set = lds;
    }
    hipLaunchKernelGGL((k_bstep16<WN>), dim3((unsigned)n_tiles), dim3(256), lds, s, g);
    DMPNN_CHECK_LAUNCH("k_bstep16");
    return DMPNN_OK;
}

}  // namespace bstep16

// ---- host side ---------------------------------------------------------------------------------------------------------
static unsigned qmagic_of(int64_t N) {
    const unsigned qn = (unsigned)(N / 4);
    return qn > 1 ? (unsigned)(((1ull << 32) + qn - 1) / qn) : 0u;
}

// One backward step launch.  site: whose sign bits / which gZ; Tin null = gather mode (gH = gMv[dst]); W null = last site.
int launch_bstep16(const dmpnn_fwd_args& f, int site, const float* Tin, const float* gMv, const SplitWView* W, float* Tout,
                   unsigned char* Zrows, hipStream_t s) {
    const int64_t nV = f.n_atoms, nE = f.n_edges, h = f.d_h;
    const PlanLayout L = plan_layout(nV, nE);
    const int* plan_i = static_cast<const int*>(f.plan);
    bstep16::BStepK g;
    memset(&g, 0, sizeof(g));
    g.M = (int)nE; g.N = (int)h; g.n_atoms = (int)nV;
    g.hdr = plan_i;
    g.tile_row = plan_i + L.tile_row; g.tile_atom = plan_i + L.tile_atom; g.row_ptr = plan_i + L.row_ptr;
    g.revp = plan_i + L.revp; g.dstp = plan_i + L.dstp;
    g.Tin = Tin; g.ldt = (int)f.ldh; g.gMv = gMv; g.ldg = (int)f.ldh;
    const int bn = step16::block_cols((int)h);
    g.bstride = bn / 8;
    g.bits = f.act == DMPNN_ACT_NONE ? nullptr : static_cast<const unsigned char*>(f.keep_bits) + (size_t)site * (size_t)nE * (size_t)g.bstride;
    g.neg = f.act == DMPNN_ACT_RELU ? 0.f : (f.act == DMPNN_ACT_LEAKYRELU ? f.act_slope : 1.f);
    g.Zrows = Zrows; g.tsz = step16::split_row_bytes((int)h);   // (split rows [n_edges][split_row_bytes(d_h)] for k_wgrad16r)
    if (W) { g.W.p = W->p; g.W.inv_scale = W->inv_scale; g.W.nc = W->nc; }
    g.Tout = W ? Tout : nullptr; g.ldo = (int)f.ldh;
    g.qmagic = qmagic_of(h);
    g.poison_mask = kPlanNoFuse;
    const int n_tiles = (int)L.max_tiles;
    switch (bn / 64) {
        case 1: return bstep16::launch_bstep<1>(g, n_tiles, s);
        case 2: return bstep16::launch_bstep<2>(g, n_tiles, s);
        case 3: return bstep16::launch_bstep<3>(g, n_tiles, s);
        case 4: return bstep16::launch_bstep<4>(g, n_tiles, s);
        default: return bstep16::launch_bstep<5>(g, n_tiles, s);
    }
}

}  // namespace dmpnn
