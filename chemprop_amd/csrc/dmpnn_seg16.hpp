// Segment epilogue of the per-step fused route on the f16 pipe (see dmpnn_step16_impl.hpp): from the fp32 LDS tile of
// tau(z) rows to the tile's per-atom sums and the next message in split rows.  Shared by k_rows16<.., SEG> (K1) and
// k_step16 (the depth updates).
#pragma once

#include "dmpnn_mega16_impl.hpp"

namespace dmpnn {
namespace step16 {

using gemm::kAtomCache;
using mega16::h4;
using mega16::split4;

constexpr int BM = 48;

struct SegOut {
    const int* row_ptr; const int* revp;
    unsigned char* Mout; int ts;   // next message, split rows [E][ts] (or null)
    float* Sout; int lds;          // per-atom sums [V][lds] fp32 (or null)
    int N;                         // live columns (the padded row holds NQP column quads, zero-filled beyond N)
    int half;                      // 1: half storage — rows of [hi 32 halfs] chunks only (DMPNN_F_STORE16), else [hi | lo] chunk pairs
    float* M32; int ldm32;         // training: the message ALSO as fp32 rows [E][ldm32] (what the weight gradients read), or null
    unsigned char* SoutS; int tss; // per-atom sums as SPLIT rows [V][tss] (exact hi | lo, each row's scale in its tail) instead of
                                   // fp32 Sout: what the finalize contraction on the step kernel consumes (or null)
};

// ---- segment epilogue shared by the K1 kernel (k_rows16<.., SEG>) and the update kernel -----------------------------
// T[BM][LDC]: y = tau(z) of the tile's rows (fp32, LDS; overwritten by the message).  meta: [BM] global row of the reverse
// edge | [kAtomCache + 1] tile-local row pointers.  Pass 1: S per (atom, column quad), Sout, the message S - y in place
// and its maximum; tile scale; pass 2 (row-major): the message rows in split form to Mout[rev r].  Sums run over the rows of an atom in increasing row order = increasing
// edge id (the reference's sequential scatter order, base.py:144-146).
template <int LDC, int NQP, int NT, class TileScale>
__device__ __forceinline__ void seg_epilogue(const SegOut& o, float* T, int* meta, int rs, int nrows, int va, int vb,
                                             int seg_rp_reg, bool poison, unsigned qmagic, TileScale&& tile_scale) {
    const int tid = threadIdx.x;
    int* rp = meta + BM;
    const int qn = o.N >> 2;
    const float nanv = __int_as_float(0x7fc00000);
    float mx = 0.f;
    // pass 1: one (atom, column quad) item per thread and step: S, then the message of the atom's rows IN PLACE
    // (T[r] <- S - y[r]: the cells of an item are its own)
    for (int a0 = va; a0 < vb; a0 += kAtomCache) {
        const int na = vb - a0 < kAtomCache ? vb - a0 : kAtomCache;
        __syncthreads();
        if (tid <= na) rp[tid] = a0 == va ? seg_rp_reg : o.row_ptr[a0 + tid] - rs;
        __syncthreads();
        const int n_items = na * qn;
        for (int it = tid; it < n_items; it += NT) {
            const int al = qn == 1 ? it : (int)__umulhi((unsigned)it, qmagic);
            const int q = it - al * qn;
            const int r0 = rp[al], r1 = rp[al + 1];
            float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r1 - r0 <= 4) {
                // an atom of a molecule has <= 4 bonds (the common case by far): its rows are requested TOGETHER — a loop with a
                // data-dependent trip count waits one LDS latency per row, twice — summed in the same increasing row order, and
                // the messages go back without being read again
                float4 y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = *reinterpret_cast<const float4*>(T + (r0 + i < r1 ? r0 + i : (r1 > r0 ? r0 : 0)) * LDC + 4 * q);
                if (r1 > r0) S = y[0];
#pragma unroll
                for (int i = 1; i < 4; ++i)
                    if (r0 + i < r1) { S.x += y[i].x; S.y += y[i].y; S.z += y[i].z; S.w += y[i].w; }
                if (o.Sout) *reinterpret_cast<float4*>(o.Sout + (long long)(a0 + al) * o.lds + 4 * q) = poison ? make_float4(nanv, nanv, nanv, nanv) : S;
                if (o.SoutS) {  // kept in the atom's first row for the split pass below (an atom without rows: zeros, nothing to keep)
                    if (r1 > r0) *reinterpret_cast<float4*>(T + r0 * LDC + 4 * q) = S;
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(S.x), fabsf(S.y)), fmaxf(fabsf(S.z), fabsf(S.w))));
                }
                if (o.Mout) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (r0 + i < r1) {
                            const float4 m = make_float4(S.x - y[i].x, S.y - y[i].y, S.z - y[i].z, S.w - y[i].w);
                            *reinterpret_cast<float4*>(T + (r0 + i) * LDC + 4 * q) = m;
                            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(m.x), fabsf(m.y)), fmaxf(fabsf(m.z), fabsf(m.w))));
                        }
                }
                continue;
            }
            for (int r = r0; r < r1; ++r) {
                const float4 y = *reinterpret_cast<const float4*>(T + r * LDC + 4 * q);
                if (r == r0) S = y;
                else { S.x += y.x; S.y += y.y; S.z += y.z; S.w += y.w; }
            }
            if (o.Sout) *reinterpret_cast<float4*>(o.Sout + (long long)(a0 + al) * o.lds + 4 * q) = poison ? make_float4(nanv, nanv, nanv, nanv) : S;
            if (o.SoutS) {
                if (r1 > r0) *reinterpret_cast<float4*>(T + r0 * LDC + 4 * q) = S;
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(S.x), fabsf(S.y)), fmaxf(fabsf(S.z), fabsf(S.w))));
            }
            if (o.Mout) {
                for (int r = r0; r < r1; ++r) {
                    float4* cell = reinterpret_cast<float4*>(T + r * LDC + 4 * q);
                    const float4 y = *cell;
                    const float4 m = make_float4(S.x - y.x, S.y - y.y, S.z - y.z, S.w - y.w);
                    *cell = m;
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(m.x), fabsf(m.y)), fmaxf(fabsf(m.z), fabsf(m.w))));
                }
            }
        }
        if (o.SoutS) {
            // the sums of this chunk of atoms as split rows (their own power-of-two scale, in the rows' tails): row-major,
            // item = (atom, 8 columns); an atom without rows writes zeros
            const float sS = poison ? 1.f : tile_scale(mx);  // (uniform; contains a barrier: every sum of the chunk is in the tile)
            mx = 0.f;
            constexpr int G8S = NQP / 2;
            typedef _Float16 h8s __attribute__((ext_vector_type(8)));
            for (int it = tid; it < na * G8S; it += NT) {
                const int al = it / G8S, g8 = it - al * G8S;
                const int r0 = rp[al], r1 = rp[al + 1];
                float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
                if (r1 > r0) {
                    m0 = *reinterpret_cast<const float4*>(T + r0 * LDC + 8 * g8);
                    m1 = *reinterpret_cast<const float4*>(T + r0 * LDC + 8 * g8 + 4);
                }
                if (poison) { m0 = make_float4(nanv, nanv, nanv, nanv); m1 = m0; }
                h4 h0, l0, h1, l1;
                split4(m0, sS, h0, l0);
                split4(m1, sS, h1, l1);
                unsigned char* p = o.SoutS + (long long)(a0 + al) * o.tss + (g8 >> 2) * 128 + (g8 & 3) * 16;
                *reinterpret_cast<h8s*>(p) = h8s{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                *reinterpret_cast<h8s*>(p + 64) = h8s{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                if (g8 == 0) *reinterpret_cast<float4*>(o.SoutS + (long long)(a0 + al) * o.tss + (NQP >> 3) * 128) = make_float4(sS, 0.f, 0.f, 0.f);
            }
        }
    }
    if (!o.Mout) return;
    const float s = poison ? 1.f : tile_scale(mx);  // (uniform; contains a barrier: the message tile is complete)
    // pass 2: row-major — item = (row, 8 columns): 16 B of hi + 16 B of lo; four lanes fill one 128-byte chunk, a wave
    // writes whole rows of the split tensor (row rev r) contiguously.  Columns beyond N are zero in T (zero weights,
    // zero residual, tau(0) = 0 for every built-in activation), so the padded chunks come out as zeros.
    constexpr int G8 = NQP / 2;  // 8-column groups of a padded row
    for (int it = tid; it < nrows * G8; it += NT) {
        const int r = it / G8, g8 = it - r * G8;
        float4 m0 = *reinterpret_cast<const float4*>(T + r * LDC + 8 * g8);
        float4 m1 = *reinterpret_cast<const float4*>(T + r * LDC + 8 * g8 + 4);
        if (poison) { m0 = make_float4(nanv, nanv, nanv, nanv); m1 = m0; }
        if (o.M32 && 8 * g8 < o.N) {  // (columns beyond N: padding of the split row only)
            float* q = o.M32 + (long long)meta[r] * o.ldm32 + 8 * g8;
            *reinterpret_cast<float4*>(q) = m0;
            if (8 * g8 + 4 < o.N) *reinterpret_cast<float4*>(q + 4) = m1;
        }
        h4 h0, l0, h1, l1;
        split4(m0, s, h0, l0);
        split4(m1, s, h1, l1);
        typedef _Float16 h8v __attribute__((ext_vector_type(8)));
        if (o.half) {  // (uniform) half storage: the rounded hi part alone — 2 bytes per element, 11-bit significands
            unsigned char* p = o.Mout + (long long)meta[r] * o.ts + (g8 >> 2) * 64 + (g8 & 3) * 16;
            *reinterpret_cast<h8v*>(p) = h8v{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            if (g8 == 0) *reinterpret_cast<float4*>(o.Mout + (long long)meta[r] * o.ts + (NQP >> 3) * 64) = make_float4(s, 0.f, 0.f, 0.f);
            continue;
        }
        unsigned char* p = o.Mout + (long long)meta[r] * o.ts + (g8 >> 2) * 128 + (g8 & 3) * 16;
        *reinterpret_cast<h8v*>(p) = h8v{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        *reinterpret_cast<h8v*>(p + 64) = h8v{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        if (g8 == 0) *reinterpret_cast<float4*>(o.Mout + (long long)meta[r] * o.ts + (NQP >> 3) * 128) = make_float4(s, 0.f, 0.f, 0.f);
    }
}

}  // namespace step16
}  // namespace dmpnn
