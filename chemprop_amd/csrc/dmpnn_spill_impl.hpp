// The whole forward / data-gradient chain of BondMessagePassing for ONE piece (molecule) of ANY size, by one
// workgroup, in plain fp32 — the path a piece tile takes inside the whole-forward tile kernels when it does not fit
// their matrix-pipe layout (more than kMegaBM directed edges or kMegaBA atoms).
//
// The reference has no limit on the size of a molecule (chemprop/data/collate.py:48-56, nn/message_passing/
// base.py:196-212).  The tile kernels do: their tiles are made for QM9-sized molecules.  A larger one used to poison the
// whole batch with NaN; now the planners hand it over as a tile of its own and the tile kernel that meets it runs this
// code instead: correct for every size, at the speed of an fp32 vector loop (a 60-edge molecule: ~100 us) — an
// outlier path, datasets of large molecules take the per-step routes (the host switches on the plan's spill count).
//
// Arithmetic: every contraction is an ascending-k fp32 fmaf chain per output element; the segment sums add the rows of
// an atom in increasing row order (the reference's sequential scatter order); every thread owns output columns
// (c = tid, tid + 256, ...) of every row, so the message / aggregate steps need no synchronisation and the only
// barriers are around the LDS staging of a contraction's operand rows.
//
// Memory: the edge / atom tensors live in global memory — the forward's kept tensors when the backward pass will
// follow, else the `spill_ws` scratch of dmpnn_fwd_args ([3][n_edges][ldh] + [n_atoms][ldh] floats, indexed by the
// batch's own row ids, so no allocation and no cursor on the device).
#pragma once

#include "dmpnn_common.hpp"

// (kernel experiments: -DDMPNN_SPILL_NOINLINE keeps the generic path out of line, -DDMPNN_NO_SPILL compiles it away)
#if defined(DMPNN_SPILL_NOINLINE)
#define DMPNN_SPILL_FN __device__ __noinline__
#else
#define DMPNN_SPILL_FN __device__ __forceinline__
#endif

namespace dmpnn {
namespace spill {

// The kernel's own argument block, read AFRESH from the kernarg segment through a pointer the compiler cannot see
// through: the generic path then keeps nothing of the hot path's scalar registers alive (with the by-value argument
// struct shared by both paths hipcc spilled ~300 SGPRs around the hot path's prologue: +2.4 us per launch, measured).
template <class KArgs>
__device__ __forceinline__ const KArgs* fresh_kernargs() {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long a = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();  // (constant address space: same 64-bit address)
    asm volatile("" : "+s"(a));
    return reinterpret_cast<const KArgs*>(a);
#else
    return nullptr;
#endif
}

constexpr int kRB = 16;          // operand rows staged per block
constexpr int kSpillThreads = 256;

// floats of LDS the staging of a K-wide operand needs
__host__ __device__ constexpr int xs_floats(int K) { return kRB * (((K + 3) & ~3) + 4); }

// emit(r, c, bias[c] + sum_k X(r, k) W(c, k))  for r < R, c < N;  X(r, k) = stage(r, k),  W(c, k) = Wp[c wrs + k wks]
template <class Stage, class Emit>
__device__ __forceinline__ void contract(float* xs, int R, int N, int K, const float* __restrict__ Wp, long long wrs, long long wks,
                                         const float* __restrict__ bias, Stage&& stage, Emit&& emit) {
    const int tid = threadIdx.x;
    const int K4 = (K + 3) & ~3, KP = K4 + 4;
    for (int r0 = 0; r0 < R; r0 += kRB) {
        __syncthreads();  // the previous block's readers are done; what other threads wrote to global memory is visible
        for (int j = 0; j < kRB; ++j)
            for (int k = tid; k < K4; k += kSpillThreads) xs[j * KP + k] = (r0 + j < R && k < K) ? stage(r0 + j, k) : 0.f;
        __syncthreads();
        for (int c = tid; c < N; c += kSpillThreads) {
            float acc[kRB];
            const float b = bias ? bias[c] : 0.f;
#pragma unroll
            for (int j = 0; j < kRB; ++j) acc[j] = b;
            const float* w = Wp + (long long)c * wrs;
            for (int k = 0; k < K4; k += 4) {
                const float w0 = w[(long long)k * wks];
                const float w1 = k + 1 < K ? w[(long long)(k + 1) * wks] : 0.f;
                const float w2 = k + 2 < K ? w[(long long)(k + 2) * wks] : 0.f;
                const float w3 = k + 3 < K ? w[(long long)(k + 3) * wks] : 0.f;
#pragma unroll
                for (int j = 0; j < kRB; ++j) {
                    const float4 x = *reinterpret_cast<const float4*>(xs + j * KP + k);
                    acc[j] = fmaf(x.w, w3, fmaf(x.z, w2, fmaf(x.y, w1, fmaf(x.x, w0, acc[j]))));
                }
            }
#pragma unroll
            for (int j = 0; j < kRB; ++j)
                if (r0 + j < R) emit(r0 + j, c, acc[j]);
        }
    }
    __syncthreads();
}

struct FwdView {
    bool lean;                      // rows = the caller's edges rs .. (tile plan); else CSR rows of a full / light plan
    int rs, nrows, va, na, nV, nE;
    const int* row_ptr; const int* srcp; const int* perm; const int* revp;  // CSR-row plan
    const long long* edge_index; const long long* rev64;                     // the caller's arrays (lean)
    const float* V; int ldv; const float* E; int lde; int d_v, d_e, h, depth;
    const float* W_i; const float* b_i; const float* W_h; const float* b_h; const float* W_o; const float* b_o;
    int act; float slope;
    float* out; int ldout;
    float* H0; float* Hs; int n_hslots; float* Ms; int n_mslots; float* Mv; int ldh; long long slot;
};

// base.py:196-212 for the piece rows [rs, rs + nrows) / atoms [va, va + na)
DMPNN_SPILL_FN void forward(const FwdView& g, float* xs) {
#if defined(DMPNN_NO_SPILL)
    return;
#endif
    const int tid = threadIdx.x, N = g.h, T = g.depth;
    const long long rs = g.rs, va = g.va;
    auto src_at = [&](int r) -> long long { return g.lean ? g.edge_index[rs + r] : (long long)g.srcp[rs + r]; };
    auto rev_of = [&](int r) -> int { return (int)((g.lean ? g.rev64[rs + r] : (long long)g.revp[rs + r]) - rs); };
    auto e_of = [&](int r) -> long long { return g.lean ? rs + r : (long long)g.perm[rs + r]; };
    auto nan_out = [&]() {
        const float nanv = __int_as_float(0x7fc00000);
        for (long long i = tid; i < (long long)g.na * N; i += kSpillThreads) g.out[(va + i / N) * g.ldout + (i % N)] = nanv;
    };
    if (g.lean) {  // the tile kernel's closure check, for every row of the piece
        int bad = 0;
        for (int r = tid; r < g.nrows; r += kSpillThreads) {
            const long long s = g.edge_index[rs + r] - va, d = g.edge_index[(long long)g.nE + rs + r] - va, rv = g.rev64[rs + r] - rs;
            bad |= (s < 0 || s >= g.na || d < 0 || d >= g.na || rv < 0 || rv >= g.nrows) ? 1 : 0;
        }
        int* flag = reinterpret_cast<int*>(xs);  // (a word of the staging area: no static LDS in the tile kernels)
        if (tid == 0) *flag = 0;
        __syncthreads();
        if (bad) atomicOr(flag, 1);
        __syncthreads();
        const int any_bad = *flag;
        __syncthreads();
        if (any_bad) { nan_out(); return; }
    }
    if (!g.H0 || !g.Mv || (T > 1 && (!g.Hs || !g.Ms))) { nan_out(); return; }  // (no workspace: never silently wrong)
    // K1  H0 = W_i [V[src] || E] (+ b_i)                               mixins.py:8-9
    contract(xs, g.nrows, N, g.d_v + g.d_e, g.W_i, g.d_v + g.d_e, 1, g.b_i,
             [&](int r, int k) -> float {
                 return k < g.d_v ? g.V[src_at(r) * g.ldv + k] : g.E[e_of(r) * g.lde + (k - g.d_v)];
             },
             [&](int r, int c, float z) { g.H0[(rs + r) * g.ldh + c] = z; });
    auto Hprev = [&](int t_prev, int r, int c) -> float {  // H^(t_prev)[r][c]; H^(0) = tau(H0) (base.py:200)
        if (t_prev == 0) return apply_act(g.H0[(rs + r) * g.ldh + c], g.act, g.slope);
        return g.Hs[(long long)((t_prev - 1) % g.n_hslots) * g.slot + (rs + r) * g.ldh + c];
    };
    // S[a] = sum of H over the rows entering a, in increasing row order -> the Mv rows of the piece
    auto segsum = [&](int t_prev) {
        for (int c = tid; c < N; c += kSpillThreads) {
            if (g.lean) {
                for (int a = 0; a < g.na; ++a) g.Mv[(va + a) * g.ldh + c] = 0.f;
                for (int r = 0; r < g.nrows; ++r) {
                    const long long a = g.edge_index[(long long)g.nE + rs + r];
                    g.Mv[a * g.ldh + c] += Hprev(t_prev, r, c);
                }
            } else {
                for (int a = 0; a < g.na; ++a) {
                    float s = 0.f;
                    for (int r = g.row_ptr[va + a] - g.rs; r < g.row_ptr[va + a + 1] - g.rs; ++r) s += Hprev(t_prev, r, c);
                    g.Mv[(va + a) * g.ldh + c] = s;
                }
            }
        }
    };
    for (int t = 1; t < T; ++t) {
        // M = S[src] - H[rev]                                          mixins.py:11-18
        segsum(t - 1);
        float* Mt = g.Ms + (long long)((t - 1) % g.n_mslots) * g.slot;
        for (int c = tid; c < N; c += kSpillThreads)
            for (int r = 0; r < g.nrows; ++r) Mt[(rs + r) * g.ldh + c] = g.Mv[src_at(r) * g.ldh + c] - Hprev(t - 1, rev_of(r), c);
        // H = tau(H0 + W_h M (+ b_h))                                  base.py:135-141
        float* Ht = g.Hs + (long long)((t - 1) % g.n_hslots) * g.slot;
        contract(xs, g.nrows, N, N, g.W_h, N, 1, g.b_h,
                 [&](int r, int k) -> float { return Mt[(rs + r) * g.ldh + k]; },
                 [&](int r, int c, float z) { Ht[(rs + r) * g.ldh + c] = apply_act(z + g.H0[(rs + r) * g.ldh + c], g.act, g.slope); });
    }
    segsum(T - 1);  // base.py:208-211
    // finalize: tau(W_o [V || Mv] + b_o)                               base.py:180-194
    contract(xs, g.na, N, g.d_v + N, g.W_o, g.d_v + N, 1, g.b_o,
             [&](int a, int k) -> float { return k < g.d_v ? g.V[(va + a) * g.ldv + k] : g.Mv[(va + a) * g.ldh + (k - g.d_v)]; },
             [&](int a, int c, float z) { g.out[(va + a) * g.ldout + c] = apply_act(z, g.act, g.slope); });
}

struct BwdView {
    int rs, nrows, va, na, h, depth, d_v;
    const int* row_ptr; const int* srcp; const int* revp;  // CSR-row plan (the forward kept its tensors in row order)
    bool lean; const long long* edge_index; const long long* rev64; int nE;  // tile plan: rows = the caller's edges, its own arrays
    int act; float slope;
    const float* gHO; int ldg; const float* HO; int ldho;
    const float* H0; const float* Hs; int ldh; long long slot;
    float* gZO; float* gZs; float* gH0;
    const float* W_o; const float* W_h;     // nn.Linear layout: W_o [h, d_v + h], W_h [h, h]
    float* gM; float* Ta;                   // scratch: [n_edges][ldh], [n_atoms][ldh]
    const float* g_edge; int ld_ge;         // dL/dH^(depth-1) of a second consumer of the edge states (dmpnn_bwd_args.g_edge) or NULL
};

// the data-gradient chain of the backward pass (what k_mpnn_tile16_bwd does for a tile) for one piece
DMPNN_SPILL_FN void backward(const BwdView& g, float* xs) {
#if defined(DMPNN_NO_SPILL)
    return;
#endif
    const int tid = threadIdx.x, N = g.h, T = g.depth;
    const long long rs = g.rs, va = g.va;
    auto dact = [&](float gv, float y, bool preact) -> float {  // g * tau'(.) from the output (or the pre-activation)
        if (g.act == DMPNN_ACT_NONE) return gv;
        if (g.act == DMPNN_ACT_RELU) return y > 0.f ? gv : 0.f;
        if (preact) y = apply_act(y, g.act, g.slope);
        return gv * act_grad_from_out(y, g.act, g.slope);
    };
    // gZO = gHO * tau'(HO)
    for (int c = tid; c < N; c += kSpillThreads)
        for (int a = 0; a < g.na; ++a) g.gZO[(va + a) * g.ldh + c] = dact(g.gHO[(va + a) * g.ldg + c], g.HO[(va + a) * g.ldho + c], false);
    // gMv = gZO . W_o[:, d_v:]   ->  Ta
    contract(xs, g.na, N, N, g.W_o + g.d_v, 1, g.d_v + N, nullptr,
             [&](int a, int k) -> float { return g.gZO[(va + a) * g.ldh + k]; },
             [&](int a, int c, float z) { g.Ta[(va + a) * g.ldh + c] = z; });
    // the graph of the piece in row coordinates: CSR tables, or (tile plan) the caller's arrays.  Every loop below walks the rows in
    // increasing order and looks up the row's atoms — the same sums in the same order either way, except the per-atom totals of the
    // message backward, whose addends come in row order (CSR rows are sorted by destination, the caller's are not)
    auto dst_of = [&](int r) -> long long { return g.edge_index[(long long)g.nE + rs + r]; };   // (lean)
    auto src_of = [&](int r) -> long long { return g.lean ? g.edge_index[rs + r] : (long long)g.srcp[rs + r]; };
    auto rev_of = [&](int r) -> long long { return g.lean ? g.rev64[rs + r] : (long long)g.revp[rs + r]; };
    if (g.lean) {  // closure of the piece, as the forward's generic path checks it: a piece that is not closed gets NaN gradients
        int bad = 0;
        for (int r = tid; r < g.nrows; r += kSpillThreads) {
            const long long s_ = src_of(r) - va, d_ = dst_of(r) - va, rv = rev_of(r) - rs;
            bad |= (s_ < 0 || s_ >= g.na || d_ < 0 || d_ >= g.na || rv < 0 || rv >= g.nrows) ? 1 : 0;
        }
        int* flag = reinterpret_cast<int*>(xs);
        if (tid == 0) *flag = 0;
        __syncthreads();
        if (bad) atomicOr(flag, 1);
        __syncthreads();
        const int any_bad = *flag;
        __syncthreads();
        if (any_bad) {
            const float nanv = __int_as_float(0x7fc00000);
            for (int c = tid; c < N; c += kSpillThreads) {
                for (int r = 0; r < g.nrows; ++r) {
                    g.gH0[(rs + r) * g.ldh + c] = nanv;
                    for (int t = 0; t < T - 1; ++t) g.gZs[(long long)t * g.slot + (rs + r) * g.ldh + c] = nanv;
                }
                for (int a = 0; a < g.na; ++a) g.gZO[(va + a) * g.ldh + c] = nanv;
            }
            return;
        }
    }
    // fn(a, r) for every row r of the piece with its destination atom a (piece-local), rows in increasing order per atom
    auto for_rows = [&](auto&& fn) {
        if (g.lean) {
            for (int r = 0; r < g.nrows; ++r) fn((int)(dst_of(r) - va), r);
        } else {
            for (int a = 0; a < g.na; ++a)
                for (int r = g.row_ptr[va + a] - g.rs; r < g.row_ptr[va + a + 1] - g.rs; ++r) fn(a, r);
        }
    };
    // gH^(T-1)[r] = gMv[dst r] (+ the second gradient input)             (aggregation backward)
    auto ge = [&](int r, int c) -> float { return g.g_edge ? g.g_edge[(rs + r) * g.ld_ge + c] : 0.f; };
    if (T == 1) {
        for (int c = tid; c < N; c += kSpillThreads)
            for_rows([&](int a, int r) { g.gH0[(rs + r) * g.ldh + c] = dact(g.Ta[(va + a) * g.ldh + c] + ge(r, c), g.H0[(rs + r) * g.ldh + c], true); });
        return;
    }
    for (int c = tid; c < N; c += kSpillThreads)
        for_rows([&](int a, int r) {
            const float gz = dact(g.Ta[(va + a) * g.ldh + c] + ge(r, c), g.Hs[(long long)(T - 2) * g.slot + (rs + r) * g.ldh + c], false);
            g.gZs[(long long)(T - 2) * g.slot + (rs + r) * g.ldh + c] = gz;
            g.gH0[(rs + r) * g.ldh + c] = gz;
        });
    for (int t = T - 1; t >= 1; --t) {
        const float* gZt = g.gZs + (long long)(t - 1) * g.slot;
        // gM = gZ^(t) . W_h
        contract(xs, g.nrows, N, N, g.W_h, 1, N, nullptr,
                 [&](int r, int k) -> float { return gZt[(rs + r) * g.ldh + k]; },
                 [&](int r, int c, float z) { g.gM[(rs + r) * g.ldh + c] = z; });
        // gH^(t-1)[r'] = sum_{r: src r = dst r'} gM[r] - gM[rev r']     (message backward)
        for (int c = tid; c < N; c += kSpillThreads) {
            for (int a = 0; a < g.na; ++a) g.Ta[(va + a) * g.ldh + c] = 0.f;
            for (int r = 0; r < g.nrows; ++r) g.Ta[src_of(r) * g.ldh + c] += g.gM[(rs + r) * g.ldh + c];
            for_rows([&](int a, int r) {
                const float gh = g.Ta[(va + a) * g.ldh + c] - g.gM[rev_of(r) * g.ldh + c];
                if (t - 1 >= 1) {
                    const float gz = dact(gh, g.Hs[(long long)(t - 2) * g.slot + (rs + r) * g.ldh + c], false);
                    g.gZs[(long long)(t - 2) * g.slot + (rs + r) * g.ldh + c] = gz;
                    g.gH0[(rs + r) * g.ldh + c] += gz;
                } else {
                    g.gH0[(rs + r) * g.ldh + c] += dact(gh, g.H0[(rs + r) * g.ldh + c], true);  // through H^(0) = tau(H0)
                }
            });
        }
    }
}

}  // namespace spill
}  // namespace dmpnn
