// K2 / K4 — the gather / segment-sum / reverse-subtract step (HBM-roofline kernels).
//
// Reference (chemprop/nn/message_passing/mixins.py:11-18):
//     index = dst.unsqueeze(1).repeat(1, h)                       # [E,h] int64, rebuilt every call
//     M_all = zeros(V,h).scatter_reduce_(0, index, H, "sum")[src] # [V,h] scatter, then [E,h] gather
//     M     = M_all - H[rev]                                      # another [E,h] gather + sub
// and base.py:208-211 for the final per-atom aggregation.
//
// MI355X form.  One 64-lane wavefront owns one ATOM v.  Lanes span the hidden dimension
// (16 B / lane, a 1200-byte H row is 75 float4 -> fully coalesced row reads).  The wave reads the
// d incoming rows H[e'_1..e'_d] (CSR order = increasing edge id = the reference's summation order),
// keeps them in registers, forms S = ((r1 + r2) + ...) and — because for a molecular graph the
// out-edges of v are exactly the reverses of its in-edges — writes
//     M[rev(e'_i)] = S - r_i
// i.e. every H row is read ONCE and every M row written ONCE: 2*E*h*4 B + indices, the
// algorithmic minimum, with no atomics, no [V,h] temporary and no LDS round trip.
// Graphs that violate the symmetry invariants (plan flag PLAN_ASYMMETRIC, decided on device, no
// host sync) take the literal edge form  M[e] = S[src(e)] - H[rev(e)]  in the same launch.
//
// tau-on-load: the first depth step consumes tau(H0) (base.py:200); applying tau while loading
// saves materialising H^(0).  Undirected (base.py:202-203) averages each row with its reverse
// while loading.
#include "dmpnn_common.hpp"

namespace dmpnn {

namespace {

constexpr int kWavesPerBlock = 4;

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    using T = float4;
    static __device__ __forceinline__ T load(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void store(float* p, T v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ T add(T a, T b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
    static __device__ __forceinline__ T sub(T a, T b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
    static __device__ __forceinline__ T half_sum(T a, T b) {
        return make_float4((a.x + b.x) / 2.f, (a.y + b.y) / 2.f, (a.z + b.z) / 2.f, (a.w + b.w) / 2.f);
    }
    static __device__ __forceinline__ T act(T a, int act, float slope) { return apply_act4(a, act, slope); }
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <>
struct Vec<1> {
    using T = float;
    static __device__ __forceinline__ T load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, T v) { *p = v; }
    static __device__ __forceinline__ T add(T a, T b) { return a + b; }
    static __device__ __forceinline__ T sub(T a, T b) { return a - b; }
    static __device__ __forceinline__ T half_sum(T a, T b) { return (a + b) / 2.f; }
    static __device__ __forceinline__ T act(T a, int act, float slope) { return apply_act(a, act, slope); }
    static __device__ __forceinline__ T zero() { return 0.f; }
};

struct SegArgs {
    PlanView pv;
    int nV, nE, h;
    const float* Hin;
    int64_t ld_in;
    float* out;  // M [E, ld_out] (message) or Mv [V, ld_out] (aggregate)
    int64_t ld_out;
    int act;
    float slope;
    const float* slope_ptr;
    int undirected;
};

// Load row e (element offset c) of the message input: tau-on-load and undirected averaging.
// UNDIR and ACT are template parameters (-1 = decided at run time, the slow generic build) so that
// in the hot instantiations no load sits under a runtime branch and no activation switch separates
// a load from its use (hipcc would wait vmcnt(0) per load, serialising the row reads).
template <int ACT>
__device__ __forceinline__ float act1(float z, int act_rt, float slope) {
    if (ACT == DMPNN_ACT_NONE) return z;
    if (ACT == DMPNN_ACT_RELU) return z < 0.f ? 0.f : z;
    return apply_act(z, act_rt, slope);
}
template <int VEC, int ACT>
__device__ __forceinline__ typename Vec<VEC>::T actv(typename Vec<VEC>::T r, int act_rt, float slope);
template <>
__device__ __forceinline__ float actv<1, DMPNN_ACT_NONE>(float r, int, float) { return r; }
template <>
__device__ __forceinline__ float actv<1, DMPNN_ACT_RELU>(float r, int a, float s) { return act1<DMPNN_ACT_RELU>(r, a, s); }
template <>
__device__ __forceinline__ float actv<1, -1>(float r, int a, float s) { return apply_act(r, a, s); }
template <>
__device__ __forceinline__ float4 actv<4, DMPNN_ACT_NONE>(float4 r, int, float) { return r; }
template <>
__device__ __forceinline__ float4 actv<4, DMPNN_ACT_RELU>(float4 r, int, float) {
    return make_float4(r.x < 0.f ? 0.f : r.x, r.y < 0.f ? 0.f : r.y, r.z < 0.f ? 0.f : r.z, r.w < 0.f ? 0.f : r.w);
}
template <>
__device__ __forceinline__ float4 actv<4, -1>(float4 r, int a, float s) { return apply_act4(r, a, s); }

template <int VEC, int UNDIR, int ACT>
__device__ __forceinline__ typename Vec<VEC>::T load_row(const SegArgs& a, int e, int er, int c, float slope) {
    using V = Vec<VEC>;
    typename V::T r = V::load(a.Hin + (int64_t)e * a.ld_in + c);
    if (UNDIR == 1 || (UNDIR == -1 && a.undirected)) {
        typename V::T q = V::load(a.Hin + (int64_t)er * a.ld_in + c);
        r = V::half_sum(actv<VEC, ACT>(r, a.act, slope), actv<VEC, ACT>(q, a.act, slope));
    } else {
        r = actv<VEC, ACT>(r, a.act, slope);
    }
    return r;
}

// One atom with in-degree exactly D (all D row loads issued back to back, unconditionally), two
// column groups per pass (lane and lane+64: a 300-float row is 75 float4).
template <int VEC, int MODE, int UNDIR, int ACT, int D>
__device__ __forceinline__ void atom_body(const SegArgs& a, int v, int beg, int lane, int n_cols, float slope) {
    using V = Vec<VEC>;
    using T = typename V::T;
    int eid[D], erev[D];
#pragma unroll
    for (int i = 0; i < D; ++i) eid[i] = a.pv.perm[beg + i];
#pragma unroll
    for (int i = 0; i < D; ++i) erev[i] = (MODE == 0 || UNDIR != 0) ? a.pv.rev[eid[i]] : 0;
    for (int cg0 = 0; cg0 < n_cols; cg0 += 128) {
        const int cgA = cg0 + lane, cgB = cg0 + 64 + lane;
        const bool okA = cgA < n_cols, okB = cgB < n_cols;
        const int cA = (okA ? cgA : 0) * VEC, cB = (okB ? cgB : 0) * VEC;
        T rA[D], rB[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            rA[i] = load_row<VEC, UNDIR, ACT>(a, eid[i], erev[i], cA, slope);
            rB[i] = load_row<VEC, UNDIR, ACT>(a, eid[i], erev[i], cB, slope);
        }
        T SA = rA[0], SB = rB[0];
#pragma unroll
        for (int i = 1; i < D; ++i) {
            SA = V::add(SA, rA[i]);
            SB = V::add(SB, rB[i]);
        }
        if (MODE == 1) {
            if (okA) V::store(a.out + (int64_t)v * a.ld_out + cA, SA);
            if (okB) V::store(a.out + (int64_t)v * a.ld_out + cB, SB);
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                float* o = a.out + (int64_t)erev[i] * a.ld_out;
                if (okA) V::store(o + cA, V::sub(SA, rA[i]));
                if (okB) V::store(o + cB, V::sub(SB, rB[i]));
            }
        }
    }
}

// Any in-degree (used for d > 6): running sum, rows re-read for the write-back (they hit L1/L2).
template <int VEC, int MODE, int UNDIR, int ACT>
__device__ __forceinline__ void atom_body_any(const SegArgs& a, int v, int beg, int d, int lane, int n_cols, float slope) {
    using V = Vec<VEC>;
    using T = typename V::T;
    for (int cg = lane; cg < n_cols; cg += 64) {
        const int c = cg * VEC;
        T S = V::zero();
        for (int i = 0; i < d; ++i) {
            const int e = a.pv.perm[beg + i];
            const T r = load_row<VEC, UNDIR, ACT>(a, e, a.pv.rev[e], c, slope);
            S = (i == 0) ? r : V::add(S, r);
        }
        if (MODE == 1) {
            V::store(a.out + (int64_t)v * a.ld_out + c, S);
        } else {
            for (int i = 0; i < d; ++i) {
                const int e = a.pv.perm[beg + i];
                const int er = a.pv.rev[e];
                V::store(a.out + (int64_t)er * a.ld_out + c, V::sub(S, load_row<VEC, UNDIR, ACT>(a, e, er, c, slope)));
            }
        }
    }
}

// MODE 0: message (atom form when the graph is symmetric, edge form otherwise); MODE 1: aggregate.
template <int VEC, int MODE, int UNDIR, int ACT>
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_segment(SegArgs a) {
    using V = Vec<VEC>;
    using T = typename V::T;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * kWavesPerBlock;
    const float slope = a.slope_ptr ? *a.slope_ptr : a.slope;
    const bool asym = (MODE == 0) && (a.pv.hdr[DMPNN_HDR_FLAGS] & PLAN_ASYMMETRIC);
    const int n_cols = a.h / VEC;  // column groups of VEC floats

    if (!asym) {
        for (int v = wave; v < a.nV; v += n_waves) {
            const int beg = a.pv.row_ptr[v];
            const int d = a.pv.row_ptr[v + 1] - beg;
            if (ACT == -1) {  // generic build: one compact loop for every in-degree
                if (d > 0 || MODE == 1) atom_body_any<VEC, MODE, UNDIR, ACT>(a, v, beg, d, lane, n_cols, slope);
                continue;
            }
            switch (d) {  // wave-uniform: one straight-line body per in-degree
                case 0:
                    if (MODE == 1)
                        for (int cg = lane; cg < n_cols; cg += 64) V::store(a.out + (int64_t)v * a.ld_out + cg * VEC, V::zero());
                    break;
                case 1: atom_body<VEC, MODE, UNDIR, ACT, 1>(a, v, beg, lane, n_cols, slope); break;
                case 2: atom_body<VEC, MODE, UNDIR, ACT, 2>(a, v, beg, lane, n_cols, slope); break;
                case 3: atom_body<VEC, MODE, UNDIR, ACT, 3>(a, v, beg, lane, n_cols, slope); break;
                case 4: atom_body<VEC, MODE, UNDIR, ACT, 4>(a, v, beg, lane, n_cols, slope); break;
                case 5: atom_body<VEC, MODE, UNDIR, ACT, 5>(a, v, beg, lane, n_cols, slope); break;
                case 6: atom_body<VEC, MODE, UNDIR, ACT, 6>(a, v, beg, lane, n_cols, slope); break;
                default: atom_body_any<VEC, MODE, UNDIR, ACT>(a, v, beg, d, lane, n_cols, slope); break;
            }
        }
    } else {
        // literal mixins.py:11-18 for arbitrary (in-range) index arrays
        for (int e = wave; e < a.nE; e += n_waves) {
            const int v = a.pv.src[e];
            const int beg = a.pv.row_ptr[v];
            const int d = a.pv.row_ptr[v + 1] - beg;
            const int er = a.pv.rev[e];
            for (int cg = lane; cg < n_cols; cg += 64) {
                const int c = cg * VEC;
                T S = V::zero();
                for (int i = 0; i < d; ++i) {
                    const int ei = a.pv.perm[beg + i];
                    const T r = load_row<VEC, UNDIR, ACT>(a, ei, a.pv.rev[ei], c, slope);
                    S = (i == 0) ? r : V::add(S, r);
                }
                V::store(a.out + (int64_t)e * a.ld_out + c, V::sub(S, load_row<VEC, UNDIR, ACT>(a, er, a.pv.rev[er], c, slope)));
            }
        }
    }
}

template <int MODE>
int launch_segment(const SegArgs& a, hipStream_t s, const char* name) {
    const int64_t items = (MODE == 0) ? (a.nV > a.nE ? a.nV : a.nE) : a.nV;
    if (items == 0 || a.h == 0) return DMPNN_OK;
    int64_t blocks = (items + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t cap = 256 * 32;  // grid-stride beyond 32 blocks per CU
    if (blocks > cap) blocks = cap;
    const bool vec = (a.h % 4 == 0) && (a.ld_in % 4 == 0) && (a.ld_out % 4 == 0) && aligned16(a.Hin) && aligned16(a.out);
    const dim3 grid((unsigned)blocks), block(kWavesPerBlock * 64);
    // hot instantiations: 16-byte lanes, directed, tau in {identity, ReLU}; everything else -> generic build
    if (vec && !a.undirected && a.act == DMPNN_ACT_NONE)
        hipLaunchKernelGGL((k_segment<4, MODE, 0, DMPNN_ACT_NONE>), grid, block, 0, s, a);
    else if (vec && !a.undirected && a.act == DMPNN_ACT_RELU)
        hipLaunchKernelGGL((k_segment<4, MODE, 0, DMPNN_ACT_RELU>), grid, block, 0, s, a);
    else if (vec)
        hipLaunchKernelGGL((k_segment<4, MODE, -1, -1>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((k_segment<1, MODE, -1, -1>), grid, block, 0, s, a);
    DMPNN_CHECK_LAUNCH(name);
    return DMPNN_OK;
}

}  // namespace

int launch_message(const PlanView& pv, int64_t nV, int64_t nE, int64_t h, const float* Hin,
                   int64_t ld_in, float* M, int64_t ld_m, int act, float slope,
                   const float* slope_ptr, unsigned flags, hipStream_t s) {
    SegArgs a{pv, (int)nV, (int)nE, (int)h, Hin, ld_in, M, ld_m, act, slope, slope_ptr,
              (flags & DMPNN_F_UNDIRECTED) ? 1 : 0};
    if (nE == 0) return DMPNN_OK;
    return launch_segment<0>(a, s, "k_segment<message>");
}

int launch_aggregate(const PlanView& pv, int64_t nV, int64_t nE, int64_t h, const float* Hin,
                     int64_t ld_in, float* Mv, int64_t ld_mv, int act, float slope,
                     const float* slope_ptr, hipStream_t s) {
    SegArgs a{pv, (int)nV, (int)nE, (int)h, Hin, ld_in, Mv, ld_mv, act, slope, slope_ptr, 0};
    return launch_segment<1>(a, s, "k_segment<aggregate>");
}

}  // namespace dmpnn
