// K6 — backward of the path (what autograd does through base.py:196-212 / mixins.py:8-18).
//
// With S[v] = sum_{dst(e)=v} Hb[e],  M[e] = S[src(e)] - Hb[rev(e)]  (Hb = H, or (H+H[rev])/2 when
// undirected),  Z = H0 + M.W_h^T (+b),  H' = tau(Z):
//
//   gZ  = gH' * tau'(.)                         (mask, fused into the producer of gH')
//   gW_h += gZ^T . M        gb_h += colsum(gZ)   (k_wgrad: MFMA, split over edges, slab reduce)
//   gM  = gZ . W_h                               (the forward contraction kernel on W_h^T)
//   gHb[e'] = gS[dst(e')] - gM[rev(e')],  gS[v] = sum_{src(e)=v} gM[e] = sum_{dst(e')=v} gM[rev(e')]
//                                               (k_edge_bwd<MESSAGE>: the forward atom-centric
//                                                kernel with read row rev(e') and write row e')
//   gH0 accumulates every gZ^(t) plus gH^(0) * tau'(tau(H0)).
//
// Graph must be symmetric (every featurizer-produced graph is); on an asymmetric plan the message
// backward writes NaN (loud) — gradients through arbitrary index arrays are not provided.
#include <stdlib.h>
#include <string.h>

#include "dmpnn_common.hpp"

namespace dmpnn {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// edge-space backward kernels
// ------------------------------------------------------------------------------------------------
enum : int { EB_GATHER = 0, EB_MESSAGE = 1, EB_AVG = 2 };

struct EdgeBwdArgs {
    PlanView pv;
    int nV, nE, h;
    const float* gin;  // GATHER: gMv [V]   MESSAGE: gM [E]   AVG: raw gHb [E]
    int64_t ld_gin;
    const float* Y;    // activation output (or pre-activation when y_preact) of the TARGET rows; may be null
    int64_t ldy;
    int y_preact;
    float* gZ;         // masked gradient out [E]; may be null
    int64_t ldgz;
    float* acc;        // gH0 accumulator [E]; may be null
    int64_t ldacc;
    int acc_init;      // 1: acc = g, 0: acc += g
    int act;
    float slope;
    const float* slope_ptr;
    int poison_mask;   // plan flags that make the message backward write NaN
};

template <int ACT>
__device__ __forceinline__ float mask1(float g, float y, int act_rt, float slope, int preact) {
    if (ACT == DMPNN_ACT_NONE) return g;
    if (ACT == DMPNN_ACT_RELU) return y > 0.f ? g : 0.f;  // sign(relu(z)) == sign(z): no need to re-apply tau
    // generic
    if (preact) y = apply_act(y, act_rt, slope);
    return g * act_grad_from_out(y, act_rt, slope);
}

// finish one float4 (or float) of target row `row`: mask, store gZ, accumulate gH0
template <int VEC, int ACT>
__device__ __forceinline__ void finish(const EdgeBwdArgs& a, int row, int c, bool ok, float slope,
                                       float gx, float gy, float gz, float gw,
                                       float yx, float yy, float yz, float yw,
                                       float ax, float ay, float az, float aw) {
    float o0 = mask1<ACT>(gx, yx, a.act, slope, a.y_preact);
    float o1 = 0.f, o2 = 0.f, o3 = 0.f;
    if (VEC == 4) {
        o1 = mask1<ACT>(gy, yy, a.act, slope, a.y_preact);
        o2 = mask1<ACT>(gz, yz, a.act, slope, a.y_preact);
        o3 = mask1<ACT>(gw, yw, a.act, slope, a.y_preact);
    }
    if (!ok) return;
    if (a.gZ) {
        float* p = a.gZ + (int64_t)row * a.ldgz + c;
        if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(o0, o1, o2, o3);
        else *p = o0;
    }
    if (a.acc) {
        float* p = a.acc + (int64_t)row * a.ldacc + c;
        if (!a.acc_init) { o0 += ax; o1 += ay; o2 += az; o3 += aw; }
        if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(o0, o1, o2, o3);
        else *p = o0;
    }
}

template <int VEC>
__device__ __forceinline__ float4 ld(const float* p) {
    if (VEC == 4) return *reinterpret_cast<const float4*>(p);
    return make_float4(*p, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// MESSAGE mode, atom with in-degree exactly D: D rows of gM, D rows of Y, D rows of acc in flight.
template <int VEC, int ACT, int D>
__device__ __forceinline__ void msg_bwd_body(const EdgeBwdArgs& a, int beg, int lane, int n_cols, float slope,
                                             const float* Yp, int64_t ldy, const float* Ap, int64_t lda) {
    int eid[D], erev[D];
#pragma unroll
    for (int i = 0; i < D; ++i) eid[i] = a.pv.perm[beg + i];
#pragma unroll
    for (int i = 0; i < D; ++i) erev[i] = a.pv.rev[eid[i]];
    for (int cg0 = 0; cg0 < n_cols; cg0 += 64) {
        const int cg = cg0 + lane;
        const bool ok = cg < n_cols;
        const int c = (ok ? cg : 0) * VEC;
        float4 r[D], y[D], ac[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            r[i] = ld<VEC>(a.gin + (int64_t)erev[i] * a.ld_gin + c);
            y[i] = ld<VEC>(Yp + (int64_t)eid[i] * ldy + c);
            ac[i] = ld<VEC>(Ap + (int64_t)eid[i] * lda + c);
        }
        float4 S = r[0];
#pragma unroll
        for (int i = 1; i < D; ++i) S = add4(S, r[i]);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const float4 g = sub4(S, r[i]);
            finish<VEC, ACT>(a, eid[i], c, ok, slope, g.x, g.y, g.z, g.w, y[i].x, y[i].y, y[i].z, y[i].w,
                             ac[i].x, ac[i].y, ac[i].z, ac[i].w);
        }
    }
}

template <int VEC, int ACT>
__device__ __forceinline__ void msg_bwd_any(const EdgeBwdArgs& a, int beg, int d, int lane, int n_cols, float slope,
                                            const float* Yp, int64_t ldy, const float* Ap, int64_t lda) {
    for (int cg = lane; cg < n_cols; cg += 64) {
        const int c = cg * VEC;
        float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < d; ++i) S = add4(S, ld<VEC>(a.gin + (int64_t)a.pv.rev[a.pv.perm[beg + i]] * a.ld_gin + c));
        for (int i = 0; i < d; ++i) {
            const int e = a.pv.perm[beg + i];
            const float4 g = sub4(S, ld<VEC>(a.gin + (int64_t)a.pv.rev[e] * a.ld_gin + c));
            const float4 y = ld<VEC>(Yp + (int64_t)e * ldy + c);
            const float4 ac = ld<VEC>(Ap + (int64_t)e * lda + c);
            finish<VEC, ACT>(a, e, c, true, slope, g.x, g.y, g.z, g.w, y.x, y.y, y.z, y.w, ac.x, ac.y, ac.z, ac.w);
        }
    }
}

template <int VEC, int MODE, int ACT>
__global__ __launch_bounds__(256) void k_edge_bwd(EdgeBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    const float slope = a.slope_ptr ? *a.slope_ptr : a.slope;
    const int n_cols = a.h / VEC;
    // dummy bases keep every load unconditional
    const float* Yp = a.Y ? a.Y : a.gin;
    const int64_t ldy = a.Y ? a.ldy : 0;
    const float* Ap = (a.acc && !a.acc_init) ? a.acc : a.gin;
    const int64_t lda = (a.acc && !a.acc_init) ? a.ldacc : 0;

    if (MODE == EB_MESSAGE) {
        const bool asym = a.pv.hdr[DMPNN_HDR_FLAGS] & a.poison_mask;
        if (asym) {  // gradients through a non-molecular index structure are not provided: poison loudly
            const float nanv = __int_as_float(0x7fc00000);
            for (int e = wave; e < a.nE; e += n_waves)
                for (int c = lane; c < a.h; c += 64) {
                    if (a.gZ) a.gZ[(int64_t)e * a.ldgz + c] = nanv;
                    if (a.acc) a.acc[(int64_t)e * a.ldacc + c] = nanv;
                }
            return;
        }
        for (int v = wave; v < a.nV; v += n_waves) {
            const int beg = a.pv.row_ptr[v];
            const int d = a.pv.row_ptr[v + 1] - beg;
            if (ACT == -1) {
                msg_bwd_any<VEC, ACT>(a, beg, d, lane, n_cols, slope, Yp, ldy, Ap, lda);
                continue;
            }
            switch (d) {
                case 0: break;
                case 1: msg_bwd_body<VEC, ACT, 1>(a, beg, lane, n_cols, slope, Yp, ldy, Ap, lda); break;
                case 2: msg_bwd_body<VEC, ACT, 2>(a, beg, lane, n_cols, slope, Yp, ldy, Ap, lda); break;
                case 3: msg_bwd_body<VEC, ACT, 3>(a, beg, lane, n_cols, slope, Yp, ldy, Ap, lda); break;
                case 4: msg_bwd_body<VEC, ACT, 4>(a, beg, lane, n_cols, slope, Yp, ldy, Ap, lda); break;
                default: msg_bwd_any<VEC, ACT>(a, beg, d, lane, n_cols, slope, Yp, ldy, Ap, lda); break;
            }
        }
    } else {
        // GATHER: g = gMv[dst(e)]      AVG: g = (gHb[e] + gHb[rev(e)]) / 2       one wave per edge
        for (int e = wave; e < a.nE; e += n_waves) {
            const int r0 = (MODE == EB_GATHER) ? a.pv.dst[e] : e;
            const int r1 = (MODE == EB_GATHER) ? r0 : a.pv.rev[e];
            for (int cg = lane; cg < n_cols; cg += 64) {
                const int c = cg * VEC;
                float4 g = ld<VEC>(a.gin + (int64_t)r0 * a.ld_gin + c);
                const float4 g1 = ld<VEC>(a.gin + (int64_t)r1 * a.ld_gin + c);
                const float4 y = ld<VEC>(Yp + (int64_t)e * ldy + c);
                const float4 ac = ld<VEC>(Ap + (int64_t)e * lda + c);
                if (MODE == EB_AVG) g = make_float4((g.x + g1.x) / 2.f, (g.y + g1.y) / 2.f, (g.z + g1.z) / 2.f, (g.w + g1.w) / 2.f);
                finish<VEC, ACT>(a, e, c, true, slope, g.x, g.y, g.z, g.w, y.x, y.y, y.z, y.w, ac.x, ac.y, ac.z, ac.w);
            }
        }
    }
}

template <int MODE>
int launch_edge_bwd(EdgeBwdArgs a, hipStream_t s, const char* name) {
    if (a.nE == 0 || a.h == 0) return DMPNN_OK;
    if (!a.Y) a.act = DMPNN_ACT_NONE;
    if (!a.poison_mask) a.poison_mask = PLAN_ASYMMETRIC;
    const int64_t items = (MODE == EB_MESSAGE) ? a.nV : a.nE;
    int64_t blocks = (items + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    bool vec = (a.h % 4 == 0) && (a.ld_gin % 4 == 0) && aligned16(a.gin);
    if (a.Y) vec = vec && (a.ldy % 4 == 0) && aligned16(a.Y);
    if (a.gZ) vec = vec && (a.ldgz % 4 == 0) && aligned16(a.gZ);
    if (a.acc) vec = vec && (a.ldacc % 4 == 0) && aligned16(a.acc);
    const dim3 grid((unsigned)blocks), block(256);
    if (vec && a.act == DMPNN_ACT_NONE) hipLaunchKernelGGL((k_edge_bwd<4, MODE, DMPNN_ACT_NONE>), grid, block, 0, s, a);
    else if (vec && a.act == DMPNN_ACT_RELU) hipLaunchKernelGGL((k_edge_bwd<4, MODE, DMPNN_ACT_RELU>), grid, block, 0, s, a);
    else if (vec) hipLaunchKernelGGL((k_edge_bwd<4, MODE, -1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_edge_bwd<1, MODE, -1>), grid, block, 0, s, a);
    DMPNN_CHECK_LAUNCH(name);
    return DMPNN_OK;
}

// elementwise: gZ = g * tau'(Y)      (finalize: atoms x h)
template <int VEC>
__global__ void k_act_bwd(const float* __restrict__ g, int64_t ldg, const float* __restrict__ Y, int64_t ldy,
                          float* __restrict__ out, int64_t ldo, int64_t rows, int h, int act, float slope,
                          const float* slope_ptr) {
    const float sl = slope_ptr ? *slope_ptr : slope;
    const int q = h / VEC;  // column groups per row
    const int64_t n = rows * q;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = n < (int64_t(1) << 31) ? (int64_t)((unsigned)i / (unsigned)q) : i / q;
        const int c = (int)(i - r * q) * VEC;
        if (VEC == 4) {
            const float4 gv = *reinterpret_cast<const float4*>(g + r * ldg + c);
            const float4 yv = *reinterpret_cast<const float4*>(Y + r * ldy + c);
            *reinterpret_cast<float4*>(out + r * ldo + c) =
                make_float4(gv.x * act_grad_from_out(yv.x, act, sl), gv.y * act_grad_from_out(yv.y, act, sl),
                            gv.z * act_grad_from_out(yv.z, act, sl), gv.w * act_grad_from_out(yv.w, act, sl));
        } else {
            out[r * ldo + c] = g[r * ldg + c] * act_grad_from_out(Y[r * ldy + c], act, sl);
        }
    }
}

// out[c][r] = in[r][c]   (weights only: a few hundred KB)
__global__ void k_transpose(const float* __restrict__ in, int64_t ldi, float* __restrict__ out, int64_t ldo,
                            int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = by + j, c = bx + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? in[(int64_t)r * ldi + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = bx + j, r = by + threadIdx.x;
        if (c < cols && r < rows) out[(int64_t)c * ldo + r] = tile[threadIdx.x][j];
    }
}

int launch_transpose(const float* in, int64_t ldi, float* out, int64_t ldo, int rows, int cols, hipStream_t s) {
    if (rows == 0 || cols == 0) return DMPNN_OK;
    hipLaunchKernelGGL(k_transpose, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, s, in, ldi, out, ldo, rows, cols);
    DMPNN_CHECK_LAUNCH("k_transpose");
    return DMPNN_OK;
}

// ------------------------------------------------------------------------------------------------
// weight gradient:  slab[split][n][k] = sum_{m in split} gZ[m][n] * Acat[m][k],   Acat = [A1[g(m)] || A2 || 1]
// ------------------------------------------------------------------------------------------------
// Both operands are "reduction-major" (m is the row index of gZ and of Acat), i.e. contiguous along
// the OUTPUT index.  MFMA 16x16x4 wants A[i][kk] from lane (i = l&15, kk = l>>4): lane loads ONE
// float4 gZ[m0+kk][n0+4i .. +3] and ONE float4 Acat[m0+kk][k0+4j .. +3]; component jn of the first
// and jk of the second feed the accumulator of the INTERLEAVED tile (jn, jk), whose rows are
// n = n0+4i+jn and columns k = k0+4j+jk.  One 16-byte load per operand per lane drives 16 MFMAs,
// straight from global memory (coalesced 256-B row segments), no LDS staging.
struct WgradArgs {
    int64_t M;
    int N, K1, K2, ones;
    const float* gZ; int64_t ldz;
    const float* A1; int64_t lda1; const int* gather1;
    const float* A2; int64_t lda2; const int* gather2;
    float* slab; int ldk; int64_t slab_stride;
    int rows_per_wg, splits;
    int vecZ, vecA;
};

constexpr int WG_U = 4;  // k-steps (of 4 rows) per batch of loads

__device__ __forceinline__ float4 wg_load_z(const WgradArgs& a, int64_t m, bool mok, int n) {
    // 4 consecutive output rows n..n+3 of gZ row m
    if (a.vecZ) {
        const bool ok = mok && n < a.N;
        float4 v = *reinterpret_cast<const float4*>(a.gZ + (ok ? m * a.ldz + n : 0));
        v.x = (ok) ? v.x : 0.f;
        v.y = (ok && n + 1 < a.N) ? v.y : 0.f;
        v.z = (ok && n + 2 < a.N) ? v.z : 0.f;
        v.w = (ok && n + 3 < a.N) ? v.w : 0.f;
        return v;
    }
    float x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool ok = mok && (n + t) < a.N;
        const float raw = a.gZ[ok ? m * a.ldz + n + t : 0];
        x[t] = ok ? raw : 0.f;
    }
    return make_float4(x[0], x[1], x[2], x[3]);
}

template <bool FULL>
__device__ __forceinline__ float4 wg_load_a(const WgradArgs& a, int64_t g2, int64_t g1, bool mok, int k) {
    const int K = a.K1 + a.K2;
    if (FULL) {  // the block's 64 columns are all < K and K1, K2 are multiples of 4: one 16-byte load
        const float* p = (k < a.K1) ? a.A1 + g1 * a.lda1 + k : a.A2 + g2 * a.lda2 + (k - a.K1);
        float4 v = *reinterpret_cast<const float4*>(mok ? p : a.A1);
        return mok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int kt = k + t;
        const bool ok = mok && kt < K;
        const float* p = (kt < a.K1) ? a.A1 + g1 * a.lda1 + kt : a.A2 + g2 * a.lda2 + (kt - a.K1);
        const float raw = *(ok ? p : a.gZ);
        x[t] = ok ? raw : ((mok && a.ones && kt == K) ? 1.f : 0.f);
    }
    return make_float4(x[0], x[1], x[2], x[3]);
}

template <bool FULL>
__device__ __forceinline__ void wgrad_loop(const WgradArgs& a, f32x4 (&acc)[4][4], int64_t m_lo, int64_t m_hi, int n0,
                                           int k0, int wave, int li, int lg) {
    // software pipeline: the loads of batch b+1 are in flight while the 64 MFMAs of batch b run
    auto load_batch = [&](int64_t mb, float4 (&z)[WG_U], float4 (&x)[WG_U]) {
#pragma unroll
        for (int u = 0; u < WG_U; ++u) {
            const int64_t m = mb + 16 * u + lg;
            const bool mok = m < m_hi;
            const int64_t mc = mok ? m : m_lo;
            const int64_t g1 = a.gather1 ? (int64_t)a.gather1[mc] : mc;
            const int64_t g2 = a.gather2 ? (int64_t)a.gather2[mc] : mc;
            z[u] = wg_load_z(a, mc, mok, n0 + 4 * li);
            x[u] = wg_load_a<FULL>(a, g2, g1, mok, k0 + 4 * li);
        }
    };
    auto mfma_batch = [&](const float4 (&z)[WG_U], const float4 (&x)[WG_U]) {
#pragma unroll
        for (int u = 0; u < WG_U; ++u) {
            const float zz[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
            const float xx[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int jk = 0; jk < 4; ++jk)
                    acc[jn][jk] = __builtin_amdgcn_mfma_f32_16x16x4f32(zz[jn], xx[jk], acc[jn][jk], 0, 0, 0);
        }
    };
    int64_t mb = m_lo + 4 * wave;
    if (mb >= m_hi) return;
    float4 z0[WG_U], x0[WG_U], z1[WG_U], x1[WG_U];
    load_batch(mb, z0, x0);
    for (;;) {
        const int64_t mb1 = mb + 16 * WG_U;
        if (mb1 < m_hi) load_batch(mb1, z1, x1);
        mfma_batch(z0, x0);
        if (mb1 >= m_hi) break;
        const int64_t mb2 = mb1 + 16 * WG_U;
        if (mb2 < m_hi) load_batch(mb2, z0, x0);
        mfma_batch(z1, x1);
        if (mb2 >= m_hi) break;
        mb = mb2;
    }
}

__global__ __launch_bounds__(256) void k_wgrad(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4][64][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int Kt = a.K1 + a.K2 + a.ones;
    const int kb = (Kt + 63) / 64;
    // XCD-aware order: workgroups go round-robin to the 8 XCDs by linear id, and every (n, k) tile of one row split reads
    // the same operand rows - so XCD c takes the contiguous (split, tile) ranks [c G/8, (c+1) G/8): each L2 then fills with
    // the rows of ~1/8 of the splits instead of all of them (measured: 1.6x on this kernel at QM9-512)
    const int tiles = ((a.N + 63) / 64) * kb;
    const int per = gridDim.x >> 3;
    const int rank = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (rank >= tiles * a.splits) return;
    const int split = rank / tiles, tile = rank - split * tiles;
    const int n0 = (tile / kb) * 64, k0 = (tile % kb) * 64;
    const int64_t m_lo = (int64_t)split * a.rows_per_wg;
    int64_t m_hi = m_lo + a.rows_per_wg;
    if (m_hi > a.M) m_hi = a.M;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // wave w owns k-steps t = w, w+4, ...; a batch is WG_U of them
    const bool full = a.vecA && (k0 + 64 <= a.K1 + a.K2);  // uniform: picks one straight-line loop
    if (full) wgrad_loop<true>(a, acc, m_lo, m_hi, n0, k0, wave, li, lg);
    else wgrad_loop<false>(a, acc, m_lo, m_hi, n0, k0, wave, li, lg);
    // cross-wave reduction through LDS, then one slab write per workgroup
    float* mine = red + wave * 4096;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
        for (int jk = 0; jk < 4; ++jk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nl = 4 * (lg * 4 + r) + jn, kl = 4 * li + jk;
                mine[nl * 64 + kl] = acc[jn][jk][r];
            }
    __syncthreads();
    float* slab = a.slab + (int64_t)split * a.slab_stride;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int nl = idx >> 6, kl = idx & 63;
        const int n = n0 + nl, k = k0 + kl;
        if (n < a.N && k < Kt) slab[(int64_t)n * a.ldk + k] = red[idx] + red[4096 + idx] + red[8192 + idx] + red[12288 + idx];
    }
}

// gW[n][k] = sum_s slab[s][n][k] (k < K),  gb[n] = sum_s slab[s][n][K]
__global__ void k_wgrad_reduce(const float* __restrict__ slab, int64_t slab_stride, int n_slabs, int ldk, int N, int K,
                               int ones, float* __restrict__ gW, int64_t ldgw, float* __restrict__ gb,
                               const int* __restrict__ poison_flags, int poison_mask) {
    const int Kt = K + ones;
    const bool poison = poison_flags && (poison_flags[0] & poison_mask);
    const int64_t total = (int64_t)N * Kt;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kt), k = (int)(i % Kt);
        // eight slab reads in flight at a time (clamped index, masked add: no load under a branch), summed in slab order
        float s = 0.f;
        const float* p0 = slab + (int64_t)n * ldk + k;
        for (int t0 = 0; t0 < n_slabs; t0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u < n_slabs ? t0 + u : n_slabs - 1;
                v[u] = p0[(int64_t)t * slab_stride];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t0 + u < n_slabs ? v[u] : 0.f;
        }
        if (poison) s = __int_as_float(0x7fc00000);
        if (k < K) {
            if (gW) gW[(int64_t)n * ldgw + k] = s;
        } else if (gb) {
            gb[n] = s;
        }
    }
}

// the same for up to four products in ONE launch (the backward's weight gradients are ready together): job j owns the
// workgroups [wg0[j], wg0[j + 1])
struct ReduceJob {
    const float* slab; int64_t slab_stride; int n_slabs, ldk, N, K, ones;
    float* gW; int64_t ldgw; float* gb;
};
struct ReduceJobs { ReduceJob job[8]; int wg0[9]; int n_jobs; const int* poison_flags; int poison_mask; };
__global__ void k_wgrad_reduce_multi(ReduceJobs a) {
    int j = 0;
    while (j + 1 < a.n_jobs && (int)blockIdx.x >= a.wg0[j + 1]) ++j;
    const ReduceJob& J = a.job[j];
    const int Kt = J.K + J.ones;
    const bool poison = a.poison_flags && (a.poison_flags[0] & a.poison_mask);
    const int64_t total = (int64_t)J.N * Kt;
    const int nblk = a.wg0[j + 1] - a.wg0[j];
    for (int64_t i = ((int64_t)blockIdx.x - a.wg0[j]) * blockDim.x + threadIdx.x; i < total; i += (int64_t)nblk * blockDim.x) {
        const int n = (int)(i / Kt), k = (int)(i % Kt);
        float s = 0.f;
        const float* p0 = J.slab + (int64_t)n * J.ldk + k;
        for (int t0 = 0; t0 < J.n_slabs; t0 += 8) {  // (summed in slab order, eight reads in flight: as k_wgrad_reduce)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u < J.n_slabs ? t0 + u : J.n_slabs - 1;
                v[u] = p0[(int64_t)t * J.slab_stride];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t0 + u < J.n_slabs ? v[u] : 0.f;
        }
        if (poison) s = __int_as_float(0x7fc00000);
        if (k < J.K) {
            if (J.gW) J.gW[(int64_t)n * J.ldgw + k] = s;
        } else if (J.gb) {
            J.gb[n] = s;
        }
    }
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct WgradPlan {
    int splits, rows_per_wg, ldk;
    int64_t slab_stride;
    // the same product on the f16 pipe (dmpnn_wgrad16.hip: operands split once, then LDS-DMA + 3-pass MFMA) where it pays — from
    // ~1 k reduction rows on; taken at launch when the operands allow it (pairs of columns 8-byte loadable), else the fp32 kernel
    bool can16; WProdPlan q; size_t split_floats;
    int max_splits() const { return can16 && q.splits > splits ? q.splits : splits; }
};
WgradPlan plan_wgrad(int64_t M, int N, int Kt) {
    WgradPlan p;
    const int nb = (N + 63) / 64, kb = (Kt + 63) / 64;
    int splits = 512 / (nb * kb);
    if (splits < 1) splits = 1;
    const int64_t max_splits = (M + 63) / 64;
    if (splits > max_splits) splits = (int)(max_splits > 0 ? max_splits : 1);
    int64_t rows = (M + splits - 1) / splits;
    rows = (rows + 15) / 16 * 16;
    if (rows < 16) rows = 16;
    p.splits = (int)((M + rows - 1) / rows);
    if (p.splits < 1) p.splits = 1;
    p.rows_per_wg = (int)rows;
    p.ldk = (Kt + 3) / 4 * 4;
    p.slab_stride = (int64_t)N * p.ldk;
    p.can16 = M >= 1024 && N % 2 == 0;  // (at 355 702 rows as well: the fp32-MFMA product there is 8.4 ms per step against 7.0, scripts/ab_configs.py)
    p.q = plan_wgrad16(M, N, Kt);
    p.split_floats = p.can16 ? align_up((wsplit16_bytes(M, N) + 3) / 4, 64) + align_up((wsplit16_bytes(M, Kt) + 3) / 4, 64) : 0;
    return p;
}

// -> *n_slabs (when given): how many slabs were written (the f16 and the fp32 kernel split the rows differently)
int launch_wgrad(WgradArgs a, const WgradPlan& p, float* slab, hipStream_t s, float* split_ws = nullptr, int* n_slabs = nullptr) {
    if (n_slabs) *n_slabs = p.splits;
    if (a.N == 0) return DMPNN_OK;
    const int Kt = a.K1 + a.K2 + a.ones;
    if (Kt == 0) return DMPNN_OK;
    if (p.can16 && split_ws && n_slabs && aligned16(split_ws) && wgrad16_operand_ok(a.gZ, a.ldz, a.N, nullptr, 0, 0) &&
        ((a.K1 > 0 && wgrad16_operand_ok(a.A1, a.lda1, a.K1, a.A2, a.lda2, a.K2)) ||
         (a.K1 == 0 && wgrad16_operand_ok(a.A2, a.lda2, a.K2, nullptr, 0, 0)))) {
        WSplitArgs sp;
        memset(&sp, 0, sizeof(sp));
        sp.n_jobs = 2;
        wsplit16_job(&sp.job[0], a.M, a.N, a.gZ, a.ldz, nullptr, a.N, nullptr, 0, nullptr, 0, 0, split_ws);
        float* aw = split_ws + align_up((wsplit16_bytes(a.M, a.N) + 3) / 4, 64);
        if (a.K1 > 0) wsplit16_job(&sp.job[1], a.M, Kt, a.A1, a.lda1, a.gather1, a.K1, a.A2, a.lda2, a.gather2, a.K2, a.ones, aw);
        else wsplit16_job(&sp.job[1], a.M, Kt, a.A2, a.lda2, a.gather2, a.K2, nullptr, 0, nullptr, 0, a.ones, aw);
        DMPNN_TRY(launch_wsplit16(sp, s));
        WProdJobs pj;
        memset(&pj, 0, sizeof(pj));
        wgrad16_add(&pj, sp.job[0], sp.job[1], p.q, a.N, Kt, slab);
        DMPNN_TRY(launch_wgrad16(pj, s));
        *n_slabs = p.q.splits;
        return DMPNN_OK;
    }
    a.slab = slab; a.ldk = p.ldk; a.slab_stride = p.slab_stride; a.rows_per_wg = p.rows_per_wg;
    a.vecZ = aligned16(a.gZ) && a.ldz % 4 == 0;
    a.vecA = (a.K1 == 0 || (aligned16(a.A1) && a.lda1 % 4 == 0)) && a.K1 % 4 == 0 &&
             (a.K2 == 0 || (aligned16(a.A2) && a.lda2 % 4 == 0 && a.K2 % 4 == 0));
    if (a.K1 == 0) { a.A1 = a.A2; a.lda1 = a.lda2; a.gather1 = nullptr; }
    if (a.K2 == 0) { a.A2 = a.A1; a.lda2 = a.lda1; a.gather2 = nullptr; }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(k_wgrad): %s", hipGetErrorString(e));
            return DMPNN_EHIP;
        }
        attr_set = true;
    }
    const int nb = (a.N + 63) / 64, kb = (Kt + 63) / 64;
    a.splits = p.splits;
    const int total = nb * kb * p.splits;
    hipLaunchKernelGGL(k_wgrad, dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), 65536, s, a);  // (8 | grid: the XCD order)
    DMPNN_CHECK_LAUNCH("k_wgrad");
    return DMPNN_OK;
}

int launch_wgrad_reduce(const float* slab, const WgradPlan& p, int n_slabs, int N, int K, int ones, float* gW,
                        int64_t ldgw, float* gb, hipStream_t s, const int* poison_flags = nullptr, int poison_mask = 0) {
    const int64_t total = (int64_t)N * (K + ones);
    if (total == 0 || (!gW && !gb)) return DMPNN_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)blocks), dim3(256), 0, s, slab, p.slab_stride, n_slabs, p.ldk, N,
                       K, ones, gW, ldgw, gb, poison_flags, poison_mask);
    DMPNN_CHECK_LAUNCH("k_wgrad_reduce");
    return DMPNN_OK;
}

struct BwdLayout {
    size_t gZa, gZb, gH0, gZO, gMv, gHO, WhT, WoT, WdT, slab_h, slab_x, total;
    size_t WhT16, WoT16;  // pre-split transposed weights of the data-gradient contractions on the f16 pipe (large batches)
    bool use16;
    size_t mega_w, gZs;   // backward tile kernel: its two pre-split matrices; gZ^(t) slots beyond the two ping-pong buffers
    size_t sp_gM;         // backward tile kernel: edge scratch of its generic path for oversize pieces (the atom scratch is gMv)
    bool mega;
    WgradPlan p_h, p_i, p_o, p_d;
    WgradPlan p_hm;       // backward tile kernel: gW_h over ALL steps' rows in one launch (the gZ^(t) / M^(t) slots are adjacent)
    // backward tile kernel: the three weight gradients on the f16 pipe (dmpnn_wgrad16.hip): six split operands + slabs
    bool w16;
    size_t w16g;                              // per-step path: split operands of one product at a time
    size_t w16_z[3], w16_a[3], w16_slab[3];   // o, h, i (float offsets)
    WProdPlan q[3];
    // tile kernels with the kept messages as split rows (mega16_keeps_rows): every weight-gradient operand as split rows — the gradients
    // written so by the backward tile kernel, [V[src] || E] and [V || Mv] by k_rows2sr — and ALL products in one k_wgrad16r launch
    bool sr;
    int tsr, tsx, tsv;                          // row bytes: d_h columns | d_v + d_e | d_v + d_h
    size_t sr_gz, sr_gh0, sr_gzo, sr_x, sr_vm, sr_slab;
    WProdRPlan r_h, r_i, r_o, r_bh, r_bi, r_bo;
    int sr_R;                                  // rows per workgroup of the product launch (wgrad16r_rows_per_split)
    // lean training on the per-step fused route (dmpnn_bstep16.hip): gZ of every site as split rows, written by the step kernels; the
    // products (k_wgrad16r) read them and the forward's kept M^(t) / x rows as they are
    bool lean;
    size_t lz, lz_stride, lslab;               // gZ rows of every site | slabs (W_h | b_h | W_i | b_i)
    WProdRPlan rh, ri, rbh, rbi;               // (rbh / rbi: the bias gradients' column sums, riding in the products of W_h / W_i: their rows)
};
BwdLayout bwd_layout(const dmpnn_fwd_args& f) {
    BwdLayout L;
    const int64_t nV = f.n_atoms, nE = f.n_edges, h = f.d_h, dvd = f.W_d ? f.d_vd : 0;
    const size_t edge = align_up((size_t)nE * f.ldh, 4), atom = align_up((size_t)nV * f.ldh, 4);
    size_t o = 0;
    L.gZa = o; o += edge;
    L.gZb = o; o += edge;
    L.gH0 = o; o += edge;
    L.gZO = o; o += atom;
    L.gMv = o; o += atom;
    L.gHO = o; o += atom;
    L.WhT = o; o += align_up((size_t)h * h, 4);
    L.WoT = o; o += align_up((size_t)h * h, 4);
    L.WdT = o; o += align_up((size_t)h * (h + dvd), 4);
    L.p_h = plan_wgrad(nE, (int)h, (int)h + (f.b_h ? 1 : 0));
    L.p_i = plan_wgrad(nE, (int)h, (int)(f.d_v + f.d_e) + (f.b_i ? 1 : 0));
    L.p_o = plan_wgrad(nV, (int)h, (int)(f.d_v + h) + 1);
    L.p_d = plan_wgrad(nV, (int)(h + dvd), (int)(h + dvd) + 1);
    const int steps = f.depth > 1 ? f.depth - 1 : 1;
    L.p_hm = plan_wgrad(nE * steps, (int)h, (int)h + (f.b_h ? 1 : 0));
    {
        const size_t per_step = (size_t)L.p_h.max_splits() * steps * L.p_h.slab_stride, merged = (size_t)L.p_hm.max_splits() * L.p_hm.slab_stride;
        L.slab_h = o; o += align_up(per_step > merged ? per_step : merged, 4);
    }
    size_t x = (size_t)L.p_i.max_splits() * L.p_i.slab_stride;
    const size_t xo = (size_t)L.p_o.max_splits() * L.p_o.slab_stride, xd = dvd ? (size_t)L.p_d.max_splits() * L.p_d.slab_stride : 0;
    if (xo > x) x = xo;
    if (xd > x) x = xd;
    L.slab_x = o; o += align_up(x, 4);
    // data gradients gM = gZ . W_h, gMv = gZO . W_o[:, d_v:] on the f16 pipe (exact operand split, dmpnn_rows16.hip)
    // (about one tile per CU: the whole d_h-wide operand row in one group; large batches: 128-column groups, two
    // workgroups per CU)
    L.use16 = nE > 0 && h % 2 == 0 && f.ldh % 2 == 0;
    L.WhT16 = L.WoT16 = 0;
    if (L.use16) {
        const size_t w = align_up((linear16_wsplit_bytes(h, h) + 3) / 4, 64);
        L.WhT16 = o; o += w;
        L.WoT16 = o; o += w;
    }
    // The forward ran as the whole-forward tile kernel on the f16 pipe and kept its tensors: the data-gradient chain
    // runs as ONE tile kernel too (dmpnn_mega16_bwd.hip).  It stores every gZ^(t): depth - 1 edge buffers (the two
    // ping-pong buffers, which are adjacent, serve depth <= 3).
    L.mega = (f.flags & DMPNN_F_MEGA) && (f.flags & DMPNN_F_SPLIT16) && (f.flags & DMPNN_F_KEEP) && nE > 0 &&
             f.ldh % 4 == 0 && f.act != DMPNN_ACT_PRELU;
    L.mega_w = L.gZs = L.sp_gM = 0;
    if (L.mega) {
        L.mega_w = o; o += align_up((mega16_bwd_wsplit_bytes(h) + 3) / 4, 64);
        if (f.depth - 1 > 2) { L.gZs = o; o += (size_t)(f.depth - 1) * edge; }
        L.sp_gM = o; o += edge;
    }
    // split operands of ONE product at a time on the per-step path (the products run one after the other on the stream)
    {
        size_t m = L.p_h.split_floats;
        for (size_t v : {L.p_i.split_floats, L.p_o.split_floats, dvd ? L.p_d.split_floats : (size_t)0, L.p_hm.split_floats}) m = v > m ? v : m;
        L.w16g = o; o += m;
    }
    L.w16 = false;
    {
        // (atom messages, DMPNN_F_ATOM: W_i is [h, d_v], W_h [h, h + d_e] — its product reads [M^(t) || ME], ME the kept [.][16] rows)
        const bool atom = (f.flags & DMPNN_F_ATOM) != 0;
        const int kt_o = (int)(f.d_v + h) + 1, kt_h = (int)h + (atom ? (int)f.d_e : 0) + (f.b_h ? 1 : 0), kt_i = (int)f.d_v + (atom ? 0 : (int)f.d_e) + (f.b_i ? 1 : 0);
        if (L.mega && h % 2 == 0 && f.ldh % 2 == 0 && wgrad16_operand_ok(f.V, f.ldv, (int)f.d_v, nullptr, f.ldh, (int)h) &&
            (atom ? (f.d_e % 2 == 0 && f.d_e >= 2 && f.d_e <= 16) : (f.d_e == 0 || wgrad16_operand_ok(f.V, f.ldv, (int)f.d_v, f.E, f.lde, (int)f.d_e)))) {
            L.w16 = true;
            const int64_t Ms[3] = {nV, nE * steps, nE};
            const int Ks[3] = {kt_o, kt_h, kt_i};
            for (int i = 0; i < 3; ++i) {
                L.q[i] = plan_wgrad16(Ms[i], (int)h, Ks[i]);
                L.w16_z[i] = o; o += align_up((wsplit16_bytes(Ms[i], h) + 3) / 4, 64);
                L.w16_a[i] = o; o += align_up((wsplit16_bytes(Ms[i], Ks[i]) + 3) / 4, 64);
                L.w16_slab[i] = o; o += align_up((size_t)L.q[i].splits * L.q[i].slab_stride, 64);
            }
        }
    }
    L.sr = L.mega && mega16_keeps_rows(f) && f.d_v + f.d_e <= 512 && f.d_v + h <= 512;
    L.tsr = L.tsx = L.tsv = 0;
    L.sr_gz = L.sr_gh0 = L.sr_gzo = L.sr_x = L.sr_vm = L.sr_slab = 0;
    L.sr_R = 0;
    if (L.sr) {
        const int T = f.depth;
        const auto row_bytes = [](int64_t K) { return (int)(((K + 31) / 32) * 128 + 16); };
        L.tsr = (int)(split_row_floats(h) * 4); L.tsx = row_bytes(f.d_v + f.d_e); L.tsv = row_bytes(f.d_v + h);
        const auto fl = [&](size_t bytes) { return align_up((bytes + 3) / 4, 64); };
        L.sr_gz = o; o += fl((size_t)(T - 1) * nE * L.tsr);
        L.sr_gh0 = o; o += fl((size_t)nE * L.tsr);
        L.sr_gzo = o; o += fl((size_t)nV * L.tsr);
        L.sr_x = o; o += fl((size_t)nE * L.tsx);
        L.sr_vm = o; o += fl((size_t)nV * L.tsv);
        // the row counts per workgroup of the one product launch (8 workgroups left to a rider: dmpnn_train_step's first predictor layer);
        // a bias gradient rides in its product's workgroups — same rows, same number of slabs
        const int64_t Mj[3] = {nE * (T - 1), nV, nE};
        const int Kj[3] = {(int)h, (int)(f.d_v + h), (int)(f.d_v + f.d_e)};
        int Rj[3];
        wgrad16r_plan_launch(Mj, Kj, 3, 8, Rj);
        L.sr_R = Rj[0];
        L.r_h = plan_wgrad16r(Mj[0], (int)h, Kj[0], Rj[0]);
        L.r_o = plan_wgrad16r(Mj[1], (int)h, Kj[1], Rj[1]);
        L.r_i = plan_wgrad16r(Mj[2], (int)h, Kj[2], Rj[2]);
        L.r_bh = plan_wgrad16r(Mj[0], (int)h, 1, L.r_h.rows_per_split);
        L.r_bo = plan_wgrad16r(Mj[1], (int)h, 1, L.r_o.rows_per_split);
        L.r_bi = plan_wgrad16r(Mj[2], (int)h, 1, L.r_i.rows_per_split);
        L.sr_slab = o;
        for (const WProdRPlan* q : {&L.r_h, &L.r_i, &L.r_o, &L.r_bh, &L.r_bi, &L.r_bo}) o += align_up((size_t)q->splits * q->slab_stride, 64);
    }
    L.lean = fused16_lean(f);
    L.lz = L.lz_stride = L.lslab = 0;
    if (L.lean) {
        const int T = f.depth;
        // (2 T - 1 product jobs of n_edges rows each in ONE launch, the bias gradients riding in them: the row counts per workgroup of all of them)
        int64_t Mj[2 * kWProdMaxJobs];
        int Kj[2 * kWProdMaxJobs], Rj[2 * kWProdMaxJobs];
        int nj = 0;
        for (int t = 1; t < T; ++t) { Mj[nj] = nE; Kj[nj++] = (int)h; }
        for (int t = 0; t < T; ++t) { Mj[nj] = nE; Kj[nj++] = (int)(f.d_v + f.d_e); }
        wgrad16r_plan_launch(Mj, Kj, nj, 0, Rj);
        L.rh = plan_wgrad16r(nE, (int)h, (int)h, Rj[0]);
        L.ri = plan_wgrad16r(nE, (int)h, (int)(f.d_v + f.d_e), Rj[nj - 1]);
        L.rbh = plan_wgrad16r(nE, (int)h, 1, L.rh.rows_per_split);
        L.rbi = plan_wgrad16r(nE, (int)h, 1, L.ri.rows_per_split);
        L.lz_stride = align_up(((size_t)nE * (size_t)(split_row_floats(h) * 4) + 3) / 4, 64);
        L.lz = o; o += (size_t)T * L.lz_stride;
        L.lslab = o;
        o += align_up((size_t)(T - 1) * L.rh.splits * L.rh.slab_stride, 64) + align_up((size_t)T * L.ri.splits * L.ri.slab_stride, 64) +
             align_up((size_t)(T - 1) * L.rbh.splits * L.rbh.slab_stride, 64) + align_up((size_t)T * L.rbi.splits * L.rbi.slab_stride, 64);
    }
    L.total = o;
    return L;
}

}  // namespace

}  // namespace dmpnn

using namespace dmpnn;

extern "C" {

size_t dmpnn_backward_ws_bytes(const dmpnn_fwd_args* f) {
    if (!f) return 0;
    return bwd_layout(*f).total * sizeof(float);
}

int dmpnn_message_bwd(const void* plan, int64_t n_atoms, int64_t n_edges, int64_t d_h, const float* gM, int64_t ld_gm,
                      float* gH, int64_t ld_gh, void* stream) {
    DMPNN_CHECK_ARG(plan && d_h >= 0 && ld_gm >= d_h && ld_gh >= d_h, "message_bwd: bad arguments");
    DMPNN_CHECK_ARG(n_edges == 0 || (gM && gH), "message_bwd: null tensor");
    EdgeBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.pv = plan_view(plan, n_atoms, n_edges);
    a.nV = (int)n_atoms; a.nE = (int)n_edges; a.h = (int)d_h;
    a.gin = gM; a.ld_gin = ld_gm; a.gZ = gH; a.ldgz = ld_gh;
    return launch_edge_bwd<EB_MESSAGE>(a, static_cast<hipStream_t>(stream), "k_edge_bwd<message>");
}

int dmpnn_aggregate_bwd(const void* plan, int64_t n_atoms, int64_t n_edges, int64_t d_h, const float* gMv,
                        int64_t ld_gmv, float* gH, int64_t ld_gh, void* stream) {
    DMPNN_CHECK_ARG(plan && d_h >= 0 && ld_gmv >= d_h && ld_gh >= d_h, "aggregate_bwd: bad arguments");
    DMPNN_CHECK_ARG(n_edges == 0 || (gMv && gH), "aggregate_bwd: null tensor");
    EdgeBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.pv = plan_view(plan, n_atoms, n_edges);
    a.nV = (int)n_atoms; a.nE = (int)n_edges; a.h = (int)d_h;
    a.gin = gMv; a.ld_gin = ld_gmv; a.gZ = gH; a.ldgz = ld_gh;
    return launch_edge_bwd<EB_GATHER>(a, static_cast<hipStream_t>(stream), "k_edge_bwd<gather>");
}

size_t dmpnn_linear_wgrad_ws_bytes(int64_t M, int64_t N, int64_t K, int has_bias) {
    const WgradPlan p = plan_wgrad(M, (int)N, (int)K + (has_bias ? 1 : 0));
    return (align_up((size_t)p.max_splits() * p.slab_stride, 64) + p.split_floats) * sizeof(float);  // slabs | split operands (f16 pipe)
}

/* gW[N, K1+K2] = gZ^T . [A1[gather] || A2],  gb[N] = colsum(gZ)   (either output may be NULL) */
int dmpnn_linear_wgrad(const dmpnn_gemm_args* g, const float* gZ, int64_t ldgz, float* gW, int64_t ldgw, float* gb,
                       void* ws, size_t ws_bytes, void* stream) {
    DMPNN_CHECK_ARG(g && gZ, "linear_wgrad: null args");
    DMPNN_CHECK_ARG(g->M >= 0 && g->N > 0 && g->K1 + g->K2 > 0, "linear_wgrad: bad sizes");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int K = (int)(g->K1 + g->K2);
    const int ones = gb ? 1 : 0;
    const WgradPlan p = plan_wgrad(g->M, (int)g->N, K + ones);
    const size_t slab_floats = align_up((size_t)p.max_splits() * p.slab_stride, 64);
    if (ws_bytes < (slab_floats + p.split_floats) * sizeof(float)) {
        set_error("linear_wgrad: workspace too small");
        return DMPNN_ENOSPC;
    }
    if (g->M == 0) {
        if (gW) for (int64_t n = 0; n < g->N; ++n) hipMemsetAsync(gW + n * ldgw, 0, (size_t)K * sizeof(float), s);
        if (gb) hipMemsetAsync(gb, 0, (size_t)g->N * sizeof(float), s);
        return DMPNN_OK;
    }
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.M = g->M; a.N = (int)g->N; a.K1 = (int)g->K1; a.K2 = (int)g->K2; a.ones = ones;
    a.gZ = gZ; a.ldz = ldgz;
    a.A1 = g->A1; a.lda1 = g->lda1; a.gather1 = g->gather1;
    a.A2 = g->A2; a.lda2 = g->lda2;
    int ns = 0;
    DMPNN_TRY(launch_wgrad(a, p, static_cast<float*>(ws), s, p.split_floats ? static_cast<float*>(ws) + slab_floats : nullptr, &ns));
    return launch_wgrad_reduce(static_cast<float*>(ws), p, ns, (int)g->N, K, ones, gW, ldgw, gb, s);
}

int dmpnn_backward(const dmpnn_bwd_args* b, void* stream) { return backward_impl(b, stream, nullptr, nullptr); }

}  // extern "C"

namespace dmpnn {
size_t extra_wgrad_ws_floats(int64_t M, int N, int Kt) {
    const WProdPlan q = plan_wgrad16(M, N, Kt);
    const size_t blocks = align_up((wsplit16_bytes(M, N) + 3) / 4, 64) + align_up((wsplit16_bytes(M, Kt) + 3) / 4, 64) + align_up((size_t)q.splits * q.slab_stride, 64);
    // ... or, on the split-row path (k_rows2sr + k_wgrad16r): both operands as split rows, the product's slabs and the column sums'
    const WProdRPlan r = plan_wgrad16r(M, N, Kt), rb = plan_wgrad16r(M, N, 1);
    const auto row_bytes = [](int64_t K) { return (size_t)(((K + 31) / 32) * 128 + 16); };
    const size_t rows = (align_up((size_t)M * row_bytes(N), 256) + align_up((size_t)M * row_bytes(Kt), 256)) / 4 + align_up((size_t)r.splits * r.slab_stride, 64) +
                        align_up((size_t)rb.splits * rb.slab_stride, 64) + 64;
    return blocks > rows ? blocks : rows;
}

int backward_impl(const dmpnn_bwd_args* b, void* stream, const ExtraWgrad* extra, bool* extra_done) {
    if (extra_done) *extra_done = false;
    DMPNN_CHECK_ARG(b != nullptr, "backward: null args");
    const dmpnn_fwd_args& f = b->f;
    const int64_t nV = f.n_atoms, nE = f.n_edges, h = f.d_h, dv = f.d_v, de = f.d_e;
    const bool has_vd = f.W_d != nullptr;
    const int64_t dvd = has_vd ? f.d_vd : 0;
    const int T = f.depth;
    // atom messages (DMPNN_F_ATOM, base.py:254-289): gW_i is [h, d_v], gW_h [h, h + d_e]; only the tile kernels carry them
    const bool atom = (f.flags & DMPNN_F_ATOM) != 0;
    const int64_t de_i = atom ? 0 : de, de_h = atom ? de : 0;
    DMPNN_CHECK_ARG(f.plan && h > 0 && dv > 0 && T >= 1, "backward: bad forward description");
    DMPNN_CHECK_ARG(f.act >= DMPNN_ACT_RELU && f.act <= DMPNN_ACT_ELU && f.act != DMPNN_ACT_PRELU,
                    "backward: activation %d has no fused backward (use the row kernels)", f.act);
    DMPNN_CHECK_ARG(nV == 0 || b->gout, "backward: null gout");
    const BwdLayout L = bwd_layout(f);
    DMPNN_CHECK_ARG(L.lean || T == 1 || nE == 0 || (f.n_hslots >= T - 1 && f.n_mslots >= T - 1),
                    "backward: the forward did not keep every H^(t) / M^(t) (n_hslots, n_mslots must be depth-1)");
    if (b->ws_bytes < L.total * sizeof(float) || !b->ws) {
        set_error("backward: workspace too small (%zu < %zu bytes)", b->ws_bytes, L.total * sizeof(float));
        return DMPNN_ENOSPC;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* ws = b->ws;
    float *gZa = ws + L.gZa, *gZb = ws + L.gZb, *gH0 = ws + L.gH0, *gZO = ws + L.gZO, *gMv = ws + L.gMv;
    float *gHO = ws + L.gHO, *WhT = ws + L.WhT, *WoT = ws + L.WoT, *WdT = ws + L.WdT;
    float *slab_h = ws + L.slab_h, *slab_x = ws + L.slab_x;
    // fused forward: the kept edge tensors (H0, H^(t), M^(t)) are in CSR-row order -> the graph in row coordinates
    const bool fused = f.flags & DMPNN_F_FUSED;
    // tile plan (DMPNN_F_TILE_PLAN): the kept tensors are in the caller's edge order, there are no CSR tables — only the tile
    // kernel + the f16 weight-gradient products below can run, with the caller's own src array as the gather of W_i's operand
    const bool lean = (f.flags & DMPNN_F_TILE_PLAN) != 0;
    DMPNN_CHECK_ARG(!lean || (fused && (f.flags & DMPNN_F_MEGA) && (f.flags & DMPNN_F_SPLIT16) && !has_vd && (nE == 0 || (f.edge_index && f.rev_edge_index))),
                    "backward: DMPNN_F_TILE_PLAN needs the tile-kernel forward (FUSED | MEGA | SPLIT16), no W_d, and the caller's index arrays");
    const PlanView pv = fused ? plan_view_rows(f.plan, nV, nE) : plan_view(f.plan, nV, nE);
    const int* pflags = fused ? static_cast<const int*>(f.plan) + DMPNN_HDR_FLAGS : nullptr;
    const int pmask = lean ? kPlanNoMegaLean : (fused ? kPlanNoFuse : 0);
    const int* e_gather = (fused && !lean) ? static_cast<const int*>(f.plan) + plan_layout(nV, nE).perm : nullptr;
    DMPNN_CHECK_ARG(!fused || !(f.flags & DMPNN_F_UNDIRECTED), "backward: fused + undirected is not a valid forward");
    DMPNN_CHECK_ARG(!fused || T == 1 || nE == 0 || f.Hs || L.lean, "backward: the fused forward did not keep H^(t) (Hs was NULL)");
    const int64_t ldh = f.ldh, slot = nE * ldh;

    auto zero2d = [&](float* p, int64_t rows, int64_t cols) {
        if (p && rows * cols > 0) hipMemsetAsync(p, 0, (size_t)(rows * cols) * sizeof(float), s);
    };
    if (nV == 0) {
        zero2d(b->gW_i, h, dv + de_i); zero2d(b->gb_i, 1, h); zero2d(b->gW_h, h, h + de_h); zero2d(b->gb_h, 1, h);
        zero2d(b->gW_o, h, dv + h); zero2d(b->gb_o, 1, h); zero2d(b->gW_d, h + dvd, h + dvd); zero2d(b->gb_d, 1, h + dvd);
        return DMPNN_OK;
    }

    // ---- finalize backward (base.py:180-194) ----
    const float* gHO_p = b->gout;
    int64_t ld_gHO = b->ldgout;
    const float* HO = f.out;
    int64_t ldHO = f.ldout;
    if (has_vd) {
        HO = f.Hv; ldHO = ldh;
        if (b->gW_d || b->gb_d) {
            WgradArgs a;
            memset(&a, 0, sizeof(a));
            a.M = nV; a.N = (int)(h + dvd); a.K1 = (int)h; a.K2 = (int)dvd; a.ones = 1;
            a.gZ = b->gout; a.ldz = b->ldgout;
            a.A1 = f.Hv; a.lda1 = ldh; a.A2 = f.V_d; a.lda2 = f.ldvd;
            int ns = 0;
            DMPNN_TRY(launch_wgrad(a, L.p_d, slab_x, s, ws + L.w16g, &ns));
            DMPNN_TRY(launch_wgrad_reduce(slab_x, L.p_d, ns, (int)(h + dvd), (int)(h + dvd), 1, b->gW_d, h + dvd, b->gb_d, s, pflags, pmask));
        }
        // gHO = gout . W_d[:, :h]      via WdT[k][n] = W_d[n][k]
        DMPNN_TRY(launch_transpose(f.W_d, h + dvd, WdT, h + dvd, (int)(h + dvd), (int)h, s));
        dmpnn_gemm_args g;
        memset(&g, 0, sizeof(g));
        g.M = nV; g.N = h; g.K1 = h + dvd; g.A1 = b->gout; g.lda1 = b->ldgout;
        g.W = WdT; g.ldw = h + dvd; g.C = gHO; g.ldc = ldh; g.act = DMPNN_ACT_NONE;
        DMPNN_TRY(launch_linear(g, s));
        gHO_p = gHO; ld_gHO = ldh;
    }
    if (L.lean) {
        // ---- the per-step fused route's lean training forward: backward STEP kernels over the same tiles (dmpnn_bstep16.hip) ----
        DMPNN_CHECK_ARG(fused && !lean && !has_vd && T >= 2 && T - 1 <= kWProdMaxJobs && T <= kWProdMaxJobs,
                        "backward(lean fused16): depth 2 .. %d, no W_d", kWProdMaxJobs);
        const bool vec = h % 4 == 0 && ld_gHO % 4 == 0 && ldHO % 4 == 0 && ldh % 4 == 0 && aligned16(gHO_p) && aligned16(HO) && aligned16(gZO);
        {   // gZO = gout * tau'(out)
            const int64_t n = nV * (vec ? h / 4 : h);
            int64_t blocks = (n + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            if (vec) hipLaunchKernelGGL(k_act_bwd<4>, dim3((unsigned)blocks), dim3(256), 0, s, gHO_p, ld_gHO, HO, ldHO, gZO, ldh, nV, (int)h,
                                        f.act, f.act_slope, f.act_slope_ptr);
            else hipLaunchKernelGGL(k_act_bwd<1>, dim3((unsigned)blocks), dim3(256), 0, s, gHO_p, ld_gHO, HO, ldHO, gZO, ldh, nV, (int)h,
                                    f.act, f.act_slope, f.act_slope_ptr);
            DMPNN_CHECK_LAUNCH("k_act_bwd");
        }
        if (b->gW_o || b->gb_o) {
            WgradArgs a;
            memset(&a, 0, sizeof(a));
            a.M = nV; a.N = (int)h; a.K1 = (int)dv; a.K2 = (int)h; a.ones = 1;
            a.gZ = gZO; a.ldz = ldh; a.A1 = f.V; a.lda1 = f.ldv; a.A2 = f.Mv; a.lda2 = ldh;
            int ns = 0;
            DMPNN_TRY(launch_wgrad(a, L.p_o, slab_x, s, ws + L.w16g, &ns));
            DMPNN_TRY(launch_wgrad_reduce(slab_x, L.p_o, ns, (int)h, (int)(dv + h), 1, b->gW_o, dv + h, b->gb_o, s, pflags, pmask));
        }
        if (!(b->gW_i || b->gb_i || b->gW_h || b->gb_h)) return DMPNN_OK;
        {   // gMv = gZO . W_o[:, d_v:]
            dmpnn_gemm_args g;
            memset(&g, 0, sizeof(g));
            g.M = nV; g.N = h; g.K1 = h; g.A1 = gZO; g.lda1 = ldh; g.W = WoT; g.ldw = h; g.C = gMv; g.ldc = ldh;
            g.act = DMPNN_ACT_NONE;
            if (L.use16 && linear16_ok(g)) {
                SplitWView w;
                DMPNN_TRY(split_weights_view(f.W_o + dv, dv + h, h, h, 1, ws + L.WoT16, &w, s));
                DMPNN_TRY(launch_linear16_view(g, w, nullptr, 0, s));
            } else {
                DMPNN_TRY(launch_transpose(f.W_o + dv, dv + h, WoT, h, (int)h, (int)h, s));
                DMPNN_TRY(launch_linear(g, s));
            }
        }
        DMPNN_CHECK_ARG(L.use16, "backward(lean fused16): even d_h");
        SplitWView whT;
        DMPNN_TRY(split_weights_view(f.W_h, h, h, h, 1, ws + L.WhT16, &whT, s));
        unsigned char* Zb = reinterpret_cast<unsigned char*>(ws + L.lz);
        const size_t zs = L.lz_stride * sizeof(float);
        float* Tb[2] = {gZa, gZb};
        // site T-1 (gather from gMv) ... site 1: gZ^(t) rows + T_next; site 0: gZ^(0) rows only
        for (int t = T - 1; t >= 1; --t)
            DMPNN_TRY(launch_bstep16(f, t, t == T - 1 ? nullptr : Tb[t & 1], gMv, &whT, Tb[(t - 1) & 1], Zb + (size_t)t * zs, s));
        DMPNN_TRY(launch_bstep16(f, 0, Tb[0], gMv, nullptr, nullptr, Zb, s));
        // the products' other operands are what the forward kept: M^(t) (slot t - 1 of msplit), x (in H0) — split rows, read as they are.
        // gW_h = sum_t gZ^(t)^T M^(t), gW_i = sum_t gZ^(t)^T x, the bias gradients = column sums of the gZ^(t): ALL jobs in one launch
        const int ts_m = (int)(split_row_floats(h) * 4), ts_x = ((int)((dv + de + 31) / 32)) * 128 + 16;
        const unsigned char* Mk = static_cast<const unsigned char*>(f.msplit);
        ReduceJobs rj;
        memset(&rj, 0, sizeof(rj));
        rj.poison_flags = pflags; rj.poison_mask = pmask;
        auto add_reduce_r = [&](float* slab, const WProdRPlan& q, int n_jobs, int K, int ones, float* gW, int64_t ldgw, float* gb) {
            ReduceJob& r = rj.job[rj.n_jobs];
            r.slab = slab; r.slab_stride = q.slab_stride; r.n_slabs = n_jobs * q.splits; r.ldk = q.ldk;
            r.N = (int)h; r.K = K; r.ones = ones; r.gW = gW; r.ldgw = ldgw; r.gb = gb;
            int64_t blocks = ((int64_t)r.N * (K + ones) + 255) / 256;
            if (blocks > 1024) blocks = 1024;
            rj.wg0[rj.n_jobs + 1] = rj.wg0[rj.n_jobs] + (int)blocks;
            ++rj.n_jobs;
        };
        WProdRJob pj[4 * kWProdMaxJobs];
        int np = 0;
        float* slab = ws + L.lslab;
        // the products of steps t0 .. t1 - 1 (gW = sum_t gZ^(t)^T A^(t)); gb = sum_t colsum(gZ^(t)) rides in them (or is a job set of its own
        // when the weight's gradient is not wanted)
        auto add_jobs = [&](int t0, int t1, const unsigned char* A, size_t a_stride, int tsa, int K, const WProdRPlan& q, const WProdRPlan& qb,
                            float* gW, int64_t ldgw, float* gb) {
            if (!gW && !gb) return;
            float* sl = slab;
            if (gW) slab += align_up((size_t)(t1 - t0) * q.splits * q.slab_stride, 64);
            float* slb = slab;
            if (gb) slab += align_up((size_t)(t1 - t0) * qb.splits * qb.slab_stride, 64);
            for (int t = t0; t < t1; ++t) {
                float* sb_t = gb ? slb + (size_t)(t - t0) * qb.splits * qb.slab_stride : nullptr;
                if (gW) pj[np++] = WProdRJob{Zb + (size_t)t * zs, ts_m, A + (size_t)(t - t0) * a_stride, tsa, nE, (int)h, K, sl + (size_t)(t - t0) * q.splits * q.slab_stride, q, sb_t};
                else pj[np++] = WProdRJob{Zb + (size_t)t * zs, ts_m, nullptr, 0, nE, (int)h, 1, sb_t, qb, nullptr};
            }
            if (gW) add_reduce_r(sl, q, t1 - t0, K, 0, gW, ldgw, nullptr);
            if (gb) add_reduce_r(slb, qb, t1 - t0, 0, 1, nullptr, 0, gb);
        };
        add_jobs(1, T, Mk, (size_t)nE * ts_m, ts_m, (int)h, L.rh, L.rbh, b->gW_h, h, f.b_h ? b->gb_h : nullptr);
        add_jobs(0, T, reinterpret_cast<const unsigned char*>(f.H0), 0, ts_x, (int)(dv + de), L.ri, L.rbi, b->gW_i, dv + de, f.b_i ? b->gb_i : nullptr);
        DMPNN_TRY(launch_wgrad16r(pj, np, s));
        if (rj.n_jobs > 0) {
            hipLaunchKernelGGL(k_wgrad_reduce_multi, dim3((unsigned)rj.wg0[rj.n_jobs]), dim3(256), 0, s, rj);
            DMPNN_CHECK_LAUNCH("k_wgrad_reduce_multi");
        }
        return DMPNN_OK;
    }
    const bool tile_bwd = L.mega && (atom || L.sr || b->g_edge || b->gW_i || b->gb_i || b->gW_h || b->gb_h) && ld_gHO % 4 == 0 && ldHO % 4 == 0 && aligned16(gHO_p) && aligned16(HO);
    DMPNN_CHECK_ARG(!b->g_edge || nE == 0 || (tile_bwd && b->ld_gedge >= h && b->ld_gedge % 4 == 0 && aligned16(b->g_edge)),
                    "backward: g_edge (a gradient w.r.t. the kept H^(depth-1)) is taken by the backward tile kernel only (tile-kernel forward with "
                    "DMPNN_F_KEEP), as 16-byte aligned rows with a leading dimension that is a multiple of 4");
    DMPNN_CHECK_ARG(!mega16_keeps_rows(f) || (tile_bwd && L.sr) || !(b->gW_h || b->gb_h),
                    "backward: the forward kept its messages as split rows (`msplit`): the weight gradient of W_h then needs the tile kernels — "
                    "16-byte aligned gout / out with leading dimensions that are multiples of 4, d_v + d_e and d_v + d_h <= 512");
    DMPNN_CHECK_ARG(!atom || (tile_bwd && L.w16 && !has_vd && (T < 2 || nE == 0 || f.msplit)),
                    "backward: DMPNN_F_ATOM needs the tile-kernel forward (FUSED | MEGA | SPLIT16 | KEEP with `msplit`), even d_v / d_e / d_h and "
                    "16-byte aligned gout / out (leading dimensions multiples of 4)");
    // in-kernel dropout (dmpnn_fwd_args.dropout_p): its 1 / (1 - p) lives in the backward TILE kernel alone — every other branch below
    // would return gradients without it, silently
    DMPNN_CHECK_ARG(!(f.dropout_p > 0.f) || tile_bwd,
                    "backward: the forward ran with dropout inside the kernels; only the backward tile kernel carries its scale — it needs a "
                    "gradient of W_i or W_h to be wanted and 16-byte aligned gout / out (leading dimensions multiples of 4)");
    if (tile_bwd) {
        // ---- the whole data-gradient chain in one launch, then the four weight gradients ----
        DMPNN_CHECK_ARG(f.H0 && (T == 1 || f.Hs), "backward: the tile-kernel forward did not keep H0 / H^(t)");
        float* gZs = (T - 1 > 2) ? ws + L.gZs : gZa;  // slot t-1 = gZ^(t)
        Mega16BwdRows br{nullptr, nullptr, nullptr, 0};
        if (L.sr) br = Mega16BwdRows{reinterpret_cast<unsigned char*>(ws + L.sr_gz), reinterpret_cast<unsigned char*>(ws + L.sr_gh0),
                                     reinterpret_cast<unsigned char*>(ws + L.sr_gzo), L.tsr};
        DMPNN_TRY(launch_mega16_backward(f, gHO_p, ld_gHO, HO, ldHO, gZO, gZs, gH0, ws + L.mega_w, ws + L.sp_gM, ws + L.gMv, s, b->g_edge, b->ld_gedge, &br));
        if (L.sr) {
            // ---- every weight-gradient operand is a set of split rows: the gradients left the tile kernel so, M^(t) the forward; the two
            // input-side operands are split here (one launch, + a rider's), then ALL products in one launch, one reduce launch ----
            unsigned char* Xr = reinterpret_cast<unsigned char*>(ws + L.sr_x);
            unsigned char* VMr = reinterpret_cast<unsigned char*>(ws + L.sr_vm);
            const bool want_o = b->gW_o || b->gb_o, want_h = (b->gW_h || b->gb_h) && T >= 2, want_i = b->gW_i || b->gb_i;
            const bool ride = extra && extra->Z && extra->A && extra->ws && (extra->gW || extra->gb) && extra->M > 0 && extra->N <= 320 && extra->K <= 512 &&
                              aligned16(extra->ws);
            const auto row_bytes = [](int64_t K) { return (int)(((K + 31) / 32) * 128 + 16); };
            SRJob sj[4];
            int ns = 0;
            if (want_i) {
                sj[ns] = SRJob{f.V, f.ldv, (int)dv, lean ? nullptr : pv.src, lean ? reinterpret_cast<const long long*>(f.edge_index) : nullptr, nV,
                               f.E, f.lde, (int)de, e_gather, nE, Xr, L.tsx};
                ++ns;
            }
            if (want_o) { sj[ns] = SRJob{f.V, f.ldv, (int)dv, nullptr, nullptr, 0, f.Mv, ldh, (int)h, nullptr, nV, VMr, L.tsv}; ++ns; }
            unsigned char *rZ = nullptr, *rA = nullptr;
            float* rslab = nullptr;
            WProdRPlan q_x, q_xb;
            int ts_rz = 0, ts_ra = 0;
            if (ride) {
                ts_rz = row_bytes(extra->N); ts_ra = row_bytes(extra->K);
                rZ = reinterpret_cast<unsigned char*>(extra->ws);
                rA = rZ + align_up((size_t)extra->M * ts_rz, 256);
                rslab = reinterpret_cast<float*>(rA + align_up((size_t)extra->M * ts_ra, 256));
                // (the launch's row count per workgroup, but never more slabs than extra_wgrad_ws_floats sized the rider's workspace for)
                const int r_x = plan_wgrad16r(extra->M, extra->N, extra->K).rows_per_split;
                q_x = plan_wgrad16r(extra->M, extra->N, extra->K, L.sr_R > r_x ? L.sr_R : r_x);
                q_xb = plan_wgrad16r(extra->M, extra->N, 1, q_x.rows_per_split);
                sj[ns] = SRJob{extra->Z, extra->ldz, extra->N, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, extra->M, rZ, ts_rz}; ++ns;
                sj[ns] = SRJob{extra->A, extra->lda, extra->K, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, extra->M, rA, ts_ra}; ++ns;
            }
            DMPNN_TRY(launch_rows2sr(sj, ns, s));
            ReduceJobs rj;
            memset(&rj, 0, sizeof(rj));
            rj.poison_flags = pflags; rj.poison_mask = pmask;
            WProdRJob pj[8];
            int np = 0;
            float* slab = ws + L.sr_slab;
            auto add_reduce = [&](float* sl, const WProdRPlan& q, int N, int K, int ones, float* gW, int64_t ldgw, float* gb) {
                ReduceJob& r = rj.job[rj.n_jobs];
                r.slab = sl; r.slab_stride = q.slab_stride; r.n_slabs = q.splits; r.ldk = q.ldk;
                r.N = N; r.K = K; r.ones = ones; r.gW = gW; r.ldgw = ldgw; r.gb = gb;
                int64_t blocks = ((int64_t)N * (ones ? 1 : K) + 255) / 256;
                if (blocks > 1024) blocks = 1024;
                rj.wg0[rj.n_jobs + 1] = rj.wg0[rj.n_jobs] + (int)blocks;
                ++rj.n_jobs;
            };
            // gW = Z^T A into q's slabs at sl (gW null: not wanted); gb = colsum(Z) into qb's slabs at slb — riding in the product's
            // workgroups (qb has the product's rows), or a column-sum job of its own when there is no product
            auto add = [&](const unsigned char* Z, int tsz, const unsigned char* A, int tsa, int64_t M, int N, int K, const WProdRPlan& q, float* sl,
                           float* gW, int64_t ldgw, const WProdRPlan& qb, float* slb, float* gb) {
                if (gW) {
                    pj[np++] = WProdRJob{Z, tsz, A, tsa, M, N, K, sl, q, gb ? slb : nullptr};
                    add_reduce(sl, q, N, K, 0, gW, ldgw, nullptr);
                } else if (gb) {
                    pj[np++] = WProdRJob{Z, tsz, nullptr, 0, M, N, 1, slb, qb, nullptr};
                }
                if (gb) add_reduce(slb, qb, N, 0, 1, nullptr, 0, gb);
            };
            auto next_slab = [&](const WProdRPlan& q) { float* p = slab; slab += align_up((size_t)q.splits * q.slab_stride, 64); return p; };
            const unsigned char* gZr = br.gZ;
            const unsigned char* Mr = static_cast<const unsigned char*>(f.msplit);
            // (the big product first: the small ones fill its tail)
            if (want_h) {
                float* sl = next_slab(L.r_h); float* slb = next_slab(L.r_bh);
                add(gZr, L.tsr, Mr, L.tsr, nE * (T - 1), (int)h, (int)h, L.r_h, sl, b->gW_h, h, L.r_bh, slb, f.b_h ? b->gb_h : nullptr);
            }
            if (want_o) {
                float* sl = next_slab(L.r_o); float* slb = next_slab(L.r_bo);
                add(br.gZO, L.tsr, VMr, L.tsv, nV, (int)h, (int)(dv + h), L.r_o, sl, b->gW_o, dv + h, L.r_bo, slb, b->gb_o);
            }
            if (want_i) {
                float* sl = next_slab(L.r_i); float* slb = next_slab(L.r_bi);
                add(br.gH0, L.tsr, Xr, L.tsx, nE, (int)h, (int)(dv + de), L.r_i, sl, b->gW_i, dv + de, L.r_bi, slb, f.b_i ? b->gb_i : nullptr);
            }
            if (ride) {
                float* sl2 = rslab + align_up((size_t)q_x.splits * q_x.slab_stride, 64);
                add(rZ, ts_rz, rA, ts_ra, extra->M, extra->N, extra->K, q_x, rslab, extra->gW, extra->ldgw, q_xb, sl2, extra->ones ? extra->gb : nullptr);
                if (extra_done) *extra_done = true;
            }
            DMPNN_TRY(launch_wgrad16r(pj, np, s));
            if (rj.n_jobs > 0) {
                hipLaunchKernelGGL(k_wgrad_reduce_multi, dim3((unsigned)rj.wg0[rj.n_jobs]), dim3(256), 0, s, rj);
                DMPNN_CHECK_LAUNCH("k_wgrad_reduce_multi");
            }
            if ((b->gW_h || b->gb_h) && T < 2) { zero2d(b->gW_h, h, h); zero2d(b->gb_h, 1, h); }
            if (!f.b_h) zero2d(b->gb_h, 1, h);
            if (!f.b_i) zero2d(b->gb_i, 1, h);
            return DMPNN_OK;
        }
        if (L.w16 && aligned16(f.Mv) && (T < 2 || (aligned16(f.Ms) && aligned16(gZs))) && aligned16(gH0) && aligned16(gZO)) {
            // ---- the three weight gradients on the f16 pipe: every operand split ONCE (one launch), three products, three reduces ----
            const bool want[3] = {b->gW_o || b->gb_o, (b->gW_h || b->gb_h) && T >= 2, b->gW_i || b->gb_i};
            WSplitArgs sp;
            memset(&sp, 0, sizeof(sp));
            WSplitJob* Zj[3] = {nullptr, nullptr, nullptr};
            WSplitJob* Aj[3] = {nullptr, nullptr, nullptr};
            const int Ks[3] = {(int)(dv + h) + 1, (int)(h + de_h) + (f.b_h ? 1 : 0), (int)(dv + de_i) + (f.b_i ? 1 : 0)};
            if (want[0]) {
                Zj[0] = &sp.job[sp.n_jobs++]; wsplit16_job(Zj[0], nV, (int)h, gZO, ldh, nullptr, (int)h, nullptr, 0, nullptr, 0, 0, ws + L.w16_z[0]);
                Aj[0] = &sp.job[sp.n_jobs++]; wsplit16_job(Aj[0], nV, Ks[0], f.V, f.ldv, nullptr, (int)dv, f.Mv, ldh, nullptr, (int)h, 1, ws + L.w16_a[0]);
            }
            if (want[1]) {
                Zj[1] = &sp.job[sp.n_jobs++]; wsplit16_job(Zj[1], nE * (T - 1), (int)h, gZs, ldh, nullptr, (int)h, nullptr, 0, nullptr, 0, 0, ws + L.w16_z[1]);
                // (atom messages: the bond-feature half of every step's message, the forward's [depth - 1][n_edges][16] rows in `msplit`)
                Aj[1] = &sp.job[sp.n_jobs++]; wsplit16_job(Aj[1], nE * (T - 1), Ks[1], f.Ms, ldh, nullptr, (int)h, static_cast<const float*>(f.msplit), 16, nullptr, (int)de_h,
                                                           f.b_h ? 1 : 0, ws + L.w16_a[1]);
            }
            if (want[2]) {
                Zj[2] = &sp.job[sp.n_jobs++]; wsplit16_job(Zj[2], nE, (int)h, gH0, ldh, nullptr, (int)h, nullptr, 0, nullptr, 0, 0, ws + L.w16_z[2]);
                Aj[2] = &sp.job[sp.n_jobs++]; wsplit16_job(Aj[2], nE, Ks[2], f.V, f.ldv, lean ? nullptr : pv.src, (int)dv, f.E, f.lde, e_gather, (int)de_i, f.b_i ? 1 : 0, ws + L.w16_a[2]);
                if (lean) { Aj[2]->g1_64 = reinterpret_cast<const long long*>(f.edge_index); Aj[2]->g1_rows = nV; }  // (row 0 of edge_index: src)
            }
            // the rider (see ExtraWgrad): two more operands to split, one more product, one more reduce job
            WSplitJob *Zx = nullptr, *Ax = nullptr;
            WProdPlan qx;
            float* slab_x16 = nullptr;
            const bool ride = extra && extra->Z && extra->A && extra->ws && (extra->gW || extra->gb) && extra->M > 0 && extra->N % 2 == 0 &&
                              wgrad16_operand_ok(extra->Z, extra->ldz, extra->N, nullptr, 0, 0) && wgrad16_operand_ok(extra->A, extra->lda, extra->K, nullptr, 0, 0) &&
                              aligned16(extra->ws) && sp.n_jobs + 2 <= 8;
            if (ride) {
                const int Ktx = extra->K + extra->ones;
                qx = plan_wgrad16(extra->M, extra->N, Ktx);
                float* wz = extra->ws;
                float* wa = wz + align_up((wsplit16_bytes(extra->M, extra->N) + 3) / 4, 64);
                slab_x16 = wa + align_up((wsplit16_bytes(extra->M, Ktx) + 3) / 4, 64);
                Zx = &sp.job[sp.n_jobs++]; wsplit16_job(Zx, extra->M, extra->N, extra->Z, extra->ldz, nullptr, extra->N, nullptr, 0, nullptr, 0, 0, wz);
                Ax = &sp.job[sp.n_jobs++]; wsplit16_job(Ax, extra->M, Ktx, extra->A, extra->lda, nullptr, extra->K, nullptr, 0, nullptr, 0, extra->ones, wa);
            }
            DMPNN_TRY(launch_wsplit16(sp, s));
            float* gWs[3] = {b->gW_o, b->gW_h, b->gW_i};
            float* gbs[3] = {b->gb_o, b->gb_h, b->gb_i};
            const int64_t ldg[3] = {dv + h, h + de_h, dv + de_i};
            const int ones[3] = {1, f.b_h ? 1 : 0, f.b_i ? 1 : 0};
            ReduceJobs rj;
            memset(&rj, 0, sizeof(rj));
            rj.poison_flags = pflags; rj.poison_mask = pmask;
            WProdJobs pj;
            memset(&pj, 0, sizeof(pj));
            for (int i = 0; i < 3; ++i)  // (the big product first: the small ones fill its tail)
                if (want[(i + 1) % 3]) wgrad16_add(&pj, *Zj[(i + 1) % 3], *Aj[(i + 1) % 3], L.q[(i + 1) % 3], (int)h, Ks[(i + 1) % 3], ws + L.w16_slab[(i + 1) % 3]);
            if (ride) wgrad16_add(&pj, *Zx, *Ax, qx, extra->N, extra->K + extra->ones, slab_x16);
            DMPNN_TRY(launch_wgrad16(pj, s));  // the three products (+ the rider) in one launch
            for (int i = 0; i < 3; ++i) {
                if (!want[i]) continue;
                ReduceJob& r = rj.job[rj.n_jobs];
                r.slab = ws + L.w16_slab[i]; r.slab_stride = L.q[i].slab_stride; r.n_slabs = L.q[i].splits; r.ldk = L.q[i].ldk;
                r.N = (int)h; r.K = Ks[i] - ones[i]; r.ones = ones[i]; r.gW = gWs[i]; r.ldgw = ldg[i]; r.gb = gbs[i];
                int64_t blocks = ((int64_t)r.N * Ks[i] + 255) / 256;
                if (blocks > 1024) blocks = 1024;
                rj.wg0[rj.n_jobs + 1] = rj.wg0[rj.n_jobs] + (int)blocks;
                ++rj.n_jobs;
            }
            if (ride && rj.n_jobs < 4) {
                ReduceJob& r = rj.job[rj.n_jobs];
                r.slab = slab_x16; r.slab_stride = qx.slab_stride; r.n_slabs = qx.splits; r.ldk = qx.ldk;
                r.N = extra->N; r.K = extra->K; r.ones = extra->ones; r.gW = extra->gW; r.ldgw = extra->ldgw; r.gb = extra->gb;
                int64_t blocks = ((int64_t)r.N * (extra->K + extra->ones) + 255) / 256;
                if (blocks > 1024) blocks = 1024;
                rj.wg0[rj.n_jobs + 1] = rj.wg0[rj.n_jobs] + (int)blocks;
                ++rj.n_jobs;
                if (extra_done) *extra_done = true;
            }
            if (rj.n_jobs > 0) {  // one reduce launch for all of them
                hipLaunchKernelGGL(k_wgrad_reduce_multi, dim3((unsigned)rj.wg0[rj.n_jobs]), dim3(256), 0, s, rj);
                DMPNN_CHECK_LAUNCH("k_wgrad_reduce_multi");
            }
            if ((b->gW_h || b->gb_h) && T < 2) { zero2d(b->gW_h, h, h + de_h); zero2d(b->gb_h, 1, h); }
            return DMPNN_OK;
        }
        DMPNN_CHECK_ARG(!lean && !atom, "backward(DMPNN_F_TILE_PLAN / DMPNN_F_ATOM): the weight-gradient products on the f16 pipe do not take these shapes / alignments");
        if (b->gW_o || b->gb_o) {
            WgradArgs a;
            memset(&a, 0, sizeof(a));
            a.M = nV; a.N = (int)h; a.K1 = (int)dv; a.K2 = (int)h; a.ones = 1;
            a.gZ = gZO; a.ldz = ldh; a.A1 = f.V; a.lda1 = f.ldv; a.A2 = f.Mv; a.lda2 = ldh;
            int ns = 0;
            DMPNN_TRY(launch_wgrad(a, L.p_o, slab_x, s, ws + L.w16g, &ns));
            DMPNN_TRY(launch_wgrad_reduce(slab_x, L.p_o, ns, (int)h, (int)(dv + h), 1, b->gW_o, dv + h, b->gb_o, s, pflags, pmask));
        }
        if (b->gW_h || b->gb_h) {
            if (T >= 2) {
                // gW_h = sum_t gZ^(t)^T M^(t) = [gZ^(1); ...; gZ^(T-1)]^T [M^(1); ...; M^(T-1)]: the kept slots are adjacent
                // (row stride ldh, slot stride n_edges * ldh), so ONE product over (T - 1) n_edges rows — one launch instead of
                // T - 1 (each has a fixed cost of ~13 us at 9 120 rows: scripts/probe_wgrad.py)
                WgradArgs a;
                memset(&a, 0, sizeof(a));
                a.M = nE * (T - 1); a.N = (int)h; a.K1 = (int)h; a.K2 = 0; a.ones = f.b_h ? 1 : 0;
                a.gZ = gZs; a.ldz = ldh; a.A1 = f.Ms; a.lda1 = ldh;
                int ns = 0;
                DMPNN_TRY(launch_wgrad(a, L.p_hm, slab_h, s, ws + L.w16g, &ns));
                DMPNN_TRY(launch_wgrad_reduce(slab_h, L.p_hm, ns, (int)h, (int)h, f.b_h ? 1 : 0, b->gW_h, h, b->gb_h, s, pflags, pmask));
            } else {
                zero2d(b->gW_h, h, h); zero2d(b->gb_h, 1, h);
            }
        }
        if (b->gW_i || b->gb_i) {
            WgradArgs a;
            memset(&a, 0, sizeof(a));
            a.M = nE; a.N = (int)h; a.K1 = (int)dv; a.K2 = (int)de; a.ones = f.b_i ? 1 : 0;
            a.gZ = gH0; a.ldz = ldh; a.A1 = f.V; a.lda1 = f.ldv; a.gather1 = pv.src; a.A2 = f.E; a.lda2 = f.lde;
            a.gather2 = e_gather;
            int ns = 0;
            DMPNN_TRY(launch_wgrad(a, L.p_i, slab_x, s, ws + L.w16g, &ns));
            DMPNN_TRY(launch_wgrad_reduce(slab_x, L.p_i, ns, (int)h, (int)(dv + de), f.b_i ? 1 : 0, b->gW_i, dv + de, b->gb_i, s, pflags, pmask));
        }
        return DMPNN_OK;
    }
    DMPNN_CHECK_ARG(!lean, "backward(DMPNN_F_TILE_PLAN): only the tile-kernel backward can read a tile plan — it needs a gradient of W_i or "
                    "W_h to be wanted, 16-byte aligned gout / out with leading dimensions that are multiples of 4");
    {
        const bool vec = h % 4 == 0 && ld_gHO % 4 == 0 && ldHO % 4 == 0 && ldh % 4 == 0 && aligned16(gHO_p) && aligned16(HO) && aligned16(gZO);
        const int64_t n = nV * (vec ? h / 4 : h);
        int64_t blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (vec) hipLaunchKernelGGL(k_act_bwd<4>, dim3((unsigned)blocks), dim3(256), 0, s, gHO_p, ld_gHO, HO, ldHO, gZO, ldh, nV, (int)h,
                                    f.act, f.act_slope, f.act_slope_ptr);
        else hipLaunchKernelGGL(k_act_bwd<1>, dim3((unsigned)blocks), dim3(256), 0, s, gHO_p, ld_gHO, HO, ldHO, gZO, ldh, nV, (int)h,
                                f.act, f.act_slope, f.act_slope_ptr);
        DMPNN_CHECK_LAUNCH("k_act_bwd");
    }
    if (b->gW_o || b->gb_o) {
        WgradArgs a;
        memset(&a, 0, sizeof(a));
        a.M = nV; a.N = (int)h; a.K1 = (int)dv; a.K2 = (int)h; a.ones = 1;
        a.gZ = gZO; a.ldz = ldh; a.A1 = f.V; a.lda1 = f.ldv; a.A2 = f.Mv; a.lda2 = ldh;
        int ns = 0;
        DMPNN_TRY(launch_wgrad(a, L.p_o, slab_x, s, ws + L.w16g, &ns));
        DMPNN_TRY(launch_wgrad_reduce(slab_x, L.p_o, ns, (int)h, (int)(dv + h), 1, b->gW_o, dv + h, b->gb_o, s, pflags, pmask));
    }
    const bool need_edges = b->gW_i || b->gb_i || b->gW_h || b->gb_h;
    if (!need_edges) return DMPNN_OK;
    if (nE == 0) {
        zero2d(b->gW_i, h, dv + de); zero2d(b->gb_i, 1, h); zero2d(b->gW_h, h, h); zero2d(b->gb_h, 1, h);
        return DMPNN_OK;
    }
    // gMv = gZO . W_o[:, d_v:]
    {
        dmpnn_gemm_args g;
        memset(&g, 0, sizeof(g));
        g.M = nV; g.N = h; g.K1 = h; g.A1 = gZO; g.lda1 = ldh; g.W = WoT; g.ldw = h; g.C = gMv; g.ldc = ldh;
        g.act = DMPNN_ACT_NONE;
        if (L.use16 && linear16_ok(g)) {  // W'[n][k] = W_o[k][d_v + n]: the pre-split reads the matrix transposed
            SplitWView w;
            DMPNN_TRY(split_weights_view(f.W_o + dv, dv + h, h, h, 1, ws + L.WoT16, &w, s));
            DMPNN_TRY(launch_linear16_view(g, w, nullptr, 0, s));
        } else {
            DMPNN_TRY(launch_transpose(f.W_o + dv, dv + h, WoT, h, (int)h, (int)h, s));
            DMPNN_TRY(launch_linear(g, s));
        }
    }
    EdgeBwdArgs e;
    memset(&e, 0, sizeof(e));
    e.pv = pv; e.nV = (int)nV; e.nE = (int)nE; e.h = (int)h;
    e.act = f.act; e.slope = f.act_slope; e.slope_ptr = f.act_slope_ptr;
    e.poison_mask = fused ? kPlanNoFuse : PLAN_ASYMMETRIC;
    const bool undirected = f.flags & DMPNN_F_UNDIRECTED;
    int n_slabs_h = 0;
    if (T >= 2) {
        SplitWView wh16;
        bool gm16 = false;
        {
            dmpnn_gemm_args g;
            memset(&g, 0, sizeof(g));
            g.M = nE; g.N = h; g.K1 = h; g.A1 = gZa; g.lda1 = ldh; g.W = WhT; g.ldw = h; g.C = gZb; g.ldc = ldh;
            gm16 = L.use16 && linear16_ok(g);
        }
        if (gm16) DMPNN_TRY(split_weights_view(f.W_h, h, h, h, 1, ws + L.WhT16, &wh16, s));
        else DMPNN_TRY(launch_transpose(f.W_h, h, WhT, h, (int)h, (int)h, s));
        // gZ^(T-1) = gMv[dst] * tau'(H^(T-1));  gH0 = gZ^(T-1)
        EdgeBwdArgs g0 = e;
        g0.gin = gMv; g0.ld_gin = ldh;
        g0.Y = f.Hs + (int64_t)(T - 2) * slot; g0.ldy = ldh; g0.y_preact = 0;
        g0.gZ = gZa; g0.ldgz = ldh; g0.acc = gH0; g0.ldacc = ldh; g0.acc_init = 1;
        DMPNN_TRY(launch_edge_bwd<EB_GATHER>(g0, s, "k_edge_bwd<gather>"));
        float* gZ = gZa;
        float* other = gZb;
        for (int t = T - 1; t >= 1; --t) {
            const float* Mt = f.Ms + (int64_t)(t - 1) * slot;
            if (b->gW_h || b->gb_h) {
                WgradArgs a;
                memset(&a, 0, sizeof(a));
                a.M = nE; a.N = (int)h; a.K1 = (int)h; a.K2 = 0; a.ones = f.b_h ? 1 : 0;
                a.gZ = gZ; a.ldz = ldh; a.A1 = Mt; a.lda1 = ldh;
                int ns = 0;
                DMPNN_TRY(launch_wgrad(a, L.p_h, slab_h + (int64_t)n_slabs_h * L.p_h.slab_stride, s, ws + L.w16g, &ns));
                n_slabs_h += ns;
            }
            // gM = gZ . W_h
            dmpnn_gemm_args g;
            memset(&g, 0, sizeof(g));
            g.M = nE; g.N = h; g.K1 = h; g.A1 = gZ; g.lda1 = ldh; g.W = WhT; g.ldw = h; g.C = other; g.ldc = ldh;
            g.act = DMPNN_ACT_NONE;
            if (gm16) DMPNN_TRY(launch_linear16_view(g, wh16, nullptr, 0, s));
            else DMPNN_TRY(launch_linear(g, s));
            // gH^(t-1) -> masked gZ^(t-1), accumulated into gH0
            const bool first = (t - 1) == 0;
            const float* Yprev = first ? f.H0 : f.Hs + (int64_t)(t - 2) * slot;
            EdgeBwdArgs m = e;
            m.gin = other; m.ld_gin = ldh;
            if (!undirected) {
                m.Y = Yprev; m.ldy = ldh; m.y_preact = first ? 1 : 0;
                m.gZ = first ? nullptr : gZ; m.ldgz = ldh;
                m.acc = gH0; m.ldacc = ldh; m.acc_init = 0;
                DMPNN_TRY(launch_edge_bwd<EB_MESSAGE>(m, s, "k_edge_bwd<message>"));
            } else {
                m.gZ = gZ; m.ldgz = ldh;  // raw gHb
                DMPNN_TRY(launch_edge_bwd<EB_MESSAGE>(m, s, "k_edge_bwd<message>"));
                EdgeBwdArgs v = e;
                v.gin = gZ; v.ld_gin = ldh;
                v.Y = Yprev; v.ldy = ldh; v.y_preact = first ? 1 : 0;
                v.gZ = first ? nullptr : other; v.ldgz = ldh;
                v.acc = gH0; v.ldacc = ldh; v.acc_init = 0;
                DMPNN_TRY(launch_edge_bwd<EB_AVG>(v, s, "k_edge_bwd<avg>"));
                float* tmp = gZ; gZ = other; other = tmp;
            }
        }
        if (b->gW_h || b->gb_h)
            DMPNN_TRY(launch_wgrad_reduce(slab_h, L.p_h, n_slabs_h, (int)h, (int)h, f.b_h ? 1 : 0, b->gW_h, h, b->gb_h, s, pflags, pmask));
    } else {
        // depth 1: gH0 = gMv[dst] * tau'(tau(H0))
        EdgeBwdArgs g0 = e;
        g0.gin = gMv; g0.ld_gin = ldh;
        g0.Y = f.H0; g0.ldy = ldh; g0.y_preact = 1;
        g0.acc = gH0; g0.ldacc = ldh; g0.acc_init = 1;
        DMPNN_TRY(launch_edge_bwd<EB_GATHER>(g0, s, "k_edge_bwd<gather>"));
        zero2d(b->gW_h, h, h); zero2d(b->gb_h, 1, h);
    }
    if (b->gW_i || b->gb_i) {
        WgradArgs a;
        memset(&a, 0, sizeof(a));
        a.M = nE; a.N = (int)h; a.K1 = (int)dv; a.K2 = (int)de; a.ones = f.b_i ? 1 : 0;
        a.gZ = gH0; a.ldz = ldh; a.A1 = f.V; a.lda1 = f.ldv; a.gather1 = pv.src; a.A2 = f.E; a.lda2 = f.lde;
        a.gather2 = e_gather;
        int ns = 0;
        DMPNN_TRY(launch_wgrad(a, L.p_i, slab_x, s, ws + L.w16g, &ns));
        DMPNN_TRY(launch_wgrad_reduce(slab_x, L.p_i, ns, (int)h, (int)(dv + de), f.b_i ? 1 : 0, b->gW_i, dv + de, b->gb_i, s, pflags, pmask));
    }
    return DMPNN_OK;
}

}  // namespace dmpnn

