// K1 / K3 / K5 — the dense contractions of the path on fp32 MFMA.
//
//   C[r, :] = act( [A1[g(r), 0:K1] || A2[r, 0:K2]] . W^T + bias + Cadd[r, :] )
//
//   K1 initialize  mixins.py:8-9    A1 = V gathered by src(e), A2 = E          W = W_i  (no act: H0)
//   K3 update      base.py:135-141  A1 = M                    Cadd = H0        W = W_h  act = tau
//   K5 finalize    base.py:180-183  A1 = V, A2 = Mv           bias = b_o       W = W_o  act = tau
//                  base.py:185-188  A1 = H_v, A2 = V_d        bias = b_d       W = W_d  (no act)
//
// The reference materialises torch.cat(...) ([E, d_v+d_e] / [V, d_v+d_h]) and the gathered
// V[src] before every nn.Linear; here concatenation and gather happen in the A-operand loader.
//
// gfx950 mapping.  fp32 has no reduced-precision matrix path on CDNA4 (no xf32): the exact-fp32
// v_mfma_f32_16x16x4_f32 (32 cycles / SIMD, 256 FLOP/clk/CU = the fp32 vector peak, 157 TF chip)
// is the roof.  A 256-thread workgroup (4 waves, one per SIMD) owns a BM x BN output panel,
// BM = 16*RT rows, BN = 64*WN columns; wave w owns the 16*WN-column slice w.  With d_h = 300,
// WN = 5 gives BN = 320 >= N, so a panel holds COMPLETE output rows: the A tile is read once and
// the H0-add / activation epilogue sees whole rows.  K is walked in 32-wide chunks, double
// buffered in LDS; global loads of chunk c+1 are issued before the MFMAs of chunk c and written to
// LDS after them (one barrier per chunk).
//
// k-permutation.  MFMA 16x16x4 takes A[i][k] from lane (i = l&15, k = l>>4).  Since a dot product
// does not care in which order k is visited, lane group g = l>>4 reads 8 CONSECUTIVE k
// (two ds_read_b128) and feeds them to 8 successive MFMAs: MFMA q of a chunk contracts
// k = {8g + q : g = 0..3}.  A and B use the same assignment, so the result is the exact fp32 dot
// product (fmaf chain in a fixed, deterministic k order).
#include "dmpnn_common.hpp"

namespace dmpnn {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;        // k chunk
constexpr int BKP = BK + 4;   // padded LDS row (floats): 144 B rows keep ds_read_b128 16-B aligned
constexpr int kThreads = 256;

struct GemmDev {
    dmpnn_gemm_args a;
    int vecA1, vecA2, vecB;  // 16-byte vector loads legal for the operand
};

// ---- global -> register staging of one k-chunk ------------------------------------------------
// A tile: BM x BK, element (r, k).  Thread t handles float4 slots f = t + 256*j, r = f / 8, kq = f % 8.
template <int RT>
struct AStage {
    static constexpr int BM = 16 * RT;
    static constexpr int SLOTS = (BM * (BK / 4) + kThreads - 1) / kThreads;
    float4 v[SLOTS];
};
template <int WN>
struct BStage {
    static constexpr int BN = 64 * WN;
    static constexpr int SLOTS = BN * (BK / 4) / kThreads;  // 2*WN
    float4 v[SLOTS];
};

__device__ __forceinline__ float a_elem(const dmpnn_gemm_args& a, int64_t arow1, int64_t row, int kk) {
    const int K = (int)(a.K1 + a.K2);
    if (kk < (int)a.K1) return a.A1[arow1 * a.lda1 + kk];
    if (kk < K) return a.A2[row * a.lda2 + (kk - (int)a.K1)];
    return 0.f;
}

template <int RT>
__device__ __forceinline__ void load_a(const GemmDev& g, AStage<RT>& st, int64_t row0, int k0, int tid) {
    const dmpnn_gemm_args& a = g.a;
    constexpr int BM = 16 * RT;
    const int K1 = (int)a.K1, K = (int)(a.K1 + a.K2);
#pragma unroll
    for (int j = 0; j < AStage<RT>::SLOTS; ++j) {
        const int f = tid + kThreads * j;
        const int r = f >> 3, kq = f & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < BM) {
            const int64_t row = row0 + r;
            if (row < a.M) {
                const int kk = k0 + kq * 4;
                const int64_t arow1 = a.gather1 ? (int64_t)a.gather1[row] : row;
                if (kk + 3 < K1 && g.vecA1) {
                    v = *reinterpret_cast<const float4*>(a.A1 + arow1 * a.lda1 + kk);
                } else if (kk >= K1 && kk + 3 < K && g.vecA2) {
                    v = *reinterpret_cast<const float4*>(a.A2 + row * a.lda2 + (kk - K1));
                } else if (kk < K) {
                    v.x = a_elem(a, arow1, row, kk);
                    v.y = a_elem(a, arow1, row, kk + 1);
                    v.z = a_elem(a, arow1, row, kk + 2);
                    v.w = a_elem(a, arow1, row, kk + 3);
                }
            }
        }
        st.v[j] = v;
    }
}

template <int WN>
__device__ __forceinline__ void load_b(const GemmDev& g, BStage<WN>& st, int64_t col0, int k0, int tid) {
    const dmpnn_gemm_args& a = g.a;
    const int K = (int)(a.K1 + a.K2);
#pragma unroll
    for (int j = 0; j < BStage<WN>::SLOTS; ++j) {
        const int f = tid + kThreads * j;
        const int n = f >> 3, kq = f & 7;
        const int64_t col = col0 + n;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < a.N) {
            const int kk = k0 + kq * 4;
            const float* w = a.W + col * a.ldw + kk;
            if (kk + 3 < K && g.vecB) {
                v = *reinterpret_cast<const float4*>(w);
            } else {
                if (kk < K) v.x = w[0];
                if (kk + 1 < K) v.y = w[1];
                if (kk + 2 < K) v.z = w[2];
                if (kk + 3 < K) v.w = w[3];
            }
        }
        st.v[j] = v;
    }
}

template <int RT>
__device__ __forceinline__ void store_a(const AStage<RT>& st, float* As, int tid) {
    constexpr int BM = 16 * RT;
#pragma unroll
    for (int j = 0; j < AStage<RT>::SLOTS; ++j) {
        const int f = tid + kThreads * j;
        const int r = f >> 3, kq = f & 7;
        if (r < BM) *reinterpret_cast<float4*>(As + r * BKP + kq * 4) = st.v[j];
    }
}
template <int WN>
__device__ __forceinline__ void store_b(const BStage<WN>& st, float* Bs, int tid) {
#pragma unroll
    for (int j = 0; j < BStage<WN>::SLOTS; ++j) {
        const int f = tid + kThreads * j;
        const int n = f >> 3, kq = f & 7;
        *reinterpret_cast<float4*>(Bs + n * BKP + kq * 4) = st.v[j];
    }
}

template <int RT, int WN>
__global__ __launch_bounds__(kThreads) void k_linear(GemmDev g) {
    constexpr int BM = 16 * RT, BN = 64 * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][BM][BKP]
    float* Bs = smem + 2 * BM * BKP;       // [2][BN][BKP]
    const dmpnn_gemm_args& a = g.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int64_t col0 = (int64_t)blockIdx.y * BN;
    const int K = (int)(a.K1 + a.K2);
    const int n_chunks = (K + BK - 1) / BK;

    f32x4 acc[RT][WN];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    AStage<RT> sa;
    BStage<WN> sb;
    load_a<RT>(g, sa, row0, 0, tid);
    load_b<WN>(g, sb, col0, 0, tid);
    store_a<RT>(sa, As, tid);
    store_b<WN>(sb, Bs, tid);
    __syncthreads();

    for (int c = 0; c < n_chunks; ++c) {
        const int cur = c & 1;
        if (c + 1 < n_chunks) {
            load_a<RT>(g, sa, row0, (c + 1) * BK, tid);
            load_b<WN>(g, sb, col0, (c + 1) * BK, tid);
        }
        const float* Ac = As + cur * BM * BKP;
        const float* Bc = Bs + cur * BN * BKP + wave * (16 * WN) * BKP;
        // fragments: lane (li, lg) holds k = 8*lg .. 8*lg+7 of row li of every 16-row tile
        float af[RT][8], bf[WN][8];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float* p = Ac + (rt * 16 + li) * BKP + lg * 8;
            const float4 t0 = *reinterpret_cast<const float4*>(p);
            const float4 t1 = *reinterpret_cast<const float4*>(p + 4);
            af[rt][0] = t0.x; af[rt][1] = t0.y; af[rt][2] = t0.z; af[rt][3] = t0.w;
            af[rt][4] = t1.x; af[rt][5] = t1.y; af[rt][6] = t1.z; af[rt][7] = t1.w;
        }
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) {
            const float* p = Bc + (ct * 16 + li) * BKP + lg * 8;
            const float4 t0 = *reinterpret_cast<const float4*>(p);
            const float4 t1 = *reinterpret_cast<const float4*>(p + 4);
            bf[ct][0] = t0.x; bf[ct][1] = t0.y; bf[ct][2] = t0.z; bf[ct][3] = t0.w;
            bf[ct][4] = t1.x; bf[ct][5] = t1.y; bf[ct][6] = t1.z; bf[ct][7] = t1.w;
        }
        // q outermost: RT*WN independent accumulators between two MFMAs on the same one
        // (dependent-accumulator latency of 16x16x4 f32 is 40 cycles vs 32-cycle issue).
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[rt][q], bf[ct][q], acc[rt][ct], 0, 0, 0);
        if (c + 1 < n_chunks) {
            store_a<RT>(sa, As + (cur ^ 1) * BM * BKP, tid);
            store_b<WN>(sb, Bs + (cur ^ 1) * BN * BKP, tid);
        }
        __syncthreads();
    }

    // ---- epilogue: bias + residual + activation.  C/D layout of 16x16x4: col = l&15,
    // row = (l>>4)*4 + reg.
    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int64_t col = col0 + wave * (16 * WN) + ct * 16 + li;
        if (col >= a.N) continue;
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + rt * 16 + lg * 4 + r;
                if (row >= a.M) continue;
                float z = acc[rt][ct][r] + bv;
                if (a.Cadd) z = a.Cadd[row * a.ldcadd + col] + z;
                if (a.Zpre) a.Zpre[row * a.ldz + col] = z;
                a.C[row * a.ldc + col] = apply_act(z, a.act, slope);
            }
        }
    }
}

// Plain one-thread-per-output kernel.  NOT a product path: selected only by the environment
// variable DMPNN_DEBUG_VALU_GEMM=1 to triage an MFMA-layout failure on real hardware.
__global__ void k_linear_valu(GemmDev g) {
    const dmpnn_gemm_args& a = g.a;
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= a.M * a.N) return;
    const int64_t row = idx / a.N, col = idx % a.N;
    const int K = (int)(a.K1 + a.K2);
    const int64_t arow1 = a.gather1 ? (int64_t)a.gather1[row] : row;
    float z = 0.f;
    for (int k = 0; k < K; ++k) z = fmaf(a_elem(a, arow1, row, k), a.W[col * a.ldw + k], z);
    if (a.bias) z += a.bias[col];
    if (a.Cadd) z = a.Cadd[row * a.ldcadd + col] + z;
    if (a.Zpre) a.Zpre[row * a.ldz + col] = z;
    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
    a.C[row * a.ldc + col] = apply_act(z, a.act, slope);
}

template <int RT, int WN>
int launch_tile(const GemmDev& g, hipStream_t s) {
    constexpr int BM = 16 * RT, BN = 64 * WN;
    const size_t lds = (size_t)2 * (BM + BN) * BKP * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear<RT, WN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(k_linear<%d,%d>, %zu B LDS): %s", RT, WN, lds, hipGetErrorString(e));
            return DMPNN_EHIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)((g.a.M + BM - 1) / BM), (unsigned)((g.a.N + BN - 1) / BN));
    hipLaunchKernelGGL((k_linear<RT, WN>), grid, dim3(kThreads), lds, s, g);
    DMPNN_CHECK_LAUNCH("k_linear");
    return DMPNN_OK;
}

// Pick the row-tile height: minimise (#rounds over 256 CUs) x (rows per tile + fixed cost of
// streaming W through the CU once per tile).
int pick_rt(int64_t M, int64_t n_col_blocks) {
    static const int cand[] = {1, 2, 3, 4, 6, 8};
    int best = 1;
    double best_cost = 1e300;
    for (int rt : cand) {
        const int64_t tiles = ((M + 16 * rt - 1) / (16 * rt)) * n_col_blocks;
        const int64_t rounds = (tiles + 255) / 256;
        const double cost = (double)rounds * (rt + 0.6);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = rt;
        }
    }
    return best;
}

}  // namespace

int launch_linear(const dmpnn_gemm_args& a, hipStream_t s) {
    DMPNN_CHECK_ARG(a.M >= 0 && a.N >= 0 && a.K1 >= 0 && a.K2 >= 0, "linear: negative size");
    if (a.M == 0 || a.N == 0) return DMPNN_OK;
    DMPNN_CHECK_ARG(a.W && a.C, "linear: null W or C");
    DMPNN_CHECK_ARG(a.K1 == 0 || a.A1, "linear: null A1 with K1 > 0");
    DMPNN_CHECK_ARG(a.K2 == 0 || a.A2, "linear: null A2 with K2 > 0");
    GemmDev g;
    g.a = a;
    g.vecA1 = a.A1 && aligned16(a.A1) && (a.lda1 % 4 == 0);
    g.vecA2 = a.A2 && aligned16(a.A2) && (a.lda2 % 4 == 0) && (a.K1 % 4 == 0);
    g.vecB = aligned16(a.W) && (a.ldw % 4 == 0);

    static const bool debug_valu = [] {
        const char* e = getenv("DMPNN_DEBUG_VALU_GEMM");
        return e && e[0] == '1';
    }();
    if (debug_valu) {
        const int64_t n = a.M * a.N;
        hipLaunchKernelGGL(k_linear_valu, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g);
        DMPNN_CHECK_LAUNCH("k_linear_valu");
        return DMPNN_OK;
    }

    const int wn = (a.N <= 128) ? 2 : 5;
    const int64_t ncb = (a.N + 64 * wn - 1) / (64 * wn);
    const int rt = pick_rt(a.M, ncb);
#define DMPNN_TILE(R, W_) \
    if (rt == R && wn == W_) return launch_tile<R, W_>(g, s);
    DMPNN_TILE(1, 2) DMPNN_TILE(2, 2) DMPNN_TILE(3, 2) DMPNN_TILE(4, 2) DMPNN_TILE(6, 2) DMPNN_TILE(8, 2)
    DMPNN_TILE(1, 5) DMPNN_TILE(2, 5) DMPNN_TILE(3, 5) DMPNN_TILE(4, 5) DMPNN_TILE(6, 5) DMPNN_TILE(8, 5)
#undef DMPNN_TILE
    set_error("linear: no tile for rt=%d wn=%d", rt, wn);
    return DMPNN_EINVAL;
}

}  // namespace dmpnn
