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
// the residual / activation epilogue sees whole rows.  K is walked in 32-wide chunks through a
// 2-slot LDS ring plus one chunk in staging registers: global loads run two chunks ahead, the LDS
// writes of chunk c+1 are issued right before the MFMAs of chunk c (one barrier per chunk).
//
// k-permutation.  MFMA 16x16x4 takes A[i][k] from lane (i = l&15, k = l>>4).  Since a dot product
// does not care in which order k is visited, lane group g = l>>4 reads 8 CONSECUTIVE k
// (two ds_read_b128) and feeds them to 8 successive MFMAs: MFMA q of a chunk contracts
// k = {8g + q : g = 0..3}.  A and B use the same assignment, so the result is the exact fp32 dot
// product (an fmaf chain in a fixed, deterministic k order), started from residual + bias.
//
// Load discipline (cdna_hip_programming.md §5 trap (c)): every global load is UNCONDITIONAL — an
// out-of-range slot reads a clamped, valid address and is zeroed by a select when it is written to
// LDS, after the MFMA block.  A load under a runtime branch makes hipcc wait vmcnt(0) per load.
#include <stdlib.h>

#include "dmpnn_common.hpp"

namespace dmpnn {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;        // k chunk
constexpr int BKP = BK + 4;   // padded LDS row (floats): 144 B rows keep ds_read_b128 16-B aligned
constexpr int kThreads = 256;

struct GemmDev {
    dmpnn_gemm_args a;
};

// ---- global -> register staging of one k-chunk ------------------------------------------------
// Tile element (r, k) lives in float4 slot f = r*8 + k/4; thread t owns slots f = t + 256*j.
template <int RT>
struct AStage {
    static constexpr int BM = 16 * RT;
    static constexpr int SLOTS = (BM * (BK / 4) + kThreads - 1) / kThreads;
    float4 v[SLOTS];
    unsigned ok[SLOTS];      // 4 validity bits per slot (bit t: element k+t is inside the matrix)
    int64_t off1[SLOTS];     // A1 row offset (elements) of the slot's (gathered) row; chunk-invariant
    int64_t off2[SLOTS];     // A2 row offset
    unsigned rowok;          // bit j: the slot's row is inside the matrix
};
template <int WN>
struct BStage {
    static constexpr int BN = 64 * WN;
    static constexpr int SLOTS = BN * (BK / 4) / kThreads;  // 2*WN
    float4 v[SLOTS];
    unsigned ok[SLOTS];
    int64_t off[SLOTS];      // W row offset of the slot's output column
    unsigned colok;
};

template <int RT>
__device__ __forceinline__ void init_a(const dmpnn_gemm_args& a, AStage<RT>& st, int64_t row0, int tid) {
    st.rowok = 0;
#pragma unroll
    for (int j = 0; j < AStage<RT>::SLOTS; ++j) {
        const int r = (tid + kThreads * j) >> 3;
        const int64_t row = row0 + r;
        const bool ok = (r < 16 * RT) && (row < a.M);
        const int64_t rc = ok ? row : 0;
        const int64_t g = a.gather1 ? (int64_t)a.gather1[rc] : rc;
        st.off1[j] = g * a.lda1;
        st.off2[j] = rc * a.lda2;
        st.rowok |= (ok ? 1u : 0u) << j;
    }
}
template <int WN>
__device__ __forceinline__ void init_b(const dmpnn_gemm_args& a, BStage<WN>& st, int64_t col0, int tid) {
    st.colok = 0;
#pragma unroll
    for (int j = 0; j < BStage<WN>::SLOTS; ++j) {
        const int64_t col = col0 + ((tid + kThreads * j) >> 3);
        const bool ok = col < a.N;
        st.off[j] = (ok ? col : 0) * a.ldw;
        st.colok |= (ok ? 1u : 0u) << j;
    }
}

template <int RT, bool VEC>
__device__ __forceinline__ void load_a(const dmpnn_gemm_args& a, AStage<RT>& st, int k0, int tid) {
    const int K1 = (int)a.K1, K = (int)(a.K1 + a.K2);
#pragma unroll
    for (int j = 0; j < AStage<RT>::SLOTS; ++j) {
        const int kk = k0 + ((tid + kThreads * j) & 7) * 4;
        const bool rok = (st.rowok >> j) & 1u;
        if (VEC) {  // K1, K2 multiples of 4: a quad is entirely in A1, entirely in A2, or past K
            const bool ok = rok && kk < K;
            const float* p = (kk < K1) ? a.A1 + st.off1[j] + kk : a.A2 + st.off2[j] + (kk - K1);
            st.v[j] = *reinterpret_cast<const float4*>(ok ? p : a.A1);
            st.ok[j] = ok ? 0xFu : 0u;
        } else {
            float x[4];
            unsigned m = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kt = kk + t;
                const bool ok = rok && kt < K;
                const float* p = (kt < K1) ? a.A1 + st.off1[j] + kt : a.A2 + st.off2[j] + (kt - K1);
                x[t] = *(ok ? p : a.W);
                m |= (ok ? 1u : 0u) << t;
            }
            st.v[j] = make_float4(x[0], x[1], x[2], x[3]);
            st.ok[j] = m;
        }
    }
}

template <int WN, bool VEC>
__device__ __forceinline__ void load_b(const dmpnn_gemm_args& a, BStage<WN>& st, int k0, int tid) {
    const int K = (int)(a.K1 + a.K2);
#pragma unroll
    for (int j = 0; j < BStage<WN>::SLOTS; ++j) {
        const int kk = k0 + ((tid + kThreads * j) & 7) * 4;
        const bool cok = (st.colok >> j) & 1u;
        if (VEC) {
            const bool ok = cok && kk < K;
            st.v[j] = *reinterpret_cast<const float4*>(a.W + (ok ? st.off[j] + kk : 0));
            st.ok[j] = ok ? 0xFu : 0u;
        } else {
            float x[4];
            unsigned m = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool ok = cok && (kk + t) < K;
                x[t] = a.W[ok ? st.off[j] + kk + t : 0];
                m |= (ok ? 1u : 0u) << t;
            }
            st.v[j] = make_float4(x[0], x[1], x[2], x[3]);
            st.ok[j] = m;
        }
    }
}

__device__ __forceinline__ float4 masked(float4 v, unsigned m) {
    return make_float4((m & 1u) ? v.x : 0.f, (m & 2u) ? v.y : 0.f, (m & 4u) ? v.z : 0.f, (m & 8u) ? v.w : 0.f);
}

template <int RT>
__device__ __forceinline__ void store_a(const AStage<RT>& st, float* As, int tid) {
    constexpr int BM = 16 * RT;
#pragma unroll
    for (int j = 0; j < AStage<RT>::SLOTS; ++j) {
        const int f = tid + kThreads * j;
        const int r = f >> 3, kq = f & 7;
        if (r < BM) *reinterpret_cast<float4*>(As + r * BKP + kq * 4) = masked(st.v[j], st.ok[j]);
    }
}
template <int WN>
__device__ __forceinline__ void store_b(const BStage<WN>& st, float* Bs, int tid) {
#pragma unroll
    for (int j = 0; j < BStage<WN>::SLOTS; ++j) {
        const int f = tid + kThreads * j;
        const int n = f >> 3, kq = f & 7;
        *reinterpret_cast<float4*>(Bs + n * BKP + kq * 4) = masked(st.v[j], st.ok[j]);
    }
}

// activation with a compile-time code: the epilogue is straight-line code per activation
template <int ACT>
__device__ __forceinline__ float act_ct(float z, float slope) {
    if (ACT == DMPNN_ACT_RELU) return z < 0.f ? 0.f : z;
    if (ACT == DMPNN_ACT_LEAKYRELU || ACT == DMPNN_ACT_PRELU) return z > 0.f ? z : slope * z;
    if (ACT == DMPNN_ACT_TANH) return tanhf(z);
    if (ACT == DMPNN_ACT_ELU) return z > 0.f ? z : expm1f(z);
    return z;
}

template <int RT, int WN, int ACT>
__device__ __forceinline__ void epilogue(const dmpnn_gemm_args& a, const f32x4 (&acc)[RT][WN], int64_t row0,
                                         int64_t colw, int li, int lg, float slope) {
    // C/D layout of 16x16x4: col = l&15, row = (l>>4)*4 + reg
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int64_t col = colw + ct * 16 + li;
        if (col >= a.N) continue;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + rt * 16 + lg * 4 + r;
                if (row >= a.M) continue;
                const float z = acc[rt][ct][r];
                if (a.Zpre) a.Zpre[row * a.ldz + col] = z;
                a.C[row * a.ldc + col] = act_ct<ACT>(z, slope);
            }
        }
    }
}

template <int RT, int WN, bool VEC>
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
    const int64_t colw = col0 + wave * (16 * WN);
    const int K = (int)(a.K1 + a.K2);
    const int n_chunks = (K + BK - 1) / BK;

    AStage<RT> sa;
    BStage<WN> sb;
    init_a<RT>(a, sa, row0, tid);
    init_b<WN>(a, sb, col0, tid);
    load_a<RT, VEC>(a, sa, 0, tid);
    load_b<WN, VEC>(a, sb, 0, tid);

    // Accumulators start from residual + bias (the C-in of the first MFMA): the H0 tile is fetched
    // here, under the first chunk's staging loads, instead of in a load-bound epilogue.
    f32x4 acc[RT][WN];
    {
        const float* bias_p = a.bias ? a.bias : a.W;  // dummy bases keep the loads unconditional
        const float* cadd_p = a.Cadd ? a.Cadd : a.W;
        const int64_t ldcadd = a.Cadd ? a.ldcadd : 0;
        const int64_t cmask = a.Cadd ? ~int64_t(0) : 0;
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) {
            const int64_t col = colw + ct * 16 + li;
            const bool okc = col < a.N;
            const float braw = bias_p[okc ? col : 0];
            const float bv = a.bias ? braw : 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + rt * 16 + lg * 4 + r;
                    const bool ok = okc && row < a.M;
                    const float craw = cadd_p[ok ? (row * ldcadd + (col & cmask)) : 0];
                    acc[rt][ct][r] = a.Cadd ? craw + bv : bv;
                }
            }
        }
    }
    store_a<RT>(sa, As, tid);
    store_b<WN>(sb, Bs, tid);
    if (n_chunks > 1) {  // chunk 1 waits in registers while chunk 0 is contracted
        load_a<RT, VEC>(a, sa, BK, tid);
        load_b<WN, VEC>(a, sb, BK, tid);
    }
    __syncthreads();

    // Software pipeline, one barrier per chunk.  At the top of iteration c: LDS[cur] holds chunk c,
    // the staging registers hold chunk c+1 (loaded a whole iteration ago), LDS[cur^1] is free.
    //   ds_read fragments(c)  ->  ds_write regs(c+1) -> LDS[cur^1]  ->  global loads (c+2) -> regs
    //   ->  120 MFMAs (the LDS writes and the global loads complete underneath them)  ->  barrier
    for (int c = 0; c < n_chunks; ++c) {
        const int cur = c & 1;
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
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < n_chunks) {  // uniform branches around WHOLE blocks only
            store_a<RT>(sa, As + (cur ^ 1) * BM * BKP, tid);
            store_b<WN>(sb, Bs + (cur ^ 1) * BN * BKP, tid);
        }
        if (c + 2 < n_chunks) {
            load_a<RT, VEC>(a, sa, (c + 2) * BK, tid);
            load_b<WN, VEC>(a, sb, (c + 2) * BK, tid);
        }
        __builtin_amdgcn_sched_barrier(0);
        // q outermost: RT*WN independent accumulators between two MFMAs on the same one
        // (dependent-accumulator latency of 16x16x4 f32 is 40 cycles vs 32-cycle issue).
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int ct = 0; ct < WN; ++ct)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[rt][q], bf[ct][q], acc[rt][ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
    switch (a.act) {
        case DMPNN_ACT_RELU: epilogue<RT, WN, DMPNN_ACT_RELU>(a, acc, row0, colw, li, lg, slope); break;
        case DMPNN_ACT_LEAKYRELU:
        case DMPNN_ACT_PRELU: epilogue<RT, WN, DMPNN_ACT_LEAKYRELU>(a, acc, row0, colw, li, lg, slope); break;
        case DMPNN_ACT_TANH: epilogue<RT, WN, DMPNN_ACT_TANH>(a, acc, row0, colw, li, lg, slope); break;
        case DMPNN_ACT_ELU: epilogue<RT, WN, DMPNN_ACT_ELU>(a, acc, row0, colw, li, lg, slope); break;
        default: epilogue<RT, WN, DMPNN_ACT_NONE>(a, acc, row0, colw, li, lg, slope); break;
    }
}

// Plain one-thread-per-output kernel.  NOT a product path: selected only by the environment
// variable DMPNN_DEBUG_VALU_GEMM=1 to triage an MFMA-layout failure on real hardware.
__global__ void k_linear_valu(GemmDev g) {
    const dmpnn_gemm_args& a = g.a;
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= a.M * a.N) return;
    const int64_t row = idx / a.N, col = idx % a.N;
    const int K1 = (int)a.K1, K = (int)(a.K1 + a.K2);
    const int64_t arow1 = a.gather1 ? (int64_t)a.gather1[row] : row;
    float z = 0.f;
    for (int k = 0; k < K; ++k) {
        const float x = k < K1 ? a.A1[arow1 * a.lda1 + k] : a.A2[row * a.lda2 + (k - K1)];
        z = fmaf(x, a.W[col * a.ldw + k], z);
    }
    if (a.bias) z += a.bias[col];
    if (a.Cadd) z = a.Cadd[row * a.ldcadd + col] + z;
    if (a.Zpre) a.Zpre[row * a.ldz + col] = z;
    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
    a.C[row * a.ldc + col] = apply_act(z, a.act, slope);
}

template <int RT, int WN, bool VEC>
int launch_tile(const GemmDev& g, hipStream_t s) {
    constexpr int BM = 16 * RT, BN = 64 * WN;
    const size_t lds = (size_t)2 * (BM + BN) * BKP * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear<RT, WN, VEC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(k_linear<%d,%d>, %zu B LDS): %s", RT, WN, lds, hipGetErrorString(e));
            return DMPNN_EHIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)((g.a.M + BM - 1) / BM), (unsigned)((g.a.N + BN - 1) / BN));
    hipLaunchKernelGGL((k_linear<RT, WN, VEC>), grid, dim3(kThreads), lds, s, g);
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
    DMPNN_CHECK_ARG(a.K1 + a.K2 > 0, "linear: empty contraction");
    DMPNN_CHECK_ARG(a.K1 == 0 || a.A1, "linear: null A1 with K1 > 0");
    DMPNN_CHECK_ARG(a.K2 == 0 || a.A2, "linear: null A2 with K2 > 0");
    GemmDev g;
    g.a = a;
    // keep the never-dereferenced side of the A1/A2 pointer select on a valid base
    if (a.K1 == 0) { g.a.A1 = a.A2; g.a.lda1 = a.lda2; g.a.gather1 = nullptr; }
    if (a.K2 == 0) { g.a.A2 = g.a.A1; g.a.lda2 = g.a.lda1; }
    const bool vecA = (a.K1 == 0 || (aligned16(a.A1) && a.lda1 % 4 == 0)) && (a.K1 % 4 == 0) &&
                      (a.K2 == 0 || (aligned16(a.A2) && a.lda2 % 4 == 0 && a.K2 % 4 == 0));
    const bool vecB = aligned16(a.W) && (a.ldw % 4 == 0) && ((a.K1 + a.K2) % 4 == 0);
    const bool vec = vecA && vecB;

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
#define DMPNN_TILE(R, W_)                                                   \
    if (rt == R && wn == W_)                                                \
        return vec ? launch_tile<R, W_, true>(g, s) : launch_tile<R, W_, false>(g, s);
    DMPNN_TILE(1, 2) DMPNN_TILE(2, 2) DMPNN_TILE(3, 2) DMPNN_TILE(4, 2) DMPNN_TILE(6, 2) DMPNN_TILE(8, 2)
    DMPNN_TILE(1, 5) DMPNN_TILE(2, 5) DMPNN_TILE(3, 5) DMPNN_TILE(4, 5) DMPNN_TILE(6, 5) DMPNN_TILE(8, 5)
#undef DMPNN_TILE
    set_error("linear: no tile for rt=%d wn=%d", rt, wn);
    return DMPNN_EINVAL;
}

}  // namespace dmpnn
