// f4: what chemprop.models.MPNN does AFTER the message-passing block in a training step, as kernels chained
// by ONE C call — and the whole step (K0 + forward + this + backward + optimizer) as one more (dmpnn_train_step):
//
//     H   = agg(H_v, batch)                     models/model.py:131      nn/agg.py:66-113     (dmpnn_molagg_*)
//     Z   = bn(H)                               models/model.py:132      nn.BatchNorm1d, batch statistics in training
//     P   = ffn(Z)                              models/model.py:146,155  nn/ffn.py:24-68, nn/predictors.py:161-169
//     l   = sum(L w_i t_j mask) / sum(mask)     models/model.py:156      nn/metrics.py:78-127 (MSE :137-141, MAE :146-148,
//                                                                         bounded variants :157-163)
// and the gradients of l with respect to every parameter above and to H_v (the `gout` of dmpnn_backward).
//
// The reference runs this as ~60 ATen launches from Python (forward + autograd); at 512 molecules every one of them is
// launch latency.  Here: segment reduction, one batch-norm kernel, one fp32-MFMA contraction per layer (activation fused),
// one loss kernel that also emits dl/dP; backward: per layer one weight-gradient product (+ reduce), one transposed
// contraction with the activation derivative fused into a small elementwise pass, one batch-norm kernel, one gather.
#include "dmpnn_common.hpp"
#include "dmpnn_mega16_impl.hpp"   // (mega16::SplitArgs / split_weights_wave / SplitW / scale_for / split4: the predictor's first layer on the f16 pipe)

namespace dmpnn {
extern thread_local long long* g_debug_stamps;   // (dmpnn_debug_timestamps: cycle stamps of one workgroup, scripts/probe_head_rows.py)
namespace {

inline size_t al256(size_t x) { return (x + 255) & ~size_t(255); }

// ---- BatchNorm1d over the rows of X [B, d] -------------------------------------------------------------------------
// One workgroup of 1024 threads per 16 columns: 64 row lanes per column, so a thread walks B / 64 rows (8 at 512 molecules;
// the first version — 4 row lanes, 128 dependent iterations per pass — took 65 us for 0.6 MB).  Two passes over the rows for
// the statistics (mean, then the mean squared deviation: the arithmetic of torch's batch_norm on a [B, d] input to fp32
// rounding), a third for y.  B x d is ~0.6 MB: it lives in L2.
constexpr int kBnCols = 16, kBnLanes = 64;
struct BnArgs {
    const float* X; int64_t ldx; float* Y; int64_t ldy;
    __device__ float* X_out() const { return const_cast<float*>(X); }   // (k_agg_bn_fwd writes the aggregate it then normalises)
    const float* gamma; const float* beta; float* run_mean; float* run_var;
    float* save_mean; float* save_invstd;      // [d] each (training: for the backward pass)
    int64_t B; int d; float eps, momentum; int training;
    int64_t* n_tracked;                        // nn.BatchNorm1d.num_batches_tracked (training: += 1) or NULL
};
// column sums over the 64 row lanes: the four row lanes of a wave by lane shuffles, the sixteen waves through LDS
__device__ __forceinline__ float bn_col_sum(float (*red)[kBnCols], int tx, int ty, float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    __syncthreads();            // (the previous use of `red` is over)
    if ((ty & 3) == 0) red[ty >> 2][tx] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kBnLanes / 4; ++i) s += red[i][tx];
    return s;
}
// REG: B <= 8 x 64 rows — a thread's rows stay in registers (X is read once instead of three times)
constexpr int kBnRegRows = 8;
template <bool REG>
__global__ __launch_bounds__(1024) void k_bn_fwd(BnArgs a) {
    __shared__ float red[kBnLanes / 4][kBnCols];
    const int tx = threadIdx.x & (kBnCols - 1), ty = threadIdx.x / kBnCols;
    const int c = blockIdx.x * kBnCols + tx;
    const bool ok = c < a.d;
    float mean, invstd;
    float xs[kBnRegRows];
    if constexpr (REG) {
#pragma unroll
        for (int i = 0; i < kBnRegRows; ++i) {
            const int64_t r = ty + (int64_t)kBnLanes * i;
            xs[i] = (ok && r < a.B) ? a.X[r * a.ldx + c] : 0.f;
        }
    }
    if (a.training) {
        if (a.n_tracked && blockIdx.x == 0 && threadIdx.x == 0) *a.n_tracked += 1;
        float s = 0.f;
        if constexpr (REG) {
#pragma unroll
            for (int i = 0; i < kBnRegRows; ++i) s += xs[i];
        } else if (ok) {
            for (int64_t r = ty; r < a.B; r += kBnLanes) s += a.X[r * a.ldx + c];
        }
        mean = bn_col_sum(red, tx, ty, s) / (float)a.B;
        float q = 0.f;
        if constexpr (REG) {
#pragma unroll
            for (int i = 0; i < kBnRegRows; ++i) { const float dlt = xs[i] - mean; q += (ty + (int64_t)kBnLanes * i < a.B) ? dlt * dlt : 0.f; }
        } else if (ok) {
            for (int64_t r = ty; r < a.B; r += kBnLanes) { const float dlt = a.X[r * a.ldx + c] - mean; q += dlt * dlt; }
        }
        const float ss = bn_col_sum(red, tx, ty, q);
        const float var = ss / (float)a.B;                       // biased: what normalises (nn.BatchNorm1d)
        invstd = 1.f / sqrtf(var + a.eps);
        if (ok && ty == 0) {
            a.save_mean[c] = mean; a.save_invstd[c] = invstd;
            if (a.run_mean) a.run_mean[c] = (1.f - a.momentum) * a.run_mean[c] + a.momentum * mean;
            // running_var takes the UNBIASED estimate (torch: var * B / (B - 1))
            if (a.run_var) a.run_var[c] = (1.f - a.momentum) * a.run_var[c] + a.momentum * (a.B > 1 ? ss / (float)(a.B - 1) : var);
        }
    } else {
        mean = ok ? a.run_mean[c] : 0.f;
        invstd = ok ? 1.f / sqrtf(a.run_var[c] + a.eps) : 0.f;
    }
    if (ok) {
        const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
        if constexpr (REG) {
#pragma unroll
            for (int i = 0; i < kBnRegRows; ++i) {
                const int64_t r = ty + (int64_t)kBnLanes * i;
                if (r < a.B) a.Y[r * a.ldy + c] = (xs[i] - mean) * invstd * g + b;
            }
        } else {
            for (int64_t r = ty; r < a.B; r += kBnLanes) a.Y[r * a.ldy + c] = (a.X[r * a.ldx + c] - mean) * invstd * g + b;
        }
    }
}

// gX = gamma invstd / B (B gY - sum gY - xhat sum(gY xhat));  g_gamma = sum gY xhat;  g_beta = sum gY      (training)
// gX = gY gamma invstd                                                                                     (eval statistics)
struct BnBwdArgs {
    const float* gY; int64_t ldgy; const float* X; int64_t ldx; float* gX; int64_t ldgx;
    const float* gamma; const float* save_mean; const float* save_invstd; const float* run_mean; const float* run_var;
    float* g_gamma; float* g_beta;
    int64_t B; int d; float eps; int training;
};
template <bool REG>
__global__ __launch_bounds__(1024) void k_bn_bwd(BnBwdArgs a) {
    __shared__ float red[kBnLanes / 4][kBnCols];
    const int tx = threadIdx.x & (kBnCols - 1), ty = threadIdx.x / kBnCols;
    const int c = blockIdx.x * kBnCols + tx;
    const bool ok = c < a.d;
    const float mean = ok ? (a.training ? a.save_mean[c] : a.run_mean[c]) : 0.f;
    const float invstd = ok ? (a.training ? a.save_invstd[c] : 1.f / sqrtf(a.run_var[c] + a.eps)) : 0.f;
    float s1 = 0.f, s2 = 0.f;
    float gs[kBnRegRows], xh[kBnRegRows];
    if constexpr (REG) {
#pragma unroll
        for (int i = 0; i < kBnRegRows; ++i) {
            const int64_t r = ty + (int64_t)kBnLanes * i;
            const bool in = ok && r < a.B;
            gs[i] = in ? a.gY[r * a.ldgy + c] : 0.f;
            xh[i] = in ? (a.X[r * a.ldx + c] - mean) * invstd : 0.f;
        }
#pragma unroll
        for (int i = 0; i < kBnRegRows; ++i) { s1 += gs[i]; s2 += gs[i] * xh[i]; }
    } else if (ok) {
        for (int64_t r = ty; r < a.B; r += kBnLanes) {
            const float g = a.gY[r * a.ldgy + c];
            s1 += g;
            s2 += g * ((a.X[r * a.ldx + c] - mean) * invstd);
        }
    }
    s1 = bn_col_sum(red, tx, ty, s1);
    s2 = bn_col_sum(red, tx, ty, s2);
    if (!ok) return;
    if (ty == 0) {
        if (a.g_gamma) a.g_gamma[c] = s2;
        if (a.g_beta) a.g_beta[c] = s1;
    }
    const float gam = a.gamma ? a.gamma[c] : 1.f;
    const float k = gam * invstd, invB = 1.f / (float)a.B;
    if constexpr (REG) {
#pragma unroll
        for (int i = 0; i < kBnRegRows; ++i) {
            const int64_t r = ty + (int64_t)kBnLanes * i;
            if (r < a.B) a.gX[r * a.ldgx + c] = a.training ? k * (gs[i] - invB * s1 - xh[i] * invB * s2) : k * gs[i];
        }
    } else {
        for (int64_t r = ty; r < a.B; r += kBnLanes) {
            const float g = a.gY[r * a.ldgy + c];
            if (a.training) {
                const float xhat = (a.X[r * a.ldx + c] - mean) * invstd;
                a.gX[r * a.ldgx + c] = k * (g - invB * s1 - xhat * invB * s2);
            } else {
                a.gX[r * a.ldgx + c] = k * g;
            }
        }
    }
}

// ---- criterion (nn/metrics.py:78-127): ONE workgroup ----------------------------------------------------------------
//   mask = isfinite(target) (models/model.py:152-153), target = nan_to_num(target)
//   bounded: P' = T where (P < T and lt) or (P > T and gt)   (metrics.py:157-161)
//   L = (P' - T)^2 | |P' - T|;   loss = sum(L w_i t_j mask) / sum(mask);   gP = dL/dP w_i t_j mask / sum(mask)
// the unreduced loss of one (prediction, target) pair and its derivative in the prediction:
//   MSE (metrics.py:137-141), MAE (:146-148), BCE with logits (:292-295: F.binary_cross_entropy_with_logits — the classification
//   predictor's train_step hands over raw logits, predictors.py:246-247):  L = (1 - y) x - log_sigmoid(x),  dL/dx = sigmoid(x) - y
__device__ __forceinline__ float loss_value(int kind, float p, float y) {
    if (kind == DMPNN_LOSS_BCE) return (1.f - y) * p - (fminf(p, 0.f) - log1pf(expf(-fabsf(p))));
    const float d = p - y;
    return kind == DMPNN_LOSS_MAE ? fabsf(d) : d * d;
}
__device__ __forceinline__ float loss_deriv(int kind, float p, float y) {
    if (kind == DMPNN_LOSS_BCE) return 1.f / (1.f + expf(-p)) - y;
    const float d = p - y;
    return kind == DMPNN_LOSS_MAE ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
}

struct LossArgs {
    const float* P; int64_t ldp; const float* T; int64_t ldt; const float* w; const float* tw;
    const unsigned char* lt; const unsigned char* gt;
    float* gP; int64_t ldg; float* out;   // out[0] = loss, out[1] = number of finite targets
    int64_t B; int t; int kind;
    int nc;                               // DMPNN_LOSS_CE: classes per task — P / gP rows hold t * nc logits, T the class index of every task
                                          // DMPNN_LOSS_MVE / _EVIDENTIAL: 2 / 4 — P / gP rows hold nc chunks of t columns (torch.chunk(Y, n_targets, 1))
    float v_kl, eps;                      // DMPNN_LOSS_EVIDENTIAL
    float q_alpha;                        // DMPNN_LOSS_QUANTILE
};
// QuantileLoss (nn/metrics.py:589-610) on the raw outputs of QuantileFFN (predictors.py:215-232): mean -+ interval / 2 ARE the lower and
// upper bounds the MLP put out; per bound amax(tau e, (tau - 1) e) with e = y - bound, tau = alpha / 2 | 1 - alpha / 2
// (its derivative in the bound: -tau for e > 0, 1 - tau for e < 0, their mean at e = 0: torch.amax shares the gradient among ties)
__device__ __forceinline__ float pinball(float e, float tau, float* g) {
    if (g) *g = e > 0.f ? -tau : (e < 0.f ? 1.f - tau : 0.5f - tau);
    return fmaxf(tau * e, (tau - 1.f) * e);
}
// F.softplus (beta 1, threshold 20) and its derivative
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float softplus_d(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }
// digamma for x >= 1 (alpha = softplus + 1): the recurrence up to x >= 6, then the asymptotic series (error < 1e-7 there)
__device__ __forceinline__ float digamma_f(float x) {
    float r = 0.f;
    while (x < 6.f) { r -= 1.f / x; x += 1.f; }
    const float i = 1.f / x, i2 = i * i;
    return r + logf(x) - 0.5f * i - i2 * (1.f / 12.f - i2 * (1.f / 120.f - i2 * (1.f / 252.f)));
}
// MVELoss (nn/metrics.py:203-219) on the raw outputs of MveFFN (predictors.py:173-190): unreduced loss, d / d mean, d / d raw variance
__device__ __forceinline__ float mve_loss(float mean, float raw, float y, float* g_mean, float* g_raw) {
    const float var = softplus_f(raw), d = mean - y;
    if (g_mean) {
        *g_mean = d / var;
        *g_raw = (0.5f / var - d * d / (2.f * var * var)) * softplus_d(raw);
    }
    return d * d / (2.f * var) + 0.5f * logf(6.283185307179586f * var);
}
// EvidentialLoss (nn/metrics.py:222-262) on the raw outputs of EvidentialFFN (predictors.py:193-212); g[4]: d / d (mean, raw v, raw alpha, raw beta)
__device__ __forceinline__ float evidential_loss(const float (&x)[4], float y, float v_kl, float eps, float* g) {
    const float v = softplus_f(x[1]), al = softplus_f(x[2]) + 1.f, be = softplus_f(x[3]);
    const float res = y - x[0], tbl = 2.f * be * (1.f + v), D = v * res * res + tbl, ares = fabsf(res);
    const float nll = 0.5f * logf(3.141592653589793f / v) - al * logf(tbl) + (al + 0.5f) * logf(D) + lgammaf(al) - lgammaf(al + 0.5f);
    const float reg = (2.f * v + al) * ares;
    if (g) {
        const float sg = res > 0.f ? 1.f : (res < 0.f ? -1.f : 0.f);
        g[0] = -((al + 0.5f) * 2.f * v * res / D + v_kl * (2.f * v + al) * sg);
        g[1] = (-0.5f / v - al / (1.f + v) + (al + 0.5f) * (res * res + 2.f * be) / D + v_kl * 2.f * ares) * softplus_d(x[1]);
        g[2] = (logf(D) - logf(tbl) + digamma_f(al) - digamma_f(al + 0.5f) + v_kl * ares) * softplus_d(x[2]);
        g[3] = (-al / be + (al + 0.5f) * 2.f * (1.f + v) / D) * softplus_d(x[3]);
    }
    return nll + v_kl * (reg - eps);
}
__global__ __launch_bounds__(1024) void k_loss(LossArgs a) {
    __shared__ float red[2][16];
    __shared__ float tot[2];
    const int64_t n = a.B * a.t;
    float sl = 0.f, sm = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int64_t r = i / a.t; const int j = (int)(i - r * a.t);
        const float y = a.T[r * a.ldt + j];
        const bool m = isfinite(y);
        if (!m) continue;
        if (a.kind == DMPNN_LOSS_CE) {   // (uniform) F.cross_entropy over the task's nc logits: logsumexp - x[class]   (metrics.py:298-304)
            const float* x = a.P + r * a.ldp + (int64_t)j * a.nc;
            float mx = x[0];
            for (int k = 1; k < a.nc; ++k) mx = fmaxf(mx, x[k]);
            float se = 0.f;
            for (int k = 0; k < a.nc; ++k) se += expf(x[k] - mx);
            const int cls = (int)y;
            const float L = (mx + logf(se)) - x[(cls >= 0 && cls < a.nc) ? cls : 0];
            sl += ((cls >= 0 && cls < a.nc) ? L : __int_as_float(0x7fc00000)) * (a.w ? a.w[r] : 1.f) * (a.tw ? a.tw[j] : 1.f);
            sm += 1.f;
            continue;
        }
        if (a.kind >= DMPNN_LOSS_MVE) {   // (uniform) chunked outputs: column k t + j is target k of task j
            const float* x = a.P + r * a.ldp + j;
            float L;
            if (a.kind == DMPNN_LOSS_MVE) L = mve_loss(x[0], x[a.t], y, nullptr, nullptr);
            else if (a.kind == DMPNN_LOSS_QUANTILE) L = pinball(y - x[0], 0.5f * a.q_alpha, nullptr) + pinball(y - x[a.t], 1.f - 0.5f * a.q_alpha, nullptr);
            else { const float xs[4] = {x[0], x[a.t], x[2 * a.t], x[3 * a.t]}; L = evidential_loss(xs, y, a.v_kl, a.eps, nullptr); }
            sl += L * (a.w ? a.w[r] : 1.f) * (a.tw ? a.tw[j] : 1.f);
            sm += 1.f;
            continue;
        }
        float p = a.P[r * a.ldp + j];
        if ((a.lt && a.lt[r * a.t + j] && p < y) || (a.gt && a.gt[r * a.t + j] && p > y)) p = y;
        const float L = loss_value(a.kind, p, y);
        sl += L * (a.w ? a.w[r] : 1.f) * (a.tw ? a.tw[j] : 1.f);
        sm += 1.f;
    }
    for (int off = 32; off > 0; off >>= 1) { sl += __shfl_xor(sl, off); sm += __shfl_xor(sm, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sl; red[1][threadIdx.x >> 6] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a0 = 0.f, a1 = 0.f;
        for (int i = 0; i < 16; ++i) { a0 += red[0][i]; a1 += red[1][i]; }
        tot[0] = a0; tot[1] = a1;
        a.out[0] = a0 / a1;   // (no finite target: 0 / 0 = NaN, like the reference)
        a.out[1] = a1;
    }
    __syncthreads();
    if (!a.gP) return;
    const float inv = 1.f / tot[1];
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int64_t r = i / a.t; const int j = (int)(i - r * a.t);
        const float y = a.T[r * a.ldt + j];
        if (a.kind == DMPNN_LOSS_CE) {   // (uniform) dL/dx_k = softmax_k - [k == class]
            const float* x = a.P + r * a.ldp + (int64_t)j * a.nc;
            float* gx = a.gP + r * a.ldg + (int64_t)j * a.nc;
            if (!isfinite(y)) {
                for (int k = 0; k < a.nc; ++k) gx[k] = 0.f;
                continue;
            }
            float mx = x[0];
            for (int k = 1; k < a.nc; ++k) mx = fmaxf(mx, x[k]);
            float se = 0.f;
            for (int k = 0; k < a.nc; ++k) se += expf(x[k] - mx);
            const float f = (a.w ? a.w[r] : 1.f) * (a.tw ? a.tw[j] : 1.f) * inv, ise = 1.f / se;
            const int cls = (int)y;
            for (int k = 0; k < a.nc; ++k) gx[k] = (expf(x[k] - mx) * ise - (k == cls ? 1.f : 0.f)) * f;
            continue;
        }
        if (a.kind >= DMPNN_LOSS_MVE) {   // (uniform)
            const float* x = a.P + r * a.ldp + j;
            float* gx = a.gP + r * a.ldg + j;
            float g4[4] = {0.f, 0.f, 0.f, 0.f};
            if (isfinite(y)) {
                if (a.kind == DMPNN_LOSS_MVE) mve_loss(x[0], x[a.t], y, &g4[0], &g4[1]);
                else if (a.kind == DMPNN_LOSS_QUANTILE) { pinball(y - x[0], 0.5f * a.q_alpha, &g4[0]); pinball(y - x[a.t], 1.f - 0.5f * a.q_alpha, &g4[1]); }
                else { const float xs[4] = {x[0], x[a.t], x[2 * a.t], x[3 * a.t]}; evidential_loss(xs, y, a.v_kl, a.eps, g4); }
            }
            const float f = (a.w ? a.w[r] : 1.f) * (a.tw ? a.tw[j] : 1.f) * inv;
            for (int k = 0; k < a.nc; ++k) gx[(int64_t)k * a.t] = g4[k] * f;
            continue;
        }
        float g = 0.f;
        if (isfinite(y)) {
            float p = a.P[r * a.ldp + j];
            if ((a.lt && a.lt[r * a.t + j] && p < y) || (a.gt && a.gt[r * a.t + j] && p > y)) p = y;
            const float dl = loss_deriv(a.kind, p, y);
            g = dl * (a.w ? a.w[r] : 1.f) * (a.tw ? a.tw[j] : 1.f) * inv;
        }
        a.gP[r * a.ldg + j] = g;
    }
}

// ---- the predictor's OUTPUT layer for a handful of tasks (n_tasks <= kOutMaxTasks: the usual regression head) ----------------
// P = A W^T + b with W [t, K]: t dot products per row — one wave per row, lanes over K (a 16 x 16 MFMA tile would be 1/16 full
// and the generic contraction kernel spends 17 us on its pipeline for 0.3 MFLOP).
constexpr int kOutMaxTasks = 4;
struct OutFwdArgs { const float* A; int64_t lda; const float* W; const float* b; float* P; int64_t B; int K, t; };
__global__ __launch_bounds__(256) void k_out_fwd(OutFwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.B) return;
    float acc[kOutMaxTasks] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < a.K; k += 64) {
        const float x = a.A[r * a.lda + k];
#pragma unroll
        for (int j = 0; j < kOutMaxTasks; ++j)
            if (j < a.t) acc[j] += x * a.W[(int64_t)j * a.K + k];
    }
#pragma unroll
    for (int j = 0; j < kOutMaxTasks; ++j) {
        float v = acc[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && j < a.t) a.P[r * a.t + j] = v + (a.b ? a.b[j] : 0.f);
    }
}
// Its whole backward in one launch (16 columns x 64 row lanes per workgroup, like the batch-norm kernels):
//   gA[r][k] = (sum_j gP[r][j] W[j][k]) tau'(A[r][k])      (A = tau(previous layer): the derivative from the output)
//   gW[j][k] = sum_r gP[r][j] A[r][k],   gb[j] = sum_r gP[r][j]
struct OutBwdArgs {
    const float* gP; const float* A; int64_t lda; const float* W; float* gA; int64_t ldga; float* gW; float* gb;
    int64_t B; int K, t, act; float slope;
};
__global__ __launch_bounds__(1024) void k_out_bwd(OutBwdArgs a) {
    __shared__ float red[kBnLanes][kBnCols];
    const int tx = threadIdx.x & (kBnCols - 1), ty = threadIdx.x / kBnCols;
    const int k = blockIdx.x * kBnCols + tx;
    const bool ok = k < a.K;
    float w[kOutMaxTasks], gw[kOutMaxTasks], gbs[kOutMaxTasks];
#pragma unroll
    for (int j = 0; j < kOutMaxTasks; ++j) { w[j] = (ok && j < a.t) ? a.W[(int64_t)j * a.K + k] : 0.f; gw[j] = 0.f; gbs[j] = 0.f; }
    for (int64_t r = ty; r < a.B; r += kBnLanes) {
        const float x = ok ? a.A[r * a.lda + k] : 0.f;
        float g = 0.f;
#pragma unroll
        for (int j = 0; j < kOutMaxTasks; ++j)
            if (j < a.t) { const float gp = a.gP[r * a.t + j]; g += gp * w[j]; gw[j] += gp * x; gbs[j] += gp; }
        if (ok && a.gA) a.gA[r * a.ldga + k] = g * act_grad_from_out(x, a.act, a.slope);
    }
    for (int j = 0; j < a.t; ++j) {
        const float sw = bn_col_sum(red, tx, ty, gw[j]);
        if (ok && ty == 0 && a.gW) a.gW[(int64_t)j * a.K + k] = sw;
        if (a.gb && blockIdx.x == 0) {     // (uniform per workgroup: every thread of workgroup 0 takes part in the reduction)
            const float sb = bn_col_sum(red, tx, ty, gbs[j]);
            if (tx == 0 && ty == 0) a.gb[j] = sb;
        }
    }
}

// Criterion + the output layer's backward in ONE launch (training, <= kOutAllMaxRows molecules).  Every workgroup (16 columns of
// the layer's input, like k_out_bwd) first forms dl/dP for ALL molecules in its own LDS from the predictions k_out_fwd wrote —
// B t values, cheaper to redo per workgroup than a launch of its own — then runs its slice of the backward on them.  Workgroup
// 0 writes the loss and the count.  Same arithmetic as the two kernels it replaces (k_loss's lane / wave order, k_out_bwd's
// column sums).  (Recomputing the predictions per workgroup as well was tried: 89 us — 16 waves walking 512 rows each is a
// latency chain, where k_out_fwd's one wave per row is 5 us.)
constexpr int64_t kOutAllMaxRows = 1024;
struct OutAllArgs {
    OutBwdArgs o;                   // (o.gP unused: dl/dP lives in LDS)
    LossArgs l;                     // (l.gP / l.ldg unused)
};
__global__ __launch_bounds__(1024) void k_out_all(OutAllArgs q) {
    __shared__ float Ps[kOutAllMaxRows * kOutMaxTasks];
    __shared__ float gPs[kOutAllMaxRows * kOutMaxTasks];
    __shared__ float red[kBnLanes][kBnCols];
    __shared__ float red2[2][16];
    __shared__ float tot[2];
    const OutBwdArgs& a = q.o;
    const LossArgs& L = q.l;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = a.t;
    for (int64_t i = threadIdx.x; i < a.B * t; i += 1024) Ps[i] = L.P[(i / t) * L.ldp + (i % t)];
    __syncthreads();
    // ---- criterion (k_loss's arithmetic) ----
    const int64_t n = a.B * t;
    float sl = 0.f, sm = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int64_t r = i / t; const int j = (int)(i - r * t);
        const float y = L.T[r * L.ldt + j];
        if (!isfinite(y)) continue;
        float p = Ps[i];
        if ((L.lt && L.lt[i] && p < y) || (L.gt && L.gt[i] && p > y)) p = y;
        const float Lv = loss_value(L.kind, p, y);
        sl += Lv * (L.w ? L.w[r] : 1.f) * (L.tw ? L.tw[j] : 1.f);
        sm += 1.f;
    }
    for (int off = 32; off > 0; off >>= 1) { sl += __shfl_xor(sl, off); sm += __shfl_xor(sm, off); }
    if (lane == 0) { red2[0][wave] = sl; red2[1][wave] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a0 = 0.f, a1 = 0.f;
        for (int i = 0; i < 16; ++i) { a0 += red2[0][i]; a1 += red2[1][i]; }
        tot[0] = a0; tot[1] = a1;
        if (blockIdx.x == 0) { L.out[0] = a0 / a1; L.out[1] = a1; }
    }
    __syncthreads();
    const float inv = 1.f / tot[1];
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int64_t r = i / t; const int j = (int)(i - r * t);
        const float y = L.T[r * L.ldt + j];
        float g = 0.f;
        if (isfinite(y)) {
            float p = Ps[i];
            if ((L.lt && L.lt[i] && p < y) || (L.gt && L.gt[i] && p > y)) p = y;
            const float dl = loss_deriv(L.kind, p, y);
            g = dl * (L.w ? L.w[r] : 1.f) * (L.tw ? L.tw[j] : 1.f) * inv;
        }
        gPs[i] = g;
    }
    __syncthreads();
    // ---- the layer's backward on this workgroup's 16 columns (k_out_bwd's arithmetic) ----
    const int tx = threadIdx.x & (kBnCols - 1), ty = threadIdx.x / kBnCols;
    const int k = blockIdx.x * kBnCols + tx;
    const bool ok = k < a.K;
    float w[kOutMaxTasks], gw[kOutMaxTasks], gbs[kOutMaxTasks];
#pragma unroll
    for (int j = 0; j < kOutMaxTasks; ++j) { w[j] = (ok && j < t) ? a.W[(int64_t)j * a.K + k] : 0.f; gw[j] = 0.f; gbs[j] = 0.f; }
    for (int64_t r = ty; r < a.B; r += kBnLanes) {
        const float x = ok ? a.A[r * a.lda + k] : 0.f;
        float g = 0.f;
#pragma unroll
        for (int j = 0; j < kOutMaxTasks; ++j)
            if (j < t) { const float gp = gPs[r * t + j]; g += gp * w[j]; gw[j] += gp * x; gbs[j] += gp; }
        if (ok && a.gA) a.gA[r * a.ldga + k] = g * act_grad_from_out(x, a.act, a.slope);
    }
    for (int j = 0; j < t; ++j) {
        const float sw = bn_col_sum(red, tx, ty, gw[j]);
        if (ok && ty == 0 && a.gW) a.gW[(int64_t)j * a.K + k] = sw;
        if (a.gb && blockIdx.x == 0) {
            const float sb = bn_col_sum(red, tx, ty, gbs[j]);
            if (tx == 0 && ty == 0) a.gb[j] = sb;
        }
    }
}

// =====================================================================================================================
// Round 5: the head of a training step in FOUR launches (was nine) for the usual predictor — one hidden layer, <= kOutMaxTasks
// outputs, <= kRowsMaxB molecules, widths <= kRowsMaxWidth (five launches beyond kFuseAggMols molecules: the aggregation in front):
//   k_agg_bn_fwd        per 8 | 16 columns: H = agg(H_v) (the rows added in increasing atom order, like k_mol_reduce) and Z = bn(H)
//                       on the values still in registers; the workgroups behind those SPLIT the hidden layer's weight (fragment-major
//                       hi | lo, both orientations) — the split rides in this launch like the block's rides in K0: nothing about
//                       weights is cached;
//   k_head_rows<., 1>   per (16 molecules, 64 columns): A1 = tau(Z W0^T + b0) on the f16 pipe (3-product split, fp32 accumulation:
//                       the block's arithmetic);
//   k_head_rows<., 2>   per (16 molecules, 64 columns), everything that is local to a row from the whole rows of A1: P = A1 W1^T + b1,
//                       the criterion (every workgroup counts the finite targets of the WHOLE batch itself: B t values), dl/dP,
//                       dl/dA1, then its slice of dl/dZ = dl/dA1 . W0 on the f16 pipe; what sums over rows (gW1, gb1, the loss)
//                       leaves as one partial per row block;
//   k_bn_agg_bwd        per 8 | 16 columns: batch norm backward and the broadcast of dl/dH to the atoms' rows (k_mol_bwd's
//                       arithmetic); one more workgroup sums the partials in a fixed order (deterministic).
// The hidden layer's weight gradient rides in the block's backward launches as before (ExtraWgrad) or runs as its own product.
// What set the shapes (profiles/r05_head_*): a CU pulls ~55 GB/s — whole rows per workgroup (32 workgroups, 820 KB of weight
// fragments each) took 27-33 us, 16 columns x all atoms per workgroup (19 workgroups) 11 us for the aggregation alone; and a
// conditional load `ok ? p[i] : 0` compiles to a branch with a full wait behind it — every scalar below is a buffer load instead.
// DMPNN_HEAD=chain (environment, read per call): the nine-launch chain of rounds 3-4, kept for every other shape; =rows: a training
// call that would take the chain is an error (tests).
constexpr int64_t kRowsMaxB = 1024;      // (4 molecules per thread of the column kernels)
constexpr int64_t kFuseAggMols = 512;    // batches up to this size are aggregated inside k_agg_bn_fwd
constexpr int kRowsMaxWidth = 320;       // WN <= 5 column tiles per wave
typedef float f32x4 __attribute__((ext_vector_type(4)));
using mega16::h4;
using mega16::h8;
using mega16::SplitW;

// Buffer loads for the column kernels' scalars: an offset out of range (or a NULL array: a zero-length buffer) reads 0 — no branch,
// no select on the loaded value, so a batch of requests goes out before the first wait (a conditional `ok ? p[i] : 0` compiles to a
// branch with a full wait behind EVERY load: 16 serialised round trips for a thread's molecule bounds, 11 us).
__device__ __forceinline__ gemm::rsrc_t col_buf(const void* p, int64_t bytes, const void* dummy) {
    return gemm::make_rsrc(p ? p : dummy, p ? gemm::clamp_bytes(bytes) : 0u);
}
__device__ __forceinline__ float col_ldf(gemm::rsrc_t r, bool ok, int64_t idx) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, ok ? (unsigned)idx * 4u : gemm::kOOB, 0, 0));
}
__device__ __forceinline__ int col_ldi(gemm::rsrc_t r, bool ok, int64_t idx) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, ok ? (unsigned)idx * 4u : gemm::kOOB, 0, 0);
}
// W0 [N, K] (nn.Linear layout) -> the two operands of the row kernel, both in the tile kernels' fragment-major order
// [column tile][chunk][hi | lo][lg][li][8 halfs] (mega16::split_weights_wave) and both from ONE power-of-two scale per row n:
//   fwd: column index n, reduction index k — the operand of Z . W0^T;       inv_scale[n] = 1 / s_n
//   bwd: column index k, reduction index n — the operand of dl/dA1 . W0, holding the SAME halves W0[n][k] s_n: the scale belongs to
//        the reduction index there, so the row kernel folds 1 / s_n into the columns of dl/dA1 before it splits them (exact).
// One workgroup per 32 rows n (= one reduction chunk of `bwd`, two column tiles of `fwd`): row maxima by the waves (coalesced),
// then every thread converts 8 consecutive k of a row (fwd) / 8 consecutive n of a column (bwd: lanes along k, coalesced) into one
// 16-byte store each.  (split_weights_wave with tr = 1 reads a column per wave — 64 cache lines per load instruction: 20 us here.)
struct HeadSplit { const float* W; int N, K; unsigned char* fwd; unsigned char* bwd; float* inv_scale; int ncf, ncb; };
__device__ __forceinline__ void head_split_block(const HeadSplit& a, int blk) {
    __shared__ float sc[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // 1024 threads
    const int n0 = blk * 32;
    const gemm::rsrc_t rW = gemm::make_rsrc(a.W, (unsigned)(a.N * a.K * 4));
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int n = n0 + 2 * wave + rr;
        float mx = 0.f;
#pragma unroll
        for (int u = 0; u < kRowsMaxWidth / 64; ++u) {   // (buffer loads: out of range reads 0 — no branch, all in flight together)
            const int k = lane + 64 * u;
            mx = fmaxf(mx, fabsf(col_ldf(rW, n < a.N && k < a.K, (int64_t)n * a.K + k)));
        }
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        const float sn = n < a.N ? mega16::scale_for(mx) : 0.f;
        if (lane == 0) {
            sc[2 * wave + rr] = sn;
            if (n < a.N) a.inv_scale[n] = 1.f / sn;
        }
    }
    __syncthreads();
    auto put = [&](unsigned char* base, int T, int nc, int c, int lg, int li, const float (&x)[8]) {
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)x[j]; lo[j] = (_Float16)(x[j] - (float)hi[j]); }
        _Float16* o = reinterpret_cast<_Float16*>(base) + (((int64_t)T * nc + c) * 2) * 512 + (lg * 16 + li) * 8;
        *reinterpret_cast<h8*>(o) = hi;
        *reinterpret_cast<h8*>(o + 512) = lo;
    };
    const int nk8 = a.ncf * 4, last_tile = (a.N - 1) >> 4;
    for (int it = tid; it < 32 * nk8; it += 1024) {
        const int nl = it / nk8, k8 = it - nl * nk8, n = n0 + nl, k = 8 * k8;
        if ((n >> 4) > last_tile) continue;   // (rows past N inside the last 16-row tile are written as zeros; tiles past it do not exist)
        const float sn = sc[nl];
        const bool l0 = n < a.N && k < a.K, l1 = n < a.N && k + 4 < a.K;   // (K % 4 == 0)
        const float4 v0 = gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(rW, l0 ? (unsigned)(n * a.K + k) * 4u : gemm::kOOB, 0, 0));
        const float4 v1 = gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(rW, l1 ? (unsigned)(n * a.K + k + 4) * 4u : gemm::kOOB, 0, 0));
        const float x[8] = {v0.x * sn, v0.y * sn, v0.z * sn, v0.w * sn, v1.x * sn, v1.y * sn, v1.z * sn, v1.w * sn};
        put(a.fwd, n >> 4, a.ncf, k8 >> 2, k8 & 3, n & 15, x);
    }
    const int Kp = (a.K + 15) & ~15;
    for (int it = tid; it < Kp * 4; it += 1024) {
        const int lg = it / Kp, k = it - lg * Kp;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + 8 * lg + j;
            x[j] = col_ldf(rW, n < a.N && k < a.K, (int64_t)n * a.K + k) * sc[8 * lg + j];
        }
        put(a.bwd, k >> 4, a.ncb, blk, lg, k & 15, x);
    }
}

struct AggBnArgs {
    BnArgs b;                              // X = H [B, d] (written here), Y = Z (BN only)
    const float* Hv; int64_t ldhv; int64_t nV; const int* bounds; int agg_mode; float agg_norm;
    int use_done;                          // 1: bounds[2 B + 4 + m] != 0 — X already holds molecule m's aggregate (the forward tile kernel wrote it)
    int n_col_blocks;                      // workgroups [0, n_col_blocks): columns; the others: the weight split (32 rows of W0 each)
    HeadSplit split;
    long long* dbg;                        // optional cycle stamps: [0..5] column workgroup 1, [6..9] the first split workgroup
};
// Geometry of the two column kernels: a workgroup = QPW column QUADS (16-byte loads and stores: a quarter of the memory instructions
// of a thread-per-column layout, whose 1 500 four-segment wave loads per CU cost the aggregation 11 us) x 1024 / QPW row lanes; thread
// (tq = tid % QPW, ty = tid / QPW) holds RR molecules (B <= RR 1024 / QPW) of its quad in registers.  QPW shrinks as the batch grows:
// a CU moves ~55 GB/s, and what these kernels move is H_v (aggregation) / dl/dH_v (broadcast) — 5.5 MB at 512 molecules, so the
// columns are spread over 38 workgroups there (2 quads each) where 64 molecules do with 19 (4 quads).
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// sums over the row lanes of every column: the row lanes of a wave by shuffles, the 16 waves through LDS
template <int QPW>
__device__ __forceinline__ float4 quad_col_sum(float4 (*red)[4], int tq, float4 v) {
#pragma unroll
    for (int off = QPW; off < 64; off <<= 1) {
        v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off); v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
    }
    __syncthreads();   // (the previous use of `red` is over)
    if ((threadIdx.x & 63) < QPW) red[threadIdx.x >> 6][tq] = v;
    __syncthreads();
    float4 s = red[0][tq];
#pragma unroll 4
    for (int w = 1; w < 16; ++w) s = f4_add(s, red[w][tq]);
    return s;
}
// two sums at once (batch norm backward: sum gY and sum gY xhat): one pair of barriers
template <int QPW>
__device__ __forceinline__ void quad_col_sum2(float4 (*red)[2][4], int tq, float4& u, float4& v) {
#pragma unroll
    for (int off = QPW; off < 64; off <<= 1) {
        u.x += __shfl_xor(u.x, off); u.y += __shfl_xor(u.y, off); u.z += __shfl_xor(u.z, off); u.w += __shfl_xor(u.w, off);
        v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off); v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
    }
    if ((threadIdx.x & 63) < QPW) { red[threadIdx.x >> 6][0][tq] = u; red[threadIdx.x >> 6][1][tq] = v; }
    __syncthreads();
    u = red[0][0][tq]; v = red[0][1][tq];
#pragma unroll 4
    for (int w = 1; w < 16; ++w) { u = f4_add(u, red[w][0][tq]); v = f4_add(v, red[w][1][tq]); }
}
template <int QPW, int RR, bool BN>
__global__ __launch_bounds__(1024) void k_agg_bn_fwd(AggBnArgs q) {
    constexpr int kQLanes = 1024 / QPW;
    if ((int)blockIdx.x >= q.n_col_blocks) {   // (uniform per workgroup)
        const bool st = q.dbg && (int)blockIdx.x == q.n_col_blocks && threadIdx.x == 0;
        if (st) q.dbg[6] = (long long)__builtin_readcyclecounter();
        head_split_block(q.split, (int)blockIdx.x - q.n_col_blocks);
        if (st) q.dbg[7] = (long long)__builtin_readcyclecounter();
        return;
    }
    int n_stamp = 0;
    auto stamp = [&]() {
        if (q.dbg && blockIdx.x == 1 && threadIdx.x == 0 && n_stamp < 6) q.dbg[n_stamp] = (long long)__builtin_readcyclecounter();
        ++n_stamp;
    };
    stamp();  // 0 entry
    __shared__ float4 red[16][4];
    const BnArgs& a = q.b;
    const int tq = threadIdx.x % QPW, ty = threadIdx.x / QPW;
    const int c = (blockIdx.x * QPW + tq) * 4;
    const bool ok = c < a.d;   // (d % 4 == 0: a quad is inside or outside)
    const gemm::rsrc_t rBd = gemm::make_rsrc(q.bounds, (unsigned)((3 * a.B + 4) * 4));
    const int flag = col_ldi(rBd, true, 2 * a.B);
    int v0[RR], nv[RR], dn[RR];
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        const int64_t r = ty + (int64_t)kQLanes * i;
        v0[i] = col_ldi(rBd, r < a.B, r);
        nv[i] = col_ldi(rBd, r < a.B, a.B + r);
        dn[i] = col_ldi(rBd, q.use_done && r < a.B, 2 * a.B + 4 + r);
    }
    // (batch norm's per-column constants: requested now, used behind the statistics)
    auto ldq = [&](const float* p) {
        return gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(col_buf(p, (int64_t)a.d * 4, q.bounds), ok ? (unsigned)c * 4u : gemm::kOOB, 0, 0));
    };
    const float4 gam0 = ldq(a.gamma), bet0 = ldq(a.beta), rm0 = ldq(a.run_mean), rv0 = ldq(a.run_var);
    stamp();  // 1 requests out
    int nmax = 0;
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        nv[i] -= v0[i];
        if (nv[i] < 0 || !ok || dn[i]) nv[i] = 0;
        nmax = nv[i] > nmax ? nv[i] : nmax;
    }
    stamp();  // 2 bounds here
    // segment sums: AT atoms of every one of the thread's molecules per round — AT RR independent 16-byte loads in flight (a molecule's
    // rows are still added in increasing atom order: the reference's scatter order)
    constexpr int AT = 16 / RR;
    const gemm::rsrc_t rH = gemm::make_rsrc(q.Hv, gemm::clamp_bytes(((int64_t)q.nV * q.ldhv) * 4));
    float4 xs[RR];
#pragma unroll
    for (int i = 0; i < RR; ++i) xs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const gemm::rsrc_t rX = gemm::make_rsrc(a.X, gemm::clamp_bytes(a.B * a.ldx * 4));
    if (!q.Hv) {   // (uniform) the aggregate is given (dmpnn_molagg_fwd ran in front: batches whose H_v 19 CUs cannot pull fast enough)
        if constexpr (!BN) return;
#pragma unroll
        for (int i = 0; i < RR; ++i) {
            const int64_t r = ty + (int64_t)kQLanes * i;
            xs[i] = gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(rX, (ok && r < a.B) ? (unsigned)(r * a.ldx + c) * 4u : gemm::kOOB, 0, 0));
        }
        nmax = 0;
    } else if (q.use_done) {   // (uniform) ... or given for the molecules the forward tile kernel carried (done): the others are summed below
        float4 hx[RR];
#pragma unroll
        for (int i = 0; i < RR; ++i) {
            const int64_t r = ty + (int64_t)kQLanes * i;
            hx[i] = gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(rX, (ok && r < a.B && dn[i]) ? (unsigned)(r * a.ldx + c) * 4u : gemm::kOOB, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < RR; ++i) xs[i] = hx[i];
    }
    for (int s0 = 0; s0 < nmax; s0 += AT) {
        float4 tv[AT][RR];
#pragma unroll
        for (int u = 0; u < AT; ++u)
#pragma unroll
            for (int i = 0; i < RR; ++i)
                tv[u][i] = gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(
                    rH, s0 + u < nv[i] ? (unsigned)((int64_t)(v0[i] + s0 + u) * q.ldhv + c) * 4u : gemm::kOOB, 0, 0));
#pragma unroll
        for (int u = 0; u < AT; ++u)
#pragma unroll
            for (int i = 0; i < RR; ++i)
                if (s0 + u < nv[i]) xs[i] = (s0 + u == 0) ? tv[u][i] : f4_add(xs[i], tv[u][i]);   // include_self=False: the first addend is copied
    }
    const float nanv = __int_as_float(0x7fc00000);
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        const int64_t r = ty + (int64_t)kQLanes * i;
        float4 y = xs[i];
        if (!q.Hv) break;
        if (!dn[i]) {   // (a molecule the tile kernel wrote arrives divided)
            if (q.agg_mode == DMPNN_MOLAGG_MEAN && nv[i] > 0) { const float n = (float)nv[i]; y = make_float4(y.x / n, y.y / n, y.z / n, y.w / n); }
            if (q.agg_mode == DMPNN_MOLAGG_NORM) y = make_float4(y.x / q.agg_norm, y.y / q.agg_norm, y.z / q.agg_norm, y.w / q.agg_norm);
        }
        if (flag) y = make_float4(nanv, nanv, nanv, nanv);
        xs[i] = (ok && r < a.B) ? y : make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && r < a.B && (!dn[i] || flag)) *reinterpret_cast<float4*>(a.X_out() + r * a.ldx + c) = y;
    }
    stamp();  // 3 aggregated
    if constexpr (!BN) return;
    float4 mean, invstd;
    if (a.training) {
        if (a.n_tracked && blockIdx.x == 0 && threadIdx.x == 0) *a.n_tracked += 1;
        float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RR; ++i) sm = f4_add(sm, xs[i]);
        sm = quad_col_sum<QPW>(red, tq, sm);
        const float fB = (float)a.B;
        mean = make_float4(sm.x / fB, sm.y / fB, sm.z / fB, sm.w / fB);
        float4 qq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RR; ++i)
            if (ty + (int64_t)kQLanes * i < a.B) {
                const float4 dl = make_float4(xs[i].x - mean.x, xs[i].y - mean.y, xs[i].z - mean.z, xs[i].w - mean.w);
                qq = f4_add(qq, make_float4(dl.x * dl.x, dl.y * dl.y, dl.z * dl.z, dl.w * dl.w));
            }
        const float4 ss = quad_col_sum<QPW>(red, tq, qq);
        const float4 var = make_float4(ss.x / fB, ss.y / fB, ss.z / fB, ss.w / fB);   // biased: what normalises (nn.BatchNorm1d)
        invstd = make_float4(1.f / sqrtf(var.x + a.eps), 1.f / sqrtf(var.y + a.eps), 1.f / sqrtf(var.z + a.eps), 1.f / sqrtf(var.w + a.eps));
        if (ok && ty == 0) {
            *reinterpret_cast<float4*>(a.save_mean + c) = mean;
            *reinterpret_cast<float4*>(a.save_invstd + c) = invstd;
            const float m = a.momentum, om = 1.f - a.momentum, fB1 = (float)(a.B - 1);
            if (a.run_mean) *reinterpret_cast<float4*>(a.run_mean + c) = make_float4(om * rm0.x + m * mean.x, om * rm0.y + m * mean.y, om * rm0.z + m * mean.z, om * rm0.w + m * mean.w);
            // running_var takes the UNBIASED estimate (torch: var * B / (B - 1))
            const float4 ub = a.B > 1 ? make_float4(ss.x / fB1, ss.y / fB1, ss.z / fB1, ss.w / fB1) : var;
            if (a.run_var) *reinterpret_cast<float4*>(a.run_var + c) = make_float4(om * rv0.x + m * ub.x, om * rv0.y + m * ub.y, om * rv0.z + m * ub.z, om * rv0.w + m * ub.w);
        }
    } else {
        mean = rm0;
        invstd = make_float4(1.f / sqrtf(rv0.x + a.eps), 1.f / sqrtf(rv0.y + a.eps), 1.f / sqrtf(rv0.z + a.eps), 1.f / sqrtf(rv0.w + a.eps));
    }
    stamp();  // 4 statistics
    if (ok) {
        const float4 g = a.gamma ? gam0 : make_float4(1.f, 1.f, 1.f, 1.f), bb = a.beta ? bet0 : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RR; ++i) {
            const int64_t r = ty + (int64_t)kQLanes * i;
            if (r < a.B)
                *reinterpret_cast<float4*>(a.Y + r * a.ldy + c) = make_float4((xs[i].x - mean.x) * invstd.x * g.x + bb.x, (xs[i].y - mean.y) * invstd.y * g.y + bb.y,
                                                                              (xs[i].z - mean.z) * invstd.z * g.z + bb.z, (xs[i].w - mean.w) * invstd.w * g.w + bb.w);
        }
    }
    stamp();  // 5 end
}

struct BnAggBwdArgs {
    BnBwdArgs b;                           // gY = dl/dZ [B, d], X = H; gX unused (the rows go to the atoms)
    float* gHv; int64_t ldg; const int* bounds; int64_t nV; int agg_mode; float agg_norm;
    // the row kernel's partials (or part == NULL): out[i] = sum over workgroups of part[w * part_stride + i]
    const float* part; int n_part, part_stride, tN, t;
    float* gW1; float* gb1; float* loss_out;
    long long* dbg;                        // optional cycle stamps of column workgroup 1
};
template <int QPW, int RR, bool BN>
__global__ __launch_bounds__(1024) void k_bn_agg_bwd(BnAggBwdArgs q) {
    constexpr int kQLanes = 1024 / QPW;
    __shared__ float4 red[16][2][4];
    if ((int)blockIdx.x * 4 * QPW >= q.b.d) {   // the workgroup behind the columns': what sums over the rows of the batch, in workgroup order
        if (!q.part) return;
        // four threads per output, each over a quarter of the row blocks in order, joined as ((q0 + q1) + (q2 + q3)): a fixed tree
        const int n_out = q.tN + q.t, sub = threadIdx.x & 3, per = (q.n_part + 3) / 4;
        for (int i0 = 0; i0 <= n_out; i0 += 256) {
            const int i = i0 + (threadIdx.x >> 2);
            const gemm::rsrc_t rP = gemm::make_rsrc(q.part, gemm::clamp_bytes((int64_t)q.n_part * q.part_stride * 4));
            float v[16];   // (n_part <= kRowsMaxB / 16 = 64 row blocks)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int w = sub * per + u;
                v[u] = col_ldf(rP, i <= n_out && u < per && w < q.n_part, (int64_t)w * q.part_stride + i);
            }
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (sub != 0 || i > n_out) continue;
            if (i < q.tN) { if (q.gW1) q.gW1[i] = s; }
            else if (i < n_out) { if (q.gb1) q.gb1[i - q.tN] = s; }
            else {   // the loss: sum / count (no finite target: 0 / 0 = NaN, like the reference)
                const float cnt = q.part[n_out + 1];
                q.loss_out[0] = s / cnt;
                q.loss_out[1] = cnt;
            }
        }
        return;
    }
    const BnBwdArgs& a = q.b;
    const int tq = threadIdx.x % QPW, ty = threadIdx.x / QPW;
    const int c = (blockIdx.x * QPW + tq) * 4;
    const bool ok = c < a.d;
    int n_stamp = 0;
    auto stamp = [&]() {
        if (q.dbg && blockIdx.x == 1 && threadIdx.x == 0 && n_stamp < 6) q.dbg[n_stamp] = (long long)__builtin_readcyclecounter();
        ++n_stamp;
    };
    stamp();  // 0 entry
    // every request of the kernel first (buffer loads: no branch, no wait in between)
    float4 gs[RR], xr[RR];
    int v0[RR], v1[RR];
    const gemm::rsrc_t rBd = gemm::make_rsrc(q.bounds, (unsigned)((3 * a.B + 4) * 4));
    const gemm::rsrc_t rG = gemm::make_rsrc(a.gY, gemm::clamp_bytes(a.B * a.ldgy * 4)), rX = gemm::make_rsrc(a.X, gemm::clamp_bytes(a.B * a.ldx * 4));
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        const int64_t r = ty + (int64_t)kQLanes * i;
        const bool in = ok && r < a.B;
        gs[i] = gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(rG, in ? (unsigned)(r * a.ldgy + c) * 4u : gemm::kOOB, 0, 0));
        xr[i] = BN ? gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(rX, in ? (unsigned)(r * a.ldx + c) * 4u : gemm::kOOB, 0, 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
        v0[i] = col_ldi(rBd, r < a.B, r);
        v1[i] = col_ldi(rBd, r < a.B, a.B + r);
    }
    const int flag = col_ldi(rBd, true, 2 * a.B);
    auto ldq = [&](const float* p) {
        return gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(col_buf(p, (int64_t)a.d * 4, q.bounds), (BN && ok) ? (unsigned)c * 4u : gemm::kOOB, 0, 0));
    };
    const float4 mean = ldq(a.training ? a.save_mean : a.run_mean), is0 = ldq(a.training ? a.save_invstd : a.run_var), gam0 = ldq(a.gamma);
    stamp();  // 1 requests out
    if constexpr (BN) {
        const float4 invstd = a.training ? is0 : make_float4(1.f / sqrtf(is0.x + a.eps), 1.f / sqrtf(is0.y + a.eps), 1.f / sqrtf(is0.z + a.eps), 1.f / sqrtf(is0.w + a.eps));
        float4 xh[RR], s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
        for (int i = 0; i < RR; ++i) {
            const bool in = ok && ty + (int64_t)kQLanes * i < a.B;
            xh[i] = in ? make_float4((xr[i].x - mean.x) * invstd.x, (xr[i].y - mean.y) * invstd.y, (xr[i].z - mean.z) * invstd.z, (xr[i].w - mean.w) * invstd.w)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
            s1 = f4_add(s1, gs[i]);
            s2 = f4_add(s2, make_float4(gs[i].x * xh[i].x, gs[i].y * xh[i].y, gs[i].z * xh[i].z, gs[i].w * xh[i].w));
        }
        stamp();  // 2 data here
        quad_col_sum2<QPW>(red, tq, s1, s2);
        stamp();  // 3 column sums
        if (ok && ty == 0) {
            if (a.g_gamma) *reinterpret_cast<float4*>(a.g_gamma + c) = s2;
            if (a.g_beta) *reinterpret_cast<float4*>(a.g_beta + c) = s1;
        }
        const float4 gam = a.gamma ? gam0 : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 k = make_float4(gam.x * invstd.x, gam.y * invstd.y, gam.z * invstd.z, gam.w * invstd.w);
        const float invB = 1.f / (float)a.B;
#pragma unroll
        for (int i = 0; i < RR; ++i)
            gs[i] = a.training ? make_float4(k.x * (gs[i].x - invB * s1.x - xh[i].x * invB * s2.x), k.y * (gs[i].y - invB * s1.y - xh[i].y * invB * s2.y),
                                             k.z * (gs[i].z - invB * s1.z - xh[i].z * invB * s2.z), k.w * (gs[i].w - invB * s1.w - xh[i].w * invB * s2.w))
                               : make_float4(k.x * gs[i].x, k.y * gs[i].y, k.z * gs[i].z, k.w * gs[i].w);
    }
    if (!ok) return;
    const float nanv = __int_as_float(0x7fc00000);
    if (flag) {   // an invalid `batch`: every row NaN (atoms outside every molecule's range included)
        for (int64_t v = ty; v < q.nV; v += kQLanes) *reinterpret_cast<float4*>(q.gHv + v * q.ldg + c) = make_float4(nanv, nanv, nanv, nanv);
        return;
    }
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        const int64_t r = ty + (int64_t)kQLanes * i;
        if (r >= a.B) continue;
        float4 g = gs[i];
        if (q.agg_mode == DMPNN_MOLAGG_MEAN) { const float n = (float)(v1[i] - v0[i]); g = make_float4(g.x / n, g.y / n, g.z / n, g.w / n); }
        if (q.agg_mode == DMPNN_MOLAGG_NORM) g = make_float4(g.x / q.agg_norm, g.y / q.agg_norm, g.z / q.agg_norm, g.w / q.agg_norm);
        for (int v = v0[i]; v < v1[i]; ++v) *reinterpret_cast<float4*>(q.gHv + (int64_t)v * q.ldg + c) = g;
    }
    stamp();  // 4 (2 without batch norm) rows issued
}

struct RowsArgs {
    const float* Z; int64_t ldz;           // [B, K] the predictor's input
    SplitW W0f, W0b;                       // W0 [N, K] as N rows over K (forward) and as K rows over N (data gradient; see HeadSplit)
    const float* b0; const float* W1; const float* b1;   // [N] | NULL, [t, N], [t] | NULL
    float* A1;                             // [B, N] the hidden layer's output tau(Z W0^T + b0): written by PH 1, read by PH 2
    float* preds;                          // [B, t]
    const float* T; const float* w; const float* tw; const unsigned char* lt; const unsigned char* gt; int kind;
    float* gA1; float* gZ;                 // [B, N], [B, K]
    float* part; int part_stride;          // per row block: gW1 [t][N] | gb1 [t] | loss sum | number of finite targets
    int64_t B; int N, K, t, act; float slope;
    long long* dbg;                        // optional cycle stamps of workgroup (1, 1)
};
// The predictor + criterion + their backward as TWO launches over (row block of 16 molecules) x (slice of 64 columns):
//   PH 1  A1[:, slice] = tau(Z W0^T + b0)                                                          160 workgroups at 512 x 300
//   PH 2  everything that is local to a row, redone by each of the row block's slices from the whole rows of A1 (16 x N values:
//         cheaper than a launch): P = A1 W1^T + b1, the criterion (every workgroup counts the finite targets of the WHOLE batch
//         itself), dl/dP, dl/dA1 — then ITS slice of dl/dZ = dl/dA1 . W0, of dl/dA1 (for the hidden layer's weight gradient) and
//         of the row block's partial of gW1; slice 0 also writes the predictions and the partials of gb1 and of the loss.
// Both contractions on the f16 pipe (3-product split, fp32 accumulation: the block's arithmetic), the weight fragments of a wave's 16
// columns requested in one go at kernel entry.  (ONE launch per row block holding whole rows was built first: 27-33 us — a CU pulls
// its 820 KB of weight fragments at ~55 GB/s, profiles/r05_head_rowblock_stamps.txt; sliced, a workgroup needs 82 KB per contraction.)
template <int WNT, int PH>
__global__ __launch_bounds__(256) void k_head_rows(RowsArgs a) {
    constexpr int BN = 64 * WNT, NCH = BN / 32, TS = NCH * 128 + 16, LDA = BN + 4;
    __shared__ __attribute__((aligned(16))) unsigned char As[16 * TS];   // split operand tile: the rows of Z (PH 1) / of dl/dA1 (PH 2)
    __shared__ __attribute__((aligned(16))) float A1s[PH == 2 ? 16 * LDA : 4];   // fp32 tile: A1, later dl/dA1
    __shared__ float inv_s[16];                                          // 1 / scale of the split tile's rows
    __shared__ float Ps[16][kOutMaxTasks];
    __shared__ __attribute__((aligned(16))) float gPs[16][kOutMaxTasks];
    __shared__ float W1s[kOutMaxTasks][PH == 2 ? BN : 1];                // W1 [t, N], zero beyond
    __shared__ float red2[2][4];
    __shared__ float tot[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int cs = blockIdx.y;                  // column slice; wave w owns the 16 columns of tile T
    const int T = cs * 4 + wave, col = T * 16 + li;
    const int nrows = (int)(a.B - row0 < 16 ? a.B - row0 : 16);
    const int t = a.t;
    int n_stamp = 0;
    auto stamp = [&]() {
        if (a.dbg && blockIdx.x == 1 && blockIdx.y == 1 && threadIdx.x == 0 && n_stamp < 8) a.dbg[n_stamp] = (long long)__builtin_readcyclecounter();
        ++n_stamp;
    };
    stamp();  // 0 entry
    // (buffer loads: an offset out of range reads 0 without a branch or a select on the data — nothing in the request section consumes a
    //  loaded value, so every request is out before the first wait; a NULL array is a zero-length buffer)
    auto buf = [&](const void* ptr, int64_t bytes) { return gemm::make_rsrc(ptr ? ptr : a.Z, ptr ? (unsigned)bytes : 0u); };
    auto ldf = [&](gemm::rsrc_t r, bool ok, int64_t idx) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, ok ? (unsigned)idx * 4u : gemm::kOOB, 0, 0));
    };
    // rows of an fp32 [B, width] tensor: thread (row i = tid >> 4, p = tid & 15) takes columns 4 p + 64 j
    auto load_rows = [&](const float* X, int64_t ld, int width, float4 (&x)[WNT]) {
        const int i = tid >> 4, p = tid & 15;
        const gemm::rsrc_t rX = gemm::make_rsrc(X + row0 * ld, (unsigned)(nrows * ld * 4));
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            const int c4 = 4 * p + 64 * j;
            x[j] = gemm::as_f4(__builtin_amdgcn_raw_buffer_load_b128(rX, (i < nrows && c4 < width) ? (unsigned)(i * ld + c4) * 4u : gemm::kOOB, 0, 0));
        }
    };
    // fp32 rows -> split tile (the row's maximum by four shuffles)
    auto split_rows = [&](const float4 (&x)[WNT]) {
        const int i = tid >> 4, p = tid & 15;
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < WNT; ++j)
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x[j].x), fabsf(x[j].y))), fmaxf(fabsf(x[j].z), fabsf(x[j].w)));
        for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        const float sr = mega16::scale_for(mx);
        if (p == 0) inv_s[i] = 1.f / sr;
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            const int c4 = 4 * p + 64 * j;
            h4 hi, lo;
            mega16::split4(x[j], sr, hi, lo);
            unsigned char* d = As + i * TS + (c4 >> 5) * 128 + (c4 & 31) * 2;
            *reinterpret_cast<h4*>(d) = hi;
            *reinterpret_cast<h4*>(d + 64) = lo;
        }
    };
    // the wave's weight fragments: every chunk of column tile T (chunks beyond W.nc and tiles beyond n_out: out of range, 0)
    h8 wh[NCH], wl[NCH];
    auto request = [&](const SplitW& W, int n_out) {
        const gemm::rsrc_t rW = gemm::make_rsrc(W.p, (unsigned)(((n_out + 15) / 16) * W.nc * 2048));
        const unsigned o0 = (unsigned)T * (unsigned)(W.nc * 2048) + (unsigned)lane * 16u;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const unsigned o = (c < W.nc && T * 16 < n_out) ? o0 + (unsigned)c * 2048u : gemm::kOOB;
            wh[c] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o, 0, 0));
            wl[c] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o == gemm::kOOB ? o : o + 1024u, 0, 0));
        }
    };
    auto contract = [&]() {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();   // the split tile (and inv_s) is complete
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const unsigned char* pa = As + li * TS + c * 128 + lg * 16;
            const h8 ah = *reinterpret_cast<const h8*>(pa), al = *reinterpret_cast<const h8*>(pa + 64);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[c], acc, 0, 0, 0);
        }
        return acc;
    };

    if constexpr (PH == 1) {
        float4 zx[WNT];
        load_rows(a.Z, a.ldz, a.K, zx);
        const float isw = ldf(gemm::make_rsrc(a.W0f.inv_scale, (unsigned)(a.N * 4)), col < a.N, col);
        const float bv = ldf(buf(a.b0, a.N * 4), col < a.N, col);
        request(a.W0f, a.N);
        __builtin_amdgcn_sched_barrier(0);   // (requests above, consumers below)
        stamp();  // 1 requests out
        split_rows(zx);
        const f32x4 acc = contract();
        stamp();  // 2 contraction issued
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * lg + r;
            if (col < a.N && i < nrows) a.A1[(row0 + i) * a.N + col] = apply_act_small(acc[r] * (isw * inv_s[i]) + bv, a.act, a.slope);
        }
        stamp();  // 3 end
        return;
    } else {
        // ---- requests: the rows of A1, the criterion's inputs (element (row ei, task ej) of this row block on thread tid < 16 t; the
        // targets of the WHOLE batch — B t <= 4 096 values, 16 per thread — for the count of finite ones), W1, 1 / s_n of W0's rows for
        // the thread's columns of dl/dA1 (n = tid, tid + 256), the fragments of the wave's 16 columns of W0 ----
        float4 ax[WNT];
        load_rows(a.A1, a.N, a.N, ax);
        const int ei = tid / t, ej = tid - ei * t;
        const bool elem = tid < 16 * t && ei < nrows;
        const int64_t egi = (row0 + ei) * t + ej;
        const gemm::rsrc_t rT = buf(a.T, a.B * t * 4);
        const float ey = ldf(rT, elem, egi);
        const float ewv = ldf(buf(a.w, a.B * 4), elem, row0 + ei), etv = ldf(buf(a.tw, t * 4), elem, ej);
        const unsigned char elt = __builtin_amdgcn_raw_buffer_load_b8(buf(a.lt, a.B * t), elem ? (unsigned)egi : gemm::kOOB, 0, 0);
        const unsigned char egt = __builtin_amdgcn_raw_buffer_load_b8(buf(a.gt, a.B * t), elem ? (unsigned)egi : gemm::kOOB, 0, 0);
        constexpr int NCNT = (int)(kRowsMaxB * kOutMaxTasks / 256);
        float cv[NCNT];
#pragma unroll
        for (int u = 0; u < NCNT; ++u) cv[u] = ldf(rT, tid + 256 * u < a.B * t, tid + 256 * u);
        constexpr int NW1 = kOutMaxTasks * BN / 256;
        float w1v[NW1];
        const gemm::rsrc_t rW1 = gemm::make_rsrc(a.W1, (unsigned)(t * a.N * 4));
#pragma unroll
        for (int u = 0; u < NW1; ++u) {
            const int idx = tid + 256 * u, j = idx / BN, n = idx - j * BN;
            w1v[u] = ldf(rW1, j < t && n < a.N, j * a.N + n);
        }
        const gemm::rsrc_t rIS = gemm::make_rsrc(a.W0f.inv_scale, (unsigned)(a.N * 4));
        constexpr int NCOL = (BN + 255) / 256;
        float isn[NCOL];
#pragma unroll
        for (int u = 0; u < NCOL; ++u) isn[u] = ldf(rIS, tid + 256 * u < a.N, tid + 256 * u);
        float b1v[kOutMaxTasks];
#pragma unroll
        for (int j = 0; j < kOutMaxTasks; ++j) b1v[j] = ldf(buf(a.b1, t * 4), j < t, j);
        request(a.W0b, a.K);
        __builtin_amdgcn_sched_barrier(0);   // (requests above, consumers below)
        stamp();  // 1 requests out
#pragma unroll
        for (int u = 0; u < NW1; ++u) { const int idx = tid + 256 * u; W1s[idx / BN][idx % BN] = w1v[u]; }   // (read by every row of the tile, twice)
        // dl/dP as 16 x 4: columns beyond t must be ZERO (they meet W1's zero rows below) — all of the tile first, the t live columns
        // after the criterion.  (Round 5 zeroed them with the threads 16 t .. 63 only: the columns >= t of the rows below 4 t kept whatever
        // the LDS held — 0 x a finite leftover on a warm box, NaN on a cold one; found in round 6 with DMPNN_DEBUG_LDS_POISON)
        if (tid < 16 * kOutMaxTasks) gPs[tid / kOutMaxTasks][tid % kOutMaxTasks] = 0.f;
#pragma unroll
        for (int j = 0; j < WNT; ++j) *reinterpret_cast<float4*>(A1s + (tid >> 4) * LDA + 4 * (tid & 15) + 64 * j) = ax[j];   // (zero beyond N and beyond the batch)
        __syncthreads();
        stamp();  // 2 A1 in the tile
        // ---- P = A1 W1^T + b1: thread (row i, part p) sums the columns n = p mod 16 ----
        {
            const int i = tid >> 4, p = tid & 15;
            float sp[kOutMaxTasks] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < BN / 16; ++q) {   // (columns beyond N: zeros in both tiles)
                const int n = p + 16 * q;
                const float x = A1s[i * LDA + n];
#pragma unroll
                for (int j = 0; j < kOutMaxTasks; ++j) sp[j] += x * W1s[j][n];
            }
#pragma unroll
            for (int j = 0; j < kOutMaxTasks; ++j) {
                for (int off = 1; off < 16; off <<= 1) sp[j] += __shfl_xor(sp[j], off);
                if (p == 0 && j < t) {
                    const float v = sp[j] + b1v[j];
                    Ps[i][j] = v;
                    if (cs == 0 && i < nrows) a.preds[(row0 + i) * t + j] = v;
                }
            }
        }
        // ---- criterion (k_loss's arithmetic per element): this row block's rows; the count over the whole batch ----
        float sm = 0.f, sl = 0.f;
#pragma unroll
        for (int u = 0; u < NCNT; ++u) sm += (tid + 256 * u < a.B * t && isfinite(cv[u])) ? 1.f : 0.f;
        __syncthreads();   // Ps
        float ef = 0.f, ep = 0.f;
        const bool efin = elem && isfinite(ey);
        if (elem) {
            ep = Ps[ei][ej];
            if ((elt && ep < ey) || (egt && ep > ey)) ep = ey;
            ef = (a.w ? ewv : 1.f) * (a.tw ? etv : 1.f);
            if (efin) sl = loss_value(a.kind, ep, ey) * ef;
        }
        for (int off = 32; off > 0; off >>= 1) { sl += __shfl_xor(sl, off); sm += __shfl_xor(sm, off); }
        if (lane == 0) { red2[0][wave] = sl; red2[1][wave] = sm; }
        __syncthreads();
        if (tid == 0) {
            tot[0] = (red2[0][0] + red2[0][1]) + (red2[0][2] + red2[0][3]);
            tot[1] = (red2[1][0] + red2[1][1]) + (red2[1][2] + red2[1][3]);
            if (cs == 0) {
                float* pp = a.part + (int64_t)blockIdx.x * a.part_stride + t * a.N + t;
                pp[0] = tot[0]; pp[1] = tot[1];
            }
        }
        __syncthreads();
        if (tid < 16 * t) gPs[ei][ej] = efin ? loss_deriv(a.kind, ep, ey) * ef * (1.f / tot[1]) : 0.f;   // (columns beyond t: zeroed above)
        __syncthreads();
        stamp();  // 3 criterion
        // ---- the output layer's backward on this row block: thread = column n of A1 (all of them: the operand of the contraction below) ----
        //   dl/dA1[i][n] = (sum_j gP[i][j] W1[j][n]) tau'(A1[i][n]),  partial gW1[j][n] = sum_i gP[i][j] A1[i][n],  partial gb1[j] = sum_i gP[i][j]
        // what leaves for memory is the slice's: columns [64 cs, 64 cs + 64)
        // (the 16 rows of a column in registers, dl/dP as 16-byte rows: straight-line code — a loop with the activation's switch inside
        //  waited for every LDS round trip, 6 us)
        const bool simple_act = a.act != DMPNN_ACT_TANH && a.act != DMPNN_ACT_ELU;
        const float neg = a.act == DMPNN_ACT_NONE ? 1.f : (a.act == DMPNN_ACT_RELU ? 0.f : a.slope);   // tau' for y <= 0 (ReLU class)
        float4 gp4[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) gp4[i] = *reinterpret_cast<const float4*>(&gPs[i][0]);
#pragma unroll
        for (int u = 0; u < NCOL; ++u) {
            const int n = tid + 256 * u;
            if (n >= BN) break;
            const bool okn = n < a.N, mine = okn && (n >> 6) == cs;
            const float isn_c = isn[u];
            const float4 w1 = make_float4(W1s[0][n], W1s[1][n], W1s[2][n], W1s[3][n]);   // (rows beyond t are zero)
            float xv[16], gv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) xv[i] = A1s[i * LDA + n];
            float4 gw = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float4 gp = gp4[i];   // (columns beyond t are zero)
                const float g = ((gp.x * w1.x + gp.y * w1.y) + gp.z * w1.z) + gp.w * w1.w;
                gw.x += gp.x * xv[i]; gw.y += gp.y * xv[i]; gw.z += gp.z * xv[i]; gw.w += gp.w * xv[i];
                const float dv = simple_act ? (xv[i] > 0.f ? 1.f : neg) : act_grad_from_out(xv[i], a.act, a.slope);
                gv[i] = okn ? g * dv : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                A1s[i * LDA + n] = gv[i] * isn_c;   // (the operand of dl/dA1 . W0 carries W0's row scales on its reduction index: see HeadSplit)
                if (mine && i < nrows) a.gA1[(row0 + i) * a.N + n] = gv[i];
            }
            if (mine) {
                const float gwv[4] = {gw.x, gw.y, gw.z, gw.w};
                for (int j = 0; j < t; ++j) a.part[(int64_t)blockIdx.x * a.part_stride + (int64_t)j * a.N + n] = gwv[j];
            }
        }
        if (cs == 0 && tid < t) {
            float sb = 0.f;
            for (int i = 0; i < 16; ++i) sb += gPs[i][tid];
            a.part[(int64_t)blockIdx.x * a.part_stride + t * a.N + tid] = sb;
        }
        __syncthreads();
        stamp();  // 4 output layer's backward
        // ---- dl/dZ[:, slice] = dl/dA1 . W0 ----
        {
            float4 gx[WNT];
#pragma unroll
            for (int j = 0; j < WNT; ++j) gx[j] = *reinterpret_cast<const float4*>(A1s + (tid >> 4) * LDA + 4 * (tid & 15) + 64 * j);
            split_rows(gx);
        }
        const f32x4 acc = contract();
        stamp();  // 5 contraction issued
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * lg + r;
            if (col < a.K && i < nrows) a.gZ[(row0 + i) * a.K + col] = acc[r] * inv_s[i];
        }
        stamp();  // 6 end
    }
}

// out[c][r] = in[r][c] for a weight matrix (<= a few hundred KB)
__global__ void k_head_transpose(const float* __restrict__ in, int64_t ldi, float* __restrict__ out, int64_t ldo, int rows, int cols) {
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
// g[r][c] *= tau'(Y[r][c])   (Y = the activated output the next layer consumed)
__global__ void k_head_act_bwd(float* __restrict__ g, int64_t ldg, const float* __restrict__ Y, int64_t ldy, int64_t rows, int cols,
                               int act, float slope) {
    const int64_t n = rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols; const int c = (int)(i - r * cols);
        g[r * ldg + c] *= act_grad_from_out(Y[r * ldy + c], act, slope);
    }
}

struct HeadLayout {
    size_t bounds, Hm, Z, mean, invstd, act[DMPNN_MAX_FFN_LAYERS], gP, gA, gB, Wt, wgrad, gHm, total;
    size_t wgrad_bytes;
    int64_t maxd;
    // the four-launch form (rows_shape): the hidden layer's split weight in both orientations, their row scales, the row kernel's partials
    size_t W0f, W0b, isf, isb, part;
    int part_stride, n_part;
};
// the shapes the four-launch form takes (training or not is the caller's business): one hidden layer, a handful of outputs
bool rows_shape(const dmpnn_head_args& h) {
    return h.n_layers == 2 && h.dims[2] <= kOutMaxTasks && h.dims[2] >= 1 && h.loss <= DMPNN_LOSS_BCE && h.n_mols <= kRowsMaxB &&
           h.dims[0] <= kRowsMaxWidth && h.dims[1] <= kRowsMaxWidth && h.dims[0] % 4 == 0;
}
HeadLayout head_layout(const dmpnn_head_args& h) {
    HeadLayout L;
    memset(&L, 0, sizeof(L));
    const int64_t B = h.n_mols > 0 ? h.n_mols : 0, d = h.d_h;
    size_t o = 0;
    L.bounds = o; o += al256(dmpnn_molagg_ws_bytes(B));
    L.Hm = o; o += al256((size_t)B * d * 4);
    L.Z = o; o += al256(h.bn_weight ? (size_t)B * d * 4 : 0);
    L.mean = o; o += al256((size_t)d * 4);
    L.invstd = o; o += al256((size_t)d * 4);
    int64_t maxd = d;
    for (int l = 0; l < h.n_layers; ++l) {
        if (h.dims[l + 1] > maxd) maxd = h.dims[l + 1];
        L.act[l] = o;
        if (l + 1 < h.n_layers) o += al256((size_t)B * h.dims[l + 1] * 4);   // (the last layer writes `preds`)
    }
    L.maxd = maxd;
    const int64_t t = h.n_layers > 0 ? h.dims[h.n_layers] : d;
    L.gP = o; o += al256((size_t)B * t * 4);
    L.gA = o; o += al256((size_t)B * maxd * 4);
    L.gB = o; o += al256((size_t)B * maxd * 4);
    L.Wt = o; o += al256((size_t)maxd * maxd * 4);
    size_t wg = 0;
    for (int l = 0; l < h.n_layers; ++l) {
        const size_t b = dmpnn_linear_wgrad_ws_bytes(B, h.dims[l + 1], h.dims[l], 1);
        if (b > wg) wg = b;
    }
    if (h.n_layers > 0) {  // (the first layer's weight gradient may ride in the block's backward launches: its split operands + slabs)
        const size_t b = extra_wgrad_ws_floats(B, (int)h.dims[1], (int)h.dims[0] + 1) * sizeof(float);
        if (b > wg) wg = b;
    }
    L.wgrad = o; L.wgrad_bytes = al256(wg); o += L.wgrad_bytes;
    L.gHm = o; o += al256((size_t)B * d * 4);
    if (h.n_layers == 2 && rows_shape(h)) {
        const int64_t N = h.dims[1], K = h.dims[0], t = h.dims[2];
        L.W0f = o; o += al256((size_t)((N + 15) / 16) * ((K + 31) / 32) * 2048);
        L.W0b = o; o += al256((size_t)((K + 15) / 16) * ((N + 31) / 32) * 2048);
        L.isf = o; o += al256((size_t)N * 4);
        L.isb = o; o += al256((size_t)K * 4);
        L.n_part = (int)((B + 15) / 16);
        L.part_stride = (int)(t * N + t + 2);
        L.part = o; o += al256((size_t)L.n_part * L.part_stride * 4);
    }
    L.total = o;
    return L;
}

}  // namespace
}  // namespace dmpnn

using namespace dmpnn;

extern "C" {

size_t dmpnn_head_ws_bytes(const dmpnn_head_args* h) {
    if (!h || h->n_layers < 0 || h->n_layers > DMPNN_MAX_FFN_LAYERS || h->d_h <= 0) return 0;
    return head_layout(*h).total;
}

}  // extern "C"

namespace {
int head_run(const dmpnn_head_args* hp, const float* Hv, int64_t ldhv, void* stream, bool bounds_done, ExtraWgrad* defer, bool agg_rode = false);
}  // namespace

extern "C" {

int dmpnn_head(const dmpnn_head_args* hp, const float* Hv, int64_t ldhv, void* stream) { return head_run(hp, Hv, ldhv, stream, false, nullptr); }

}  // extern "C"

namespace {
// quads per workgroup / molecules per thread of the column kernels: (4, 1) up to 256 molecules, (2, 1) up to 512, (2, 2) up to 1 024
// (DMPNN_HEAD_QPW=4 | 2: the A/B switch)
int col_quads(int64_t B) {
    const char* e = getenv("DMPNN_HEAD_QPW");
    if (e && !strcmp(e, "4")) return 4;
    if (e && !strcmp(e, "2")) return 2;
    return B <= 256 ? 4 : 2;
}
#define DMPNN_COL_LAUNCH(KERNEL, QPW_V, B_V, BN_V, GRID, STREAM, ARGS)                                                   \
    do {                                                                                                                \
        const int rr_ = (int)(((B_V) * (QPW_V) + 1023) / 1024);                                                          \
        if ((QPW_V) == 4 && rr_ <= 1) { if (BN_V) hipLaunchKernelGGL((KERNEL<4, 1, true>), GRID, dim3(1024), 0, STREAM, ARGS); else hipLaunchKernelGGL((KERNEL<4, 1, false>), GRID, dim3(1024), 0, STREAM, ARGS); } \
        else if ((QPW_V) == 4 && rr_ <= 2) { if (BN_V) hipLaunchKernelGGL((KERNEL<4, 2, true>), GRID, dim3(1024), 0, STREAM, ARGS); else hipLaunchKernelGGL((KERNEL<4, 2, false>), GRID, dim3(1024), 0, STREAM, ARGS); } \
        else if ((QPW_V) == 4) { if (BN_V) hipLaunchKernelGGL((KERNEL<4, 4, true>), GRID, dim3(1024), 0, STREAM, ARGS); else hipLaunchKernelGGL((KERNEL<4, 4, false>), GRID, dim3(1024), 0, STREAM, ARGS); } \
        else if (rr_ <= 1) { if (BN_V) hipLaunchKernelGGL((KERNEL<2, 1, true>), GRID, dim3(1024), 0, STREAM, ARGS); else hipLaunchKernelGGL((KERNEL<2, 1, false>), GRID, dim3(1024), 0, STREAM, ARGS); } \
        else { if (BN_V) hipLaunchKernelGGL((KERNEL<2, 2, true>), GRID, dim3(1024), 0, STREAM, ARGS); else hipLaunchKernelGGL((KERNEL<2, 2, false>), GRID, dim3(1024), 0, STREAM, ARGS); } \
    } while (0)
int launch_bn_agg_bwd(const BnAggBwdArgs& q, bool bn, hipStream_t s) {
    const int qpw = col_quads(q.b.B);
    const dim3 grid((unsigned)((q.b.d + 4 * qpw - 1) / (4 * qpw)) + (q.part ? 1u : 0u));   // (+ 1: the row kernels' partials)
    DMPNN_COL_LAUNCH(k_bn_agg_bwd, qpw, q.b.B, bn, grid, s, q);
    DMPNN_CHECK_LAUNCH("k_bn_agg_bwd");
    return DMPNN_OK;
}
// defer (a whole training step only): the weight gradient of the predictor's FIRST layer is not launched here but described in
// *defer — it rides in the launches of the block's backward pass (ExtraWgrad); its inputs (the layer's output gradient, the
// layer's input) stay untouched in the workspace until then
// agg_rode (a whole training step only): the forward tile kernel wrote the aggregate of the molecules it carried into the workspace's H
// and marked them in the bounds table's done[] (AggRide)
int head_run(const dmpnn_head_args* hp, const float* Hv, int64_t ldhv, void* stream, bool bounds_done, ExtraWgrad* defer, bool agg_rode) {
    DMPNN_CHECK_ARG(hp != nullptr, "head: null args");
    const dmpnn_head_args& h = *hp;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t B = h.n_mols, d = h.d_h, nV = h.n_atoms;
    const int Ln = h.n_layers;
    DMPNN_CHECK_ARG(B >= 0 && nV >= 0 && d > 0 && ldhv >= d, "head: bad sizes");
    DMPNN_CHECK_ARG(Ln >= 1 && Ln <= DMPNN_MAX_FFN_LAYERS && h.dims[0] == d, "head: 1..%d predictor layers, dims[0] == d_h", DMPNN_MAX_FFN_LAYERS);
    for (int l = 0; l < Ln; ++l) DMPNN_CHECK_ARG(h.W[l] && h.dims[l + 1] > 0, "head: layer %d has no weight / width", l);
    DMPNN_CHECK_ARG(h.act >= DMPNN_ACT_NONE && h.act <= DMPNN_ACT_ELU && h.act != DMPNN_ACT_PRELU, "head: activation %d is not built in", h.act);
    DMPNN_CHECK_ARG(h.loss >= DMPNN_LOSS_MSE && h.loss <= DMPNN_LOSS_QUANTILE, "head: unknown criterion %d", h.loss);
    DMPNN_CHECK_ARG(h.loss <= DMPNN_LOSS_MAE || (!h.lt_mask && !h.gt_mask), "head: only the MSE / MAE criteria have bounds (lt_mask / gt_mask)");
    DMPNN_CHECK_ARG((h.loss != DMPNN_LOSS_MVE && h.loss != DMPNN_LOSS_QUANTILE) || h.dims[Ln] % 2 == 0, "head: the MVE / quantile criteria need an output layer 2 n_tasks wide");
    DMPNN_CHECK_ARG(h.loss != DMPNN_LOSS_QUANTILE || (h.quantile_alpha > 0.f && h.quantile_alpha < 1.f), "head: the quantile criterion needs 0 < quantile_alpha < 1");
    DMPNN_CHECK_ARG(h.loss != DMPNN_LOSS_EVIDENTIAL || h.dims[Ln] % 4 == 0, "head: the evidential criterion needs an output layer 4 n_tasks wide");
    DMPNN_CHECK_ARG(h.loss != DMPNN_LOSS_CE || (h.n_classes >= 2 && h.dims[Ln] % h.n_classes == 0), "head: cross entropy needs n_classes >= 2 dividing the output width");
    DMPNN_CHECK_ARG(h.preds && (nV == 0 || (Hv && h.batch)), "head: null H_v / batch / preds");
    DMPNN_CHECK_ARG(!h.bn_weight || (h.bn_running_mean && h.bn_running_var), "head: batch norm without running statistics");
    // (torch.nn.BatchNorm1d in training mode — hence the reference — raises "Expected more than 1 value per channel": a batch of one
    //  molecule has no variance, and the output would silently be beta)
    DMPNN_CHECK_ARG(!(h.bn_weight && h.bn_training) || B != 1, "head: batch norm in training mode needs more than 1 molecule per batch");
    const bool want_grad = h.gHv != nullptr;
    DMPNN_CHECK_ARG(!want_grad || (h.targets && h.loss_out && h.ldg >= d), "head: gradients need targets, loss_out and ldg >= d_h");
    const HeadLayout L = head_layout(h);
    if (!h.ws || h.ws_bytes < L.total) {
        set_error("head: workspace missing or too small (%zu < %zu bytes)", h.ws_bytes, L.total);
        return DMPNN_ENOSPC;
    }
    DMPNN_CHECK_ARG(aligned16(h.ws), "head: workspace must be 16-byte aligned");
    if (B == 0) return DMPNN_OK;
    unsigned char* ws = static_cast<unsigned char*>(h.ws);
    float* Hm = reinterpret_cast<float*>(ws + L.Hm);
    const int t_out = (int)h.dims[Ln];                                  // width of the output layer
    const int nc = h.loss == DMPNN_LOSS_CE ? h.n_classes : ((h.loss == DMPNN_LOSS_MVE || h.loss == DMPNN_LOSS_QUANTILE) ? 2 : (h.loss == DMPNN_LOSS_EVIDENTIAL ? 4 : 1));   // outputs per task
    const int t = t_out / nc;                                           // tasks (= columns of `targets`)

    // ---- forward ----
    if (!bounds_done) DMPNN_TRY(dmpnn_molagg_bounds(h.batch, nV, B, ws + L.bounds, dmpnn_molagg_ws_bytes(B), stream));
    // round 5: the four-launch form (see k_agg_bn_fwd) — aggregation + batch norm as ONE column kernel for <= kRowsMaxB molecules
    // (forward, inference included), the whole predictor + criterion + their backward as ONE row kernel when the shape fits
    const char* head_env = getenv("DMPNN_HEAD");
    const bool chain = head_env && !strcmp(head_env, "chain");
    // (the column kernels move 16-byte quads: widths and strides in multiples of 4 floats, 16-byte aligned rows)
    const bool cols_fused = !chain && B <= kRowsMaxB && nV > 0 && d % 4 == 0 && ldhv % 4 == 0 && aligned16(Hv) &&
                            (!want_grad || (h.ldg % 4 == 0 && aligned16(h.gHv))) && (int64_t)nV * ldhv < (1ll << 29) &&
                            (!h.bn_weight || (aligned16(h.bn_weight) && aligned16(h.bn_bias) && aligned16(h.bn_running_mean) && aligned16(h.bn_running_var) &&
                                              (!h.g_bn_weight || aligned16(h.g_bn_weight)) && (!h.g_bn_bias || aligned16(h.g_bn_bias))));
    const bool rows = cols_fused && want_grad && h.targets && rows_shape(h) && aligned16(h.W[0]);
    // (DMPNN_HEAD=rows: tests — a training call that does NOT take the four-launch form is an error instead of a silent chain)
    DMPNN_CHECK_ARG(!(head_env && !strcmp(head_env, "rows")) || rows || !want_grad, "head: DMPNN_HEAD=rows, but this shape takes the chain");
    const float* Z = Hm;
    float* mean = reinterpret_cast<float*>(ws + L.mean);
    float* invstd = reinterpret_cast<float*>(ws + L.invstd);
    SplitW W0f{ws + L.W0f, reinterpret_cast<float*>(ws + L.isf), (int)((h.dims[0] + 31) / 32)};
    SplitW W0b{ws + L.W0b, reinterpret_cast<float*>(ws + L.isb), (int)((h.dims[1] + 31) / 32)};
    if (cols_fused) {
        AggBnArgs q;
        memset(&q, 0, sizeof(q));
        q.b = BnArgs{Hm, d, reinterpret_cast<float*>(ws + L.Z), d, h.bn_weight, h.bn_bias, h.bn_running_mean, h.bn_running_var, mean, invstd,
                     B, (int)d, h.bn_eps, h.bn_momentum, h.bn_training, h.bn_num_batches_tracked};
        // the aggregation inside this launch up to 512 molecules (one launch less: 159 against 166 us per step at 64 molecules, 221
        // against 220 at 512 — the column workgroups pull H_v at ~55 GB/s each, 5.5 MB over 38 of them); beyond that
        // dmpnn_molagg_fwd — every CU — runs in front (DMPNN_HEAD_AGG=fused | split: the A/B switch)
        const char* agg_env = getenv("DMPNN_HEAD_AGG");
        const bool fuse_agg = agg_rode || (agg_env ? !strcmp(agg_env, "fused") : B <= kFuseAggMols);   // (rode: only what the tile kernel left is summed here)
        q.use_done = agg_rode ? 1 : 0;
        if (!fuse_agg) DMPNN_TRY(dmpnn_molagg_fwd(Hv, ldhv, nV, d, B, ws + L.bounds, h.agg_mode, h.agg_norm, Hm, d, stream));
        q.Hv = fuse_agg ? Hv : nullptr; q.ldhv = ldhv; q.nV = nV; q.bounds = reinterpret_cast<const int*>(ws + L.bounds); q.agg_mode = h.agg_mode; q.agg_norm = h.agg_norm;
        const int qpw = col_quads(B);
        q.n_col_blocks = (int)((d + 4 * qpw - 1) / (4 * qpw));
        q.dbg = g_debug_stamps ? g_debug_stamps + 80 : nullptr;
        int split_blocks = 0;
        if (rows) {
            const int N = (int)h.dims[1], K = (int)h.dims[0];
            q.split = HeadSplit{h.W[0], N, K, ws + L.W0f, ws + L.W0b, reinterpret_cast<float*>(ws + L.isf), W0f.nc, W0b.nc};
            split_blocks = W0b.nc;
        }
        const dim3 grid((unsigned)(q.n_col_blocks + split_blocks));
        const bool bn = h.bn_weight != nullptr;
        if (fuse_agg || bn || split_blocks) DMPNN_COL_LAUNCH(k_agg_bn_fwd, qpw, B, bn, grid, s, q);
        DMPNN_CHECK_LAUNCH("k_agg_bn_fwd");
        if (bn) Z = reinterpret_cast<float*>(ws + L.Z);
    } else
        DMPNN_TRY(dmpnn_molagg_fwd(Hv, ldhv, nV, d, B, ws + L.bounds, h.agg_mode, h.agg_norm, Hm, d, stream));
    if (h.bn_weight && !cols_fused) {
        BnArgs b{Hm, d, reinterpret_cast<float*>(ws + L.Z), d, h.bn_weight, h.bn_bias, h.bn_running_mean, h.bn_running_var, mean, invstd,
                 B, (int)d, h.bn_eps, h.bn_momentum, h.bn_training, h.bn_num_batches_tracked};
        if (B <= kBnRegRows * kBnLanes) hipLaunchKernelGGL(k_bn_fwd<true>, dim3((unsigned)((d + kBnCols - 1) / kBnCols)), dim3(1024), 0, s, b);
        else hipLaunchKernelGGL(k_bn_fwd<false>, dim3((unsigned)((d + kBnCols - 1) / kBnCols)), dim3(1024), 0, s, b);
        DMPNN_CHECK_LAUNCH("k_bn_fwd");
        Z = reinterpret_cast<float*>(ws + L.Z);
    }
    if (rows) {
        const int N = (int)h.dims[1], K = (int)h.dims[0];
        float* gA1 = reinterpret_cast<float*>(ws + L.gA);
        float* gZr = reinterpret_cast<float*>(ws + L.gB);
        RowsArgs r{Z, d, W0f, W0b, h.b[0], h.W[1], h.b[1], reinterpret_cast<float*>(ws + L.act[0]), h.preds, h.targets, h.weights, h.task_weights,
                   h.lt_mask, h.gt_mask, h.loss, gA1, gZr, reinterpret_cast<float*>(ws + L.part), L.part_stride, B, N, K, t, h.act, h.act_slope,
                   g_debug_stamps ? g_debug_stamps + 64 : nullptr};
        const int widest = N > K ? N : K;
        const dim3 g1((unsigned)L.n_part, (unsigned)((N + 63) / 64)), g2((unsigned)L.n_part, (unsigned)((widest + 63) / 64));
        if (widest <= 128) {
            hipLaunchKernelGGL((k_head_rows<2, 1>), g1, dim3(256), 0, s, r);
            if (r.dbg) r.dbg += 8;
            hipLaunchKernelGGL((k_head_rows<2, 2>), g2, dim3(256), 0, s, r);
        } else {
            hipLaunchKernelGGL((k_head_rows<5, 1>), g1, dim3(256), 0, s, r);
            if (r.dbg) r.dbg += 8;
            hipLaunchKernelGGL((k_head_rows<5, 2>), g2, dim3(256), 0, s, r);
        }
        DMPNN_CHECK_LAUNCH("k_head_rows");
        if (h.gW[0] || h.gb[0]) {   // the hidden layer's weight gradient: in the block's backward launches, or its own product
            if (defer && h.gW[0] && N % 2 == 0 && K % 2 == 0) {
                *defer = ExtraWgrad{gA1, N, Z, K, B, N, K, h.b[0] ? 1 : 0, h.gW[0], K, h.b[0] ? h.gb[0] : nullptr, reinterpret_cast<float*>(ws + L.wgrad)};
            } else {
                dmpnn_gemm_args g;
                memset(&g, 0, sizeof(g));
                g.M = B; g.N = N; g.K1 = K; g.A1 = Z; g.lda1 = K;
                float* gw = h.gW[0] ? h.gW[0] : reinterpret_cast<float*>(ws + L.Wt);
                DMPNN_TRY(dmpnn_linear_wgrad(&g, gA1, N, gw, K, h.b[0] ? h.gb[0] : nullptr, ws + L.wgrad, L.wgrad_bytes, stream));
            }
        }
        BnAggBwdArgs q;
        memset(&q, 0, sizeof(q));
        q.b = BnBwdArgs{gZr, d, Hm, d, nullptr, d, h.bn_weight, mean, invstd, h.bn_running_mean, h.bn_running_var, h.g_bn_weight, h.g_bn_bias,
                        B, (int)d, h.bn_eps, h.bn_training};
        q.gHv = h.gHv; q.ldg = h.ldg; q.bounds = reinterpret_cast<const int*>(ws + L.bounds); q.nV = nV; q.agg_mode = h.agg_mode; q.agg_norm = h.agg_norm;
        q.part = reinterpret_cast<const float*>(ws + L.part); q.n_part = L.n_part; q.part_stride = L.part_stride; q.tN = t * N; q.t = t;
        q.gW1 = h.gW[1]; q.gb1 = h.b[1] ? h.gb[1] : nullptr; q.loss_out = h.loss_out;
        q.dbg = g_debug_stamps ? g_debug_stamps + 96 : nullptr;
        return launch_bn_agg_bwd(q, h.bn_weight != nullptr, s);
    }
    const float* A[DMPNN_MAX_FFN_LAYERS + 1];
    A[0] = Z;
    const bool small_out = h.dims[Ln] <= kOutMaxTasks && Ln >= 1;   // the output layer as dot products (k_out_fwd / k_out_bwd)
    // training on a short batch: the criterion and the output layer's backward are ONE launch further down (k_out_all)
    const bool out_all = small_out && want_grad && h.targets && B <= kOutAllMaxRows && nc == 1;
    for (int l = 0; l < Ln; ++l) {
        if (small_out && l == Ln - 1) {
            OutFwdArgs q{A[l], h.dims[l], h.W[l], h.b[l], h.preds, B, (int)h.dims[l], (int)h.dims[Ln]};
            hipLaunchKernelGGL(k_out_fwd, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, q);
            DMPNN_CHECK_LAUNCH("k_out_fwd");
            A[l + 1] = h.preds;
            break;
        }
        dmpnn_gemm_args g;
        memset(&g, 0, sizeof(g));
        g.M = B; g.N = h.dims[l + 1]; g.K1 = h.dims[l];
        g.A1 = A[l]; g.lda1 = h.dims[l];
        g.W = h.W[l]; g.ldw = h.dims[l]; g.bias = h.b[l];
        float* out = l + 1 < Ln ? reinterpret_cast<float*>(ws + L.act[l]) : h.preds;
        g.C = out; g.ldc = h.dims[l + 1];
        g.act = l + 1 < Ln ? h.act : DMPNN_ACT_NONE;   // sigma of the NEXT block fused here (ffn.py:49-58)
        g.act_slope = h.act_slope;
        DMPNN_TRY(dmpnn_linear_fwd(&g, stream));
        A[l + 1] = out;
    }
    if (!h.targets) return DMPNN_OK;
    float* gP = reinterpret_cast<float*>(ws + L.gP);
    if (!out_all) {
        LossArgs q{h.preds, t_out, h.targets, t, h.weights, h.task_weights, h.lt_mask, h.gt_mask, want_grad ? gP : nullptr, t_out, h.loss_out, B, t, h.loss, nc,
                   h.evid_v_kl, h.evid_eps, h.quantile_alpha};
        DMPNN_CHECK_ARG(h.loss_out != nullptr, "head: targets without loss_out");
        hipLaunchKernelGGL(k_loss, dim3(1), dim3(1024), 0, s, q);
        DMPNN_CHECK_LAUNCH("k_loss");
    }
    if (!want_grad) return DMPNN_OK;

    // ---- backward ----
    float* bufs[2] = {reinterpret_cast<float*>(ws + L.gA), reinterpret_cast<float*>(ws + L.gB)};
    float* Wt = reinterpret_cast<float*>(ws + L.Wt);
    const float* g_cur = gP;   // gradient w.r.t. the pre-activation of layer l's output
    int pp = 0;
    for (int l = Ln - 1; l >= 0; --l) {
        const int64_t N = h.dims[l + 1], K = h.dims[l];
        if (small_out && l == Ln - 1) {
            float* out = bufs[pp]; pp ^= 1;
            // (l == 0: no activation in front of the only layer — the derivative factor is 1)
            OutBwdArgs q{g_cur, A[l], K, h.W[l], out, K, h.gW[l], h.b[l] ? h.gb[l] : nullptr, B, (int)K, (int)N, l > 0 ? h.act : DMPNN_ACT_NONE, h.act_slope};
            if (out_all) {
                DMPNN_CHECK_ARG(h.loss_out != nullptr, "head: targets without loss_out");
                OutAllArgs qa{q, LossArgs{h.preds, t, h.targets, t, h.weights, h.task_weights, h.lt_mask, h.gt_mask, nullptr, t, h.loss_out, B, t, h.loss, 1, 0.f, 0.f, 0.f}};
                qa.o.gP = nullptr;
                hipLaunchKernelGGL(k_out_all, dim3((unsigned)((K + kBnCols - 1) / kBnCols)), dim3(1024), 0, s, qa);
                DMPNN_CHECK_LAUNCH("k_out_all");
                g_cur = out;
                continue;
            }
            hipLaunchKernelGGL(k_out_bwd, dim3((unsigned)((K + kBnCols - 1) / kBnCols)), dim3(1024), 0, s, q);
            DMPNN_CHECK_LAUNCH("k_out_bwd");
            g_cur = out;
            continue;
        }
        if (h.gW[l] || h.gb[l]) {
            dmpnn_gemm_args g;
            memset(&g, 0, sizeof(g));
            g.M = B; g.N = N; g.K1 = K; g.A1 = A[l]; g.lda1 = K;
            float* gw = h.gW[l] ? h.gW[l] : Wt;  // (the product writes both; an unwanted one lands in scratch)
            if (defer && l == 0 && h.gW[l] && N % 2 == 0 && K % 2 == 0) {
                *defer = ExtraWgrad{g_cur, N, A[l], K, B, (int)N, (int)K, h.b[l] ? 1 : 0, h.gW[l], K, h.b[l] ? h.gb[l] : nullptr,
                                    reinterpret_cast<float*>(ws + L.wgrad)};
            } else {
                DMPNN_TRY(dmpnn_linear_wgrad(&g, g_cur, N, gw, K, h.b[l] ? h.gb[l] : nullptr, ws + L.wgrad, L.wgrad_bytes, stream));
            }
        }
        // data gradient: gA[l] = g . W_l   (the contraction kernel on W_l^T)
        hipLaunchKernelGGL(k_head_transpose, dim3((unsigned)((K + 31) / 32), (unsigned)((N + 31) / 32)), dim3(32, 8), 0, s, h.W[l], K, Wt, N, (int)N, (int)K);
        DMPNN_CHECK_LAUNCH("k_head_transpose");
        dmpnn_gemm_args g;
        memset(&g, 0, sizeof(g));
        g.M = B; g.N = K; g.K1 = N; g.A1 = g_cur; g.lda1 = N; g.W = Wt; g.ldw = N;
        float* out = bufs[pp]; pp ^= 1;
        g.C = out; g.ldc = K; g.act = DMPNN_ACT_NONE;
        DMPNN_TRY(dmpnn_linear_fwd(&g, stream));
        if (l > 0 && h.act != DMPNN_ACT_NONE) {
            const int64_t n = B * K;
            int64_t blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(k_head_act_bwd, dim3((unsigned)blocks), dim3(256), 0, s, out, K, A[l], K, B, (int)K, h.act, h.act_slope);
            DMPNN_CHECK_LAUNCH("k_head_act_bwd");
        }
        g_cur = out;
    }
    const float* gZ = g_cur;   // [B, d]: gradient w.r.t. the fingerprint
    float* gHm = reinterpret_cast<float*>(ws + L.gHm);
    if (cols_fused) {   // batch norm backward + the broadcast to the atoms' rows: one column kernel
        BnAggBwdArgs q;
        memset(&q, 0, sizeof(q));
        q.b = BnBwdArgs{gZ, d, Hm, d, nullptr, d, h.bn_weight, mean, invstd, h.bn_running_mean, h.bn_running_var, h.g_bn_weight, h.g_bn_bias,
                        B, (int)d, h.bn_eps, h.bn_training};
        q.gHv = h.gHv; q.ldg = h.ldg; q.bounds = reinterpret_cast<const int*>(ws + L.bounds); q.nV = nV; q.agg_mode = h.agg_mode; q.agg_norm = h.agg_norm;
        return launch_bn_agg_bwd(q, h.bn_weight != nullptr, s);
    }
    if (h.bn_weight) {
        BnBwdArgs b{gZ, d, Hm, d, gHm, d, h.bn_weight, mean, invstd, h.bn_running_mean, h.bn_running_var, h.g_bn_weight, h.g_bn_bias,
                    B, (int)d, h.bn_eps, h.bn_training};
        if (B <= kBnRegRows * kBnLanes) hipLaunchKernelGGL(k_bn_bwd<true>, dim3((unsigned)((d + kBnCols - 1) / kBnCols)), dim3(1024), 0, s, b);
        else hipLaunchKernelGGL(k_bn_bwd<false>, dim3((unsigned)((d + kBnCols - 1) / kBnCols)), dim3(1024), 0, s, b);
        DMPNN_CHECK_LAUNCH("k_bn_bwd");
        gZ = gHm;
    }
    // (Running the hidden layers' weight-gradient products and the molecule bounds on a second stream of the library's own was
    //  built and measured: each fork / join pair costs ~6 us of cross-queue synchronisation on this runtime — the step got 12 us
    //  SLOWER, profiles/r03_side_stream_ab.txt.  One stream.)
    return dmpnn_molagg_bwd(gZ, d, h.batch, nV, d, B, ws + L.bounds, h.agg_mode, h.agg_norm, h.gHv, h.ldg, stream);
}
}  // namespace

extern "C" {

// K0 + forward (kept tensors) + the head above + backward + optimizer: one training step of models/model.py:148-161 with
// torch.optim.Adam (model.py:208-231), every kernel enqueued by this one call.
int dmpnn_train_step(const dmpnn_step_args* a, void* stream) {
    DMPNN_CHECK_ARG(a != nullptr, "train_step: null args");
    // (round 6: on a tile plan built here from the batch vector, the tile kernels' launches are bounded by the batch's molecule count — a tile
    //  holds at least one molecule — instead of the layout's bound: 512 instead of 745 workgroups at 512 molecules, forward and backward;
    //  a plan that turns out to hold more tiles than that comes back NaN from both kernels)
    dmpnn_bwd_args bw = a->bwd;
    if ((bw.f.flags & DMPNN_F_TILE_PLAN) && bw.f.n_tiles_launch == 0 && a->head.n_mols > 0 && a->head.batch == a->batch) bw.f.n_tiles_launch = a->head.n_mols;
    const dmpnn_fwd_args& f = bw.f;
    DMPNN_CHECK_ARG((f.flags & DMPNN_F_KEEP) != 0, "train_step: the forward must keep its tensors (DMPNN_F_KEEP)");
    DMPNN_CHECK_ARG(a->head.gHv == a->bwd.gout && a->head.ldg == a->bwd.ldgout, "train_step: head.gHv must be the backward's gout");
    DMPNN_CHECK_ARG(a->head.n_atoms == f.n_atoms && a->head.d_h == f.d_h + (f.W_d ? f.d_vd : 0), "train_step: head and block sizes differ");
    const int stages = a->stages ? a->stages : (DMPNN_STEP_FORWARD | DMPNN_STEP_BACKWARD | DMPNN_STEP_UPDATE);
    ExtraWgrad rider;
    memset(&rider, 0, sizeof(rider));
    if (stages & DMPNN_STEP_FORWARD) {
        bool bounds_done = false, split_done = false;
        if (!a->plan_ready) {
            DMPNN_CHECK_ARG(a->edge_index && a->rev_edge_index, "train_step: null index arrays");
            if (f.flags & DMPNN_F_TILE_PLAN) {  // (the tile table alone: the kept tensors stay in the caller's edge order, dmpnn.h)
                // ... and the planner already holds every molecule's atom range: it writes the aggregation's bounds table on the side
                const dmpnn_head_args& h = a->head;
                int* mb = nullptr;
                if (h.ws && h.n_mols > 0 && h.batch == a->batch && h.n_atoms == f.n_atoms) {
                    const HeadLayout HL = head_layout(h);
                    if (h.ws_bytes >= HL.total) mb = reinterpret_cast<int*>(static_cast<unsigned char*>(h.ws) + HL.bounds);
                }
                DMPNN_TRY(prepare_tiles_and_bounds(a->edge_index, a->rev_edge_index, a->batch, f.n_atoms, f.n_edges, const_cast<void*>(f.plan),
                                                   a->plan_bytes, mb, h.n_mols, stream, &bounds_done, &f, &split_done));
            } else
                DMPNN_TRY(dmpnn_prepare_with_batch(a->edge_index, a->rev_edge_index, a->batch, f.n_atoms, f.n_edges, const_cast<void*>(f.plan),
                                                   a->plan_bytes, stream));
        }
        // the aggregate of the block's output leaves with the tile kernel's tiles (AggRide) when K0 wrote the bounds table (and zeroed
        // its done[]) and the head is going to take its column kernels on this shape; DMPNN_HEAD_AGG=fused | split switches it off
        bool agg_rode = false;
        {
            const dmpnn_head_args& h = a->head;
            const char* he = getenv("DMPNN_HEAD");
            const char* ae = getenv("DMPNN_HEAD_AGG");
            if (bounds_done && h.ws && h.n_mols > 0 && h.n_mols <= kRowsMaxB && h.d_h % 4 == 0 && h.d_h == f.d_h && !f.W_d && !(he && !strcmp(he, "chain")) &&
                !(ae && strcmp(ae, "tile"))) {
                const HeadLayout HL = head_layout(h);
                unsigned char* hws = static_cast<unsigned char*>(h.ws);
                g_agg_ride = AggRide{reinterpret_cast<float*>(hws + HL.Hm), (int)h.d_h, a->batch, reinterpret_cast<int*>(hws + HL.bounds), (int)h.n_mols,
                                     h.agg_mode, h.agg_norm, false};
            }
        }
        int frc;
        if (split_done) {   // (the weight pre-split rode in K0's launch)
            dmpnn_fwd_args f2 = f;
            f2.flags |= DMPNN_F_WSPLIT_READY;
            frc = dmpnn_forward(&f2, stream);
        } else
            frc = dmpnn_forward(&f, stream);
        agg_rode = g_agg_ride.taken;
        g_agg_ride = AggRide{nullptr, 0, nullptr, nullptr, 0, 0, 0.f, false};
        if (frc != DMPNN_OK) return frc;
        // (a whole step in one call: the first predictor layer's weight gradient rides in the block's backward launches; a staged
        //  step — data parallel — has the head's gradients final after this stage, so nothing is deferred there)
        DMPNN_TRY(head_run(&a->head, f.out, f.ldout, stream, bounds_done, (stages & DMPNN_STEP_BACKWARD) ? &rider : nullptr, agg_rode));
    }
    if (stages & DMPNN_STEP_BACKWARD) {
        bool rode = false;
        DMPNN_TRY(backward_impl(&bw, stream, rider.Z ? &rider : nullptr, &rode));
        if (rider.Z && !rode) {  // (the backward pass did not take the f16 products: the product of its own, as dmpnn_head would have run it)
            dmpnn_gemm_args g;
            memset(&g, 0, sizeof(g));
            g.M = rider.M; g.N = rider.N; g.K1 = rider.K; g.A1 = rider.A; g.lda1 = rider.lda;
            const HeadLayout HL = head_layout(a->head);
            DMPNN_TRY(dmpnn_linear_wgrad(&g, rider.Z, rider.ldz, rider.gW, rider.ldgw, rider.gb, rider.ws, HL.wgrad_bytes, stream));
        }
    }
    if ((stages & DMPNN_STEP_UPDATE) && a->n_params > 0 && a->clip_val > 0.f)   // Trainer(gradient_clip_val): between backward and update
        DMPNN_TRY(dmpnn_clip_grad(const_cast<float*>(a->g), a->n_params, a->clip_val, a->clip_mode, a->grad_scale > 0.f ? a->grad_scale : 1.f,
                                  a->clip_ws, stream));
    if ((stages & DMPNN_STEP_UPDATE) && a->n_params > 0)
        DMPNN_TRY(dmpnn_adam_step(a->p, a->g, a->m, a->v, a->n_params, a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, a->bias_corr1,
                                  a->sqrt_bias_corr2, a->grad_scale, a->dev_scalars, stream));
    return DMPNN_OK;
}

}  // extern "C"
