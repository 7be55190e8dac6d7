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

namespace dmpnn {
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
};
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
};
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
int head_run(const dmpnn_head_args* hp, const float* Hv, int64_t ldhv, void* stream, bool bounds_done, ExtraWgrad* defer);
}  // namespace

extern "C" {

int dmpnn_head(const dmpnn_head_args* hp, const float* Hv, int64_t ldhv, void* stream) { return head_run(hp, Hv, ldhv, stream, false, nullptr); }

}  // extern "C"

namespace {
// defer (a whole training step only): the weight gradient of the predictor's FIRST layer is not launched here but described in
// *defer — it rides in the launches of the block's backward pass (ExtraWgrad); its inputs (the layer's output gradient, the
// layer's input) stay untouched in the workspace until then
int head_run(const dmpnn_head_args* hp, const float* Hv, int64_t ldhv, void* stream, bool bounds_done, ExtraWgrad* defer) {
    DMPNN_CHECK_ARG(hp != nullptr, "head: null args");
    const dmpnn_head_args& h = *hp;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t B = h.n_mols, d = h.d_h, nV = h.n_atoms;
    const int Ln = h.n_layers;
    DMPNN_CHECK_ARG(B >= 0 && nV >= 0 && d > 0 && ldhv >= d, "head: bad sizes");
    DMPNN_CHECK_ARG(Ln >= 1 && Ln <= DMPNN_MAX_FFN_LAYERS && h.dims[0] == d, "head: 1..%d predictor layers, dims[0] == d_h", DMPNN_MAX_FFN_LAYERS);
    for (int l = 0; l < Ln; ++l) DMPNN_CHECK_ARG(h.W[l] && h.dims[l + 1] > 0, "head: layer %d has no weight / width", l);
    DMPNN_CHECK_ARG(h.act >= DMPNN_ACT_NONE && h.act <= DMPNN_ACT_ELU && h.act != DMPNN_ACT_PRELU, "head: activation %d is not built in", h.act);
    DMPNN_CHECK_ARG(h.loss == DMPNN_LOSS_MSE || h.loss == DMPNN_LOSS_MAE || h.loss == DMPNN_LOSS_BCE || h.loss == DMPNN_LOSS_CE, "head: unknown criterion %d", h.loss);
    DMPNN_CHECK_ARG((h.loss != DMPNN_LOSS_BCE && h.loss != DMPNN_LOSS_CE) || (!h.lt_mask && !h.gt_mask), "head: the BCE / CE criteria have no bounds (lt_mask / gt_mask)");
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
    const int nc = h.loss == DMPNN_LOSS_CE ? h.n_classes : 1;           // logits per task
    const int t = t_out / nc;                                           // tasks (= columns of `targets`)

    // ---- forward ----
    if (!bounds_done) DMPNN_TRY(dmpnn_molagg_bounds(h.batch, nV, B, ws + L.bounds, dmpnn_molagg_ws_bytes(B), stream));
    DMPNN_TRY(dmpnn_molagg_fwd(Hv, ldhv, nV, d, B, ws + L.bounds, h.agg_mode, h.agg_norm, Hm, d, stream));
    const float* Z = Hm;
    float* mean = reinterpret_cast<float*>(ws + L.mean);
    float* invstd = reinterpret_cast<float*>(ws + L.invstd);
    if (h.bn_weight) {
        BnArgs b{Hm, d, reinterpret_cast<float*>(ws + L.Z), d, h.bn_weight, h.bn_bias, h.bn_running_mean, h.bn_running_var, mean, invstd,
                 B, (int)d, h.bn_eps, h.bn_momentum, h.bn_training, h.bn_num_batches_tracked};
        if (B <= kBnRegRows * kBnLanes) hipLaunchKernelGGL(k_bn_fwd<true>, dim3((unsigned)((d + kBnCols - 1) / kBnCols)), dim3(1024), 0, s, b);
        else hipLaunchKernelGGL(k_bn_fwd<false>, dim3((unsigned)((d + kBnCols - 1) / kBnCols)), dim3(1024), 0, s, b);
        DMPNN_CHECK_LAUNCH("k_bn_fwd");
        Z = reinterpret_cast<float*>(ws + L.Z);
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
        LossArgs q{h.preds, t_out, h.targets, t, h.weights, h.task_weights, h.lt_mask, h.gt_mask, want_grad ? gP : nullptr, t_out, h.loss_out, B, t, h.loss, nc};
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
                OutAllArgs qa{q, LossArgs{h.preds, t, h.targets, t, h.weights, h.task_weights, h.lt_mask, h.gt_mask, nullptr, t, h.loss_out, B, t, h.loss, 1}};
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
    const dmpnn_fwd_args& f = a->bwd.f;
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
        if (split_done) {   // (the weight pre-split rode in K0's launch)
            dmpnn_fwd_args f2 = f;
            f2.flags |= DMPNN_F_WSPLIT_READY;
            DMPNN_TRY(dmpnn_forward(&f2, stream));
        } else
            DMPNN_TRY(dmpnn_forward(&f, stream));
        // (a whole step in one call: the first predictor layer's weight gradient rides in the block's backward launches; a staged
        //  step — data parallel — has the head's gradients final after this stage, so nothing is deferred there)
        DMPNN_TRY(head_run(&a->head, f.out, f.ldout, stream, bounds_done, (stages & DMPNN_STEP_BACKWARD) ? &rider : nullptr));
    }
    if (stages & DMPNN_STEP_BACKWARD) {
        bool rode = false;
        DMPNN_TRY(backward_impl(&a->bwd, stream, rider.Z ? &rider : nullptr, &rode));
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
