// f4: the optimizer step of a data-parallel training step as ONE launch over the flat buffers.
//
// chemprop trains with torch.optim.Adam (models/model.py:208-231, Noam learning-rate schedule on top).  torch's foreach Adam
// is ~8 launches over the parameter list; with the gradients already in ONE flat buffer (distributed.GradSync) and the
// parameters as views of another, the update is one HBM-bound elementwise pass — 0.3 M parameters: launch-bound, ~3 us.
// Arithmetic = torch.optim.Adam (amsgrad = false, maximize = false), in this order:
//     g   = grad + weight_decay * p
//     m   = beta1 * m + (1 - beta1) * g
//     v   = beta2 * v + (1 - beta2) * g * g
//     p  -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)          bc_i = 1 - beta_i^step
// `lr` and the two bias corrections come from a 4-float DEVICE array (lr, bc1, sqrt(bc2), grad_scale) when `dev_scalars` is
// given — a captured hipGraph replays with fresh values written by a 16-byte copy — else from the arguments.
// `grad_scale` multiplies the gradient first (1 / world_size after a SUM all-reduce).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dmpnn_common.hpp"

namespace dmpnn {
namespace {

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n4, float lr, float beta1, float beta2, float eps,
                                              float wd, float bc1, float sqrt_bc2, float grad_scale, const float* __restrict__ dev) {
    if (dev) { lr = dev[0]; bc1 = dev[1]; sqrt_bc2 = dev[2]; grad_scale = dev[3]; }
    const float step = lr / bc1, inv = 1.f / sqrt_bc2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float* P = &pp.x; float* M = &mm.x; float* V = &vv.x;
        const float* G = &gg.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gr = G[c] * grad_scale + wd * P[c];
            M[c] = beta1 * M[c] + (1.f - beta1) * gr;
            V[c] = beta2 * V[c] + (1.f - beta2) * gr * gr;
            P[c] -= step * (M[c] / (sqrtf(V[c]) * inv + eps));
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}

// ---- gradient clipping over the flat gradient buffer (Lightning: Trainer(gradient_clip_val=...), cli/train.py:1937) ----------
// torch.nn.utils.clip_grad_norm_ (what lightning's precision plugin calls, algorithm "norm"): total = || g ||_2 over ALL gradients,
// coef = min(1, max_norm / (total + 1e-6)), g *= coef.  Two launches, no host read: per-block partial sums of squares, then every
// block re-reduces the (<= 256) partials by itself and scales its share.  `grad_scale` (1 / world after a SUM all-reduce) enters the
// norm, not the buffer (dmpnn_adam_step applies it): the clipped quantity is the averaged gradient, as under DDP.
constexpr int kClipBlocks = 256;

__global__ __launch_bounds__(256) void k_clip_sqsum(const float* __restrict__ g, int64_t n4, float* __restrict__ partial) {
    float acc = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ float w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(256) void k_clip_scale(float* __restrict__ g, int64_t n4, const float* __restrict__ partial, int n_partial,
                                                    float max_norm, float grad_scale, float* __restrict__ norm_out) {
    float acc = (int)threadIdx.x < n_partial ? partial[threadIdx.x] : 0.f;
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ float w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float total = sqrtf(w[0] + w[1] + w[2] + w[3]) * grad_scale;
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
    const float coef = max_norm / (total + 1e-6f);
    if (coef >= 1.f) return;     // (torch multiplies by the coefficient clamped to 1: the identity; a NaN coefficient — a non-finite
                                 //  total norm — is NOT >= 1: it multiplies through, as in torch, and the step becomes visibly NaN)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<float4*>(g)[i];
        v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
        reinterpret_cast<float4*>(g)[i] = v;
    }
}

// torch.nn.utils.clip_grad_value_ (algorithm "value"): clamp every (averaged) gradient element to [-c, c]
__global__ __launch_bounds__(256) void k_clip_value(float* __restrict__ g, int64_t n4, float c) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<float4*>(g)[i];
        v.x = fminf(fmaxf(v.x, -c), c); v.y = fminf(fmaxf(v.y, -c), c); v.z = fminf(fmaxf(v.z, -c), c); v.w = fminf(fmaxf(v.w, -c), c);
        reinterpret_cast<float4*>(g)[i] = v;
    }
}

}  // namespace
}  // namespace dmpnn

using namespace dmpnn;

extern "C" size_t dmpnn_clip_grad_ws_bytes(void) { return (size_t)(kClipBlocks + 4) * sizeof(float); }

extern "C" int dmpnn_clip_grad(float* g, int64_t n, float clip_val, int32_t mode, float grad_scale, float* ws, void* stream) {
    DMPNN_CHECK_ARG(n >= 0 && n % 4 == 0, "clip_grad: the flat buffer holds whole 16-byte groups");
    DMPNN_CHECK_ARG(mode == DMPNN_CLIP_NORM || mode == DMPNN_CLIP_VALUE, "clip_grad: mode is DMPNN_CLIP_NORM or DMPNN_CLIP_VALUE");
    if (n == 0 || !(clip_val > 0.f)) return DMPNN_OK;
    DMPNN_CHECK_ARG(g && aligned16(g), "clip_grad: the buffer must be 16-byte aligned");
    DMPNN_CHECK_ARG(grad_scale > 0.f, "clip_grad: grad_scale must be positive");
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > kClipBlocks) blocks = kClipBlocks;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == DMPNN_CLIP_VALUE) {
        hipLaunchKernelGGL(k_clip_value, dim3((unsigned)blocks), dim3(256), 0, s, g, n4, clip_val / grad_scale);
        DMPNN_CHECK_LAUNCH("k_clip_value");
        return DMPNN_OK;
    }
    DMPNN_CHECK_ARG(ws && aligned16(ws), "clip_grad: the norm needs its scratch (dmpnn_clip_grad_ws_bytes)");
    hipLaunchKernelGGL(k_clip_sqsum, dim3((unsigned)blocks), dim3(256), 0, s, g, n4, ws);
    DMPNN_CHECK_LAUNCH("k_clip_sqsum");
    hipLaunchKernelGGL(k_clip_scale, dim3((unsigned)blocks), dim3(256), 0, s, g, n4, ws, (int)blocks, clip_val, grad_scale, ws + kClipBlocks);
    DMPNN_CHECK_LAUNCH("k_clip_scale");
    return DMPNN_OK;
}

extern "C" int dmpnn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float bias_corr1, float sqrt_bias_corr2, float grad_scale, const float* dev_scalars,
                               void* stream) {
    DMPNN_CHECK_ARG(n >= 0 && n % 4 == 0, "adam_step: the flat buffers hold whole 16-byte groups");
    if (n == 0) return DMPNN_OK;
    DMPNN_CHECK_ARG(p && g && m && v, "adam_step: null buffer");
    DMPNN_CHECK_ARG(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adam_step: buffers must be 16-byte aligned");
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n4, lr, beta1, beta2, eps,
                       weight_decay, bias_corr1, sqrt_bias_corr2, grad_scale, dev_scalars);
    DMPNN_CHECK_LAUNCH("k_adam");
    return DMPNN_OK;
}
