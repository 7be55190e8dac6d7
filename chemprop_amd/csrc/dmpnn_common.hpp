// Shared declarations for the gfx950 D-MPNN engine (internal; the public boundary is include/dmpnn.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "dmpnn.h"

namespace dmpnn {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
void count_launch();

#define DMPNN_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            ::dmpnn::set_error(__VA_ARGS__);       \
            return DMPNN_EINVAL;                   \
        }                                          \
    } while (0)

#define DMPNN_CHECK_LAUNCH(name)                                                         \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            ::dmpnn::set_error("launch of %s failed: %s", name, hipGetErrorString(e__)); \
            return DMPNN_EHIP;                                                           \
        }                                                                                \
        ::dmpnn::count_launch();                                                         \
    } while (0)

#define DMPNN_TRY(expr)               \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != DMPNN_OK) return rc__; \
    } while (0)

// ---- plan layout ----------------------------------------------------------------------------
// One int32 blob:  hdr[16] | src[E] | dst[E] | rev[E] | row_ptr[V+1] | perm[E] | cursor[V]
// every array starts on a 16-byte boundary.
struct PlanLayout {
    int64_t src, dst, rev, row_ptr, perm, cursor, words;
};
inline int64_t align4(int64_t x) { return (x + 3) & ~int64_t(3); }
inline PlanLayout plan_layout(int64_t nV, int64_t nE) {
    PlanLayout L;
    int64_t o = DMPNN_HDR_WORDS;
    L.src = o; o += align4(nE);
    L.dst = o; o += align4(nE);
    L.rev = o; o += align4(nE);
    L.row_ptr = o; o += align4(nV + 1);
    L.perm = o; o += align4(nE);
    L.cursor = o; o += align4(nV);
    L.words = o;
    return L;
}

enum : int { PLAN_ASYMMETRIC = 1, PLAN_RANGE_ERROR = 2 };

// Device view of a plan (pointers into the blob).
struct PlanView {
    const int* hdr;
    const int* src;
    const int* dst;
    const int* rev;
    const int* row_ptr;
    const int* perm;
};
inline PlanView plan_view(const void* plan, int64_t nV, int64_t nE) {
    const int* p = static_cast<const int*>(plan);
    PlanLayout L = plan_layout(nV, nE);
    return PlanView{p, p + L.src, p + L.dst, p + L.rev, p + L.row_ptr, p + L.perm};
}

// ---- activations ----------------------------------------------------------------------------
__device__ __forceinline__ float apply_act(float z, int act, float slope) {
    switch (act) {
        case DMPNN_ACT_RELU: return z < 0.f ? 0.f : z;
        case DMPNN_ACT_LEAKYRELU:
        case DMPNN_ACT_PRELU: return z > 0.f ? z : slope * z;
        case DMPNN_ACT_TANH: return tanhf(z);
        case DMPNN_ACT_ELU: return z > 0.f ? z : expm1f(z);
        default: return z;
    }
}
__device__ __forceinline__ float4 apply_act4(float4 z, int act, float slope) {
    if (act == DMPNN_ACT_NONE) return z;
    return make_float4(apply_act(z.x, act, slope), apply_act(z.y, act, slope),
                       apply_act(z.z, act, slope), apply_act(z.w, act, slope));
}
// d tau / dz given pre-activation z (when available) or output y.
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
    switch (act) {
        case DMPNN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case DMPNN_ACT_LEAKYRELU: return y > 0.f ? 1.f : slope;  // slope > 0: sign(y) == sign(z)
        case DMPNN_ACT_TANH: return 1.f - y * y;
        case DMPNN_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
        default: return 1.f;
    }
}

// ---- host launchers implemented in the .hip files -------------------------------------------
int launch_prepare(const int64_t* edge_index, const int64_t* rev, int64_t nV, int64_t nE, int* plan,
                   hipStream_t s);
int launch_message(const PlanView& pv, int64_t nV, int64_t nE, int64_t h, const float* Hin,
                   int64_t ld_in, float* M, int64_t ld_m, int act, float slope,
                   const float* slope_ptr, unsigned flags, hipStream_t s);
int launch_aggregate(const PlanView& pv, int64_t nV, int64_t nE, int64_t h, const float* Hin,
                     int64_t ld_in, float* Mv, int64_t ld_mv, int act, float slope,
                     const float* slope_ptr, hipStream_t s);
int launch_linear(const dmpnn_gemm_args& a, hipStream_t s);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace dmpnn
