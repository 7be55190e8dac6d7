// Micro-test (round 6; measured, NOT used by the kernels — see split2 in dmpnn_mega16_impl.hpp): the exact f16 pair split x s = hi + lo as four v_fma_mix*_f16 instructions per two elements (2 VALU per element)
// against the sequence hipcc emits for the C++ form (pk_mul, cvt_pk, 2 cvt back, pk_fma, cvt_pk: 3 per element) — bit for bit, over
// magnitudes that put lo into the f16 subnormals, zeros, negative zeros, infinities and NaN; and v_maximum3_f32(x, 0, 0) against
// (x > 0 ? x : 0 * x) + 0.  Prints the number of differing words.   hipcc --offload-arch=gfx950 -O3 split_probe.hip -o split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_mix(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
    unsigned h, l;   // (x s - 0: a negative zero stays one, as in f16(x s))
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 neg_lo:[0,0,1]" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 neg_lo:[0,0,1]" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
    hi = h; lo = l;
}
__global__ void k(const float* x, unsigned* out, unsigned* ref, float* r, float* rref, float s, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned hi, lo;
    split2_mix(x[2 * i], x[2 * i + 1], s, hi, lo);
    out[2 * i] = hi; out[2 * i + 1] = lo;
    float a = x[2 * i] * s, b = x[2 * i + 1] * s;
    h2 rh = h2{(_Float16)a, (_Float16)b};
    h2 rl = h2{(_Float16)(a - (float)rh[0]), (_Float16)(b - (float)rh[1])};
    ref[2 * i] = __builtin_bit_cast(unsigned, rh); ref[2 * i + 1] = __builtin_bit_cast(unsigned, rl);
    for (int j = 0; j < 2; ++j) {
        const float v = x[2 * i + j];
        r[2 * i + j] = __builtin_elementwise_maximum(v, 0.f);
        rref[2 * i + j] = (v > 0.f ? v : 0.f * v) + 0.f;
    }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::uniform_int_distribution<int> e(-40, 6);
    for (int i = 0; i < n; ++i) h[i] = ldexpf(u(g), e(g));
    h[0] = 0.f; h[1] = -0.f; h[2] = INFINITY; h[3] = -INFINITY; h[4] = NAN; h[5] = -NAN; h[6] = 1e-45f; h[7] = -1e-45f; h[8] = 65504.f; h[9] = 7.99f;
    float *x, *r, *rr; unsigned *o, *f;
    hipMalloc(&x, n * 4); hipMalloc(&o, n * 4); hipMalloc(&f, n * 4); hipMalloc(&r, n * 4); hipMalloc(&rr, n * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    for (float s : {1.f, 8192.f, 1.f / 1024.f, 16384.f}) {
        hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, x, o, f, r, rr, s, n);
        std::vector<unsigned> ho(n), hf(n), hr(n), hrr(n);
        hipMemcpy(ho.data(), o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hf.data(), f, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), r, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hrr.data(), rr, n * 4, hipMemcpyDeviceToHost);
        long bad = 0, badr = 0, sub = 0;
        for (int i = 0; i < n; ++i) {
            if (ho[i] != hf[i]) { if (bad < 5) printf("  split word %d: x = %g %g  mix %08x  ref %08x\n", i, h[i & ~1], h[i | 1], ho[i], hf[i]); ++bad; }
            if ((i & 1) && ((hf[i] & 0x7c00u) == 0 && (hf[i] & 0x3ffu))) ++sub;
            const bool nan_a = (hr[i] & 0x7fffffffu) > 0x7f800000u, nan_b = (hrr[i] & 0x7fffffffu) > 0x7f800000u;
            if (hr[i] != hrr[i] && !(nan_a && nan_b)) { if (badr < 5) printf("  relu %d: x = %g  maximum3 %08x  ref %08x\n", i, h[i], hr[i], hrr[i]); ++badr; }
        }
        printf("s = %g: split words differing %ld of %d (lo subnormal in %ld), relu differing %ld\n", s, bad, n, sub, badr);
    }
    return 0;
}
