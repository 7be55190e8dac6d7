// Micro-benchmark: can a wave overlap the weight-fragment stream (L2 -> registers, 1 KiB loads) with the MFMAs that consume it?
// mode 0: loads only; 1: MFMAs only; 2: both, fragments of chunk c+2 requested at the end of chunk c (tile kernel schedule);
// 3: both, requested at the start of chunk c (three-deep ring).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned char* W, int n_chunks, int reps, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(W), 0, 20 * n_chunks * 2048, 0x00020000);
    f32x4 acc[3][5];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    h8 a[3];
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(lane * 0.001f + i);
    h8 bh[3][5], bl[3][5];
    for (int s = 0; s < 3; ++s) for (int j = 0; j < 5; ++j) for (int e = 0; e < 8; ++e) { bh[s][j][e] = (_Float16)1.f; bl[s][j][e] = (_Float16)0.5f; }
    auto load = [&](int c, h8 (&xh)[5], h8 (&xl)[5]) {
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) {
            const unsigned off = (unsigned)(wave * 5 + ct) * (unsigned)(n_chunks * 2048) + (unsigned)(c % n_chunks) * 2048u + (unsigned)lane * 16u;
            xh[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
            xl[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 1024u, 0, 0));
        }
    };
    auto mf = [&](h8 (&xh)[5], h8 (&xl)[5]) {
#pragma unroll
        for (int ct = 0; ct < 5; ++ct)
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], xh[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 5; ++ct)
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], xl[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 5; ++ct)
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(rt + 1) % 3], xh[ct], acc[rt][ct], 0, 0, 0);
    };
    const int total = reps * n_chunks;
    if (MODE == 0) {
        for (int c = 0; c < total; ++c) { load(c, bh[0], bl[0]); asm volatile("" ::: "memory"); }
    } else if (MODE == 1) {
        for (int c = 0; c < total; ++c) { mf(bh[0], bl[0]); __builtin_amdgcn_sched_barrier(0); }
    } else if (MODE == 2) {
        load(0, bh[0], bl[0]); load(1, bh[1], bl[1]);
        for (int c = 0; c < total; c += 2) {
            mf(bh[0], bl[0]); __builtin_amdgcn_sched_barrier(0); load(c + 2, bh[0], bl[0]); __builtin_amdgcn_sched_barrier(0);
            mf(bh[1], bl[1]); __builtin_amdgcn_sched_barrier(0); load(c + 3, bh[1], bl[1]); __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        load(0, bh[0], bl[0]); load(1, bh[1], bl[1]);
        for (int c = 0; c < total; c += 3) {
            load(c + 2, bh[2], bl[2]); __builtin_amdgcn_sched_barrier(0); mf(bh[0], bl[0]); __builtin_amdgcn_sched_barrier(0);
            load(c + 3, bh[0], bl[0]); __builtin_amdgcn_sched_barrier(0); mf(bh[1], bl[1]); __builtin_amdgcn_sched_barrier(0);
            load(c + 4, bh[1], bl[1]); __builtin_amdgcn_sched_barrier(0); mf(bh[2], bl[2]); __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 5; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    for (int j = 0; j < 5; ++j) s += (float)bh[0][j][0] + (float)bl[0][j][0];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const unsigned char* W, float* sink, int wgs) {
    const int n_chunks = 10, reps = 30;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, W, n_chunks, reps, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, W, n_chunks, reps, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode %d workgroups %4d: %.0f ns per chunk\n", MODE, wgs, ms * 1e6 / (reps * n_chunks));
}
int main() {
    unsigned char* W; float* sink;
    hipMalloc(&W, 20 * 10 * 2048); hipMemset(W, 0, 20 * 10 * 2048);
    hipMalloc(&sink, 1024 * 256 * 4);
    for (int wgs : {1, 212}) { run<0>(W, sink, wgs); run<1>(W, sink, wgs); run<2>(W, sink, wgs); run<3>(W, sink, wgs); }
    return 0;
}
