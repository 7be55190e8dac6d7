// Micro-benchmark: how fast can the four waves of one workgroup per CU stream the SAME buffer from L2 into registers
// with 1 KiB-per-wave buffer loads (the weight-fragment pattern of the tile kernel)?  Prints cycles per 40 KiB chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream(const unsigned char* W, int n_chunks, int reps, unsigned* sink, long long* cyc, int stagger) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(W), 0, 20 * n_chunks * 2048, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r)
        for (int c0 = 0; c0 < n_chunks; ++c0) {
            const int c = stagger ? (c0 + blockIdx.x) % n_chunks : c0;
#pragma unroll
            for (int ct = 0; ct < 5; ++ct) {
                const unsigned off = (unsigned)(wave * 5 + ct) * (unsigned)(n_chunks * 2048) + (unsigned)c * 2048u + (unsigned)lane * 16u;
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 1024u, 0, 0);
            }
        }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
int main() {
    const int n_chunks = 10, reps = 20;
    unsigned char* W; unsigned* sink; long long* cyc;
    hipMalloc(&W, 20 * n_chunks * 2048); hipMemset(W, 1, 20 * n_chunks * 2048);
    hipMalloc(&sink, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int wgs : {1, 32, 212, 256}) for (int st : {0, 1}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, 0, W, n_chunks, reps, sink, cyc, st);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, 0, W, n_chunks, reps, sink, cyc, st);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<long long> h(wgs); hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
        long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
        printf("workgroups %4d stagger %d: %.1f us total, %.0f ns per 40 KiB chunk, %lld counter ticks per chunk (max over WGs), %.1f B/ns per CU\n",
               wgs, st, ms * 1e3, ms * 1e6 / (reps * n_chunks), mx / (reps * n_chunks), 40960.0 / (ms * 1e6 / (reps * n_chunks)));
    }
    return 0;
}
