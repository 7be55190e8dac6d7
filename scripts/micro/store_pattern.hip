// Micro-benchmark (round 3): what does a training-step tile kernel's store PATTERN cost on this box?
//   A: every wave instruction writes 1 KB contiguous (64 lanes x 16 B)              -- tile_to_global (H0, H^(t))
//   B: every wave instruction writes 16 rows x 64 B (4 lanes x 16 B per row)        -- the fragment stores (M^(t), gZ^(t))
//   C: B, but two consecutive instructions complete each 128-byte line
//   R: reads in pattern B (the backward kernel's loads of the kept tensors)
// 210 workgroups x 256 threads, each writes its own 48 rows x 1200 B x SLOTS like one tile; repeated REPS times per launch.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/store_pattern.hip -o scripts/micro/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int LD = 300, ROWS = 48, SLOTS = 5;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* __restrict__ base, long long slot_stride, int reps, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    float* tile = base + (long long)blockIdx.x * ROWS * LD;
    float acc = 0.f;
    for (int rep = 0; rep < reps; ++rep)
        for (int s = 0; s < SLOTS; ++s) {
            float* T = tile + s * slot_stride;
            const float4 v = make_float4(rep, s, tid, 1.f);
            if (MODE == 0) {          // A: row-major float4 per thread, 75 quads per row
                for (int it = tid; it < ROWS * 75; it += 256) *reinterpret_cast<float4*>(T + (it / 75) * LD + 4 * (it % 75)) = v;
            } else if (MODE == 1 || MODE == 2) {   // B / C: wave w owns columns 80w..; (ct, jt): row jt*16+li, col 80w + 16ct + 4lg
                for (int jt = 0; jt < 3; ++jt)
                    for (int ct = 0; ct < 5; ++ct) {
                        const int col = wave * 80 + ct * 16 + lg * 4;
                        if (col < LD) *reinterpret_cast<float4*>(T + (jt * 16 + li) * LD + col) = v;
                    }
            } else {                  // R: reads in pattern B
                for (int jt = 0; jt < 3; ++jt)
                    for (int ct = 0; ct < 5; ++ct) {
                        const int col = wave * 80 + ct * 16 + lg * 4;
                        if (col < LD) { const float4 y = *reinterpret_cast<const float4*>(T + (jt * 16 + li) * LD + col); acc += y.x + y.w; }
                    }
            }
        }
    if (acc == 12345.f) *sink = acc;
}

int main() {
    const int tiles = 210;
    const long long slot = (long long)tiles * ROWS * LD;
    float* buf; float* sink;
    CHECK(hipMalloc(&buf, slot * SLOTS * sizeof(float)));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, slot * SLOTS * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double mb = (double)slot * SLOTS * 4 / 1e6;
    const char* names[4] = {"A full-line stores (1 KB / wave instr)", "B 64-byte pieces (16 rows x 64 B / wave instr)", "B again", "R reads, 64-byte pieces"};
    for (int round = 0; round < 2; ++round)
        for (int mode = 0; mode < 4; ++mode) {
            for (int reps : {1, 4}) {
                float best = 1e9f;
                for (int it = 0; it < 20; ++it) {
                    CHECK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(tiles), dim3(256), 0, 0, buf, slot, reps, sink);
                    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(tiles), dim3(256), 0, 0, buf, slot, reps, sink);
                    else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(tiles), dim3(256), 0, 0, buf, slot, reps, sink);
                    else hipLaunchKernelGGL(k<3>, dim3(tiles), dim3(256), 0, 0, buf, slot, reps, sink);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (it >= 5 && ms < best) best = ms;
                }
                printf("%-52s reps=%d  %7.1f us  %6.1f MB  %5.2f TB/s\n", names[mode], reps, best * 1e3, mb * reps, mb * reps / (best * 1e-3) / 1e6);
            }
        }
    return 0;
}
