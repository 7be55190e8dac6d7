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

// Latency chain: per iteration one 16-byte store to the workgroup's own rows, then one dependent 16-byte LOAD (from another buffer)
// whose result feeds the next address — s_waitcnt vmcnt(0) per iteration, so an iteration costs max(load latency, store ACK latency).
// mode 0: load only; 1: store + load; 2: store (nt) + load.  One wave per workgroup, `wgs` workgroups.
template <int MODE>
__global__ __launch_bounds__(64) void k_lat(float* __restrict__ out, const float* __restrict__ in, int iters, long long* cycles) {
    const int lane = threadIdx.x;
    float* o = out + (long long)blockIdx.x * 64 * 1024;
    const float* p = in + (long long)blockIdx.x * 64 * 1024;
    unsigned off = lane * 4;
    float acc = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1) *reinterpret_cast<float4*>(o + ((i * 64 * 4 + lane * 4) & (64 * 1024 - 1))) = make_float4(acc, i, lane, 1.f);
        if (MODE == 2) __builtin_nontemporal_store(acc + i, o + ((i * 64 * 4 + lane * 4) & (64 * 1024 - 1)));
        const float4 v = *reinterpret_cast<const float4*>(p + (off & (64 * 1024 - 4)));
        acc += v.x;
        off = off + 256 + ((unsigned)(v.y) & 3u) * 4u;   // (dependent address: the next load cannot issue before this one returns)
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 12345.f) out[0] = acc;
}

static void latency_test() {
    const int wgs_list[3] = {1, 64, 840};
    float* out; float* in; long long* cyc;
    const size_t n = (size_t)840 * 64 * 1024;
    CHECK(hipMalloc(&out, n * 4)); CHECK(hipMalloc(&in, n * 4)); CHECK(hipMalloc(&cyc, 840 * 8));
    CHECK(hipMemset(in, 0, n * 4)); CHECK(hipMemset(out, 0, n * 4));
    const char* names[3] = {"load only", "store + dependent load", "nt store + dependent load"};
    for (int w = 0; w < 3; ++w)
        for (int mode = 0; mode < 3; ++mode) {
            const int iters = 200;
            for (int rep = 0; rep < 3; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k_lat<0>, dim3(wgs_list[w]), dim3(64), 0, 0, out, in, iters, cyc);
                else if (mode == 1) hipLaunchKernelGGL(k_lat<1>, dim3(wgs_list[w]), dim3(64), 0, 0, out, in, iters, cyc);
                else hipLaunchKernelGGL(k_lat<2>, dim3(wgs_list[w]), dim3(64), 0, 0, out, in, iters, cyc);
                CHECK(hipDeviceSynchronize());
            }
            long long h[840];
            CHECK(hipMemcpy(h, cyc, wgs_list[w] * 8, hipMemcpyDeviceToHost));
            double s = 0; long long mx = 0;
            for (int i = 0; i < wgs_list[w]; ++i) { s += h[i]; if (h[i] > mx) mx = h[i]; }
            printf("latency chain, %3d waves: %-26s  %7.0f cycles / iteration (mean), %7.0f (slowest wave)   [cycle counter ticks]\n", wgs_list[w], names[mode],
                   s / wgs_list[w] / iters, (double)mx / iters);
        }
}

int main() {
    latency_test();
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
