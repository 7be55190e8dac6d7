// Micro-benchmark for the next step on the weight-gradient kernel: gW[N,K] = Z[M,N]^T . X[M,K] in fp32 MFMA
// 16x16x4 with both operands reduction-major straight from global memory, split over rows, slab per split.
//   mode 0   the product kernel's tiling: 64x64 outputs per workgroup, the four waves take different rows (k-steps
//            w, w+4, ...), cross-wave reduction through LDS; 16 flop per operand byte out of L2;
//   mode 1   128x128 outputs per workgroup: 2x2 waves, each a 64x64 quadrant over ALL rows of the split — the two waves
//            that share an operand block hit the same L1 lines, 32 flop per operand byte out of L2, no LDS reduction.
// Both use the XCD-aware (split, tile) order of the product kernel.  Prints the time of each and checks both against a
// double-precision host reference on sampled entries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/wgrad_tiles.hip -o /tmp/wgrad_tiles && /tmp/wgrad_tiles [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Args {
    const float* Z; const float* X; int M, N, K;
    float* slab; long long slab_stride; int ldk;
    int splits, rows_per_wg;
};

constexpr int U = 4;  // k-steps (of 4 rows) per batch of loads

// one wave: acc[jn][jk] += sum over its rows of Z[m][n0 + 4 li + jn] * X[m][k0 + 4 li' + jk]   (interleaved 16x16 tiles)
__device__ __forceinline__ void wave_loop(const Args& a, f32x4 (&acc)[4][4], long long m_first, int m_stride, long long m_hi, int n0,
                                          int k0, int li, int lg) {
    auto load = [&](long long mb, float4 (&z)[U], float4 (&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long m = mb + (long long)m_stride * u + lg;
            const bool mok = m < m_hi;
            const long long mc = mok ? m : 0;
            const int n = n0 + 4 * li, k = k0 + 4 * li;
            // unconditional 16-byte loads from a clamped address, zeroed by select (N, K multiples of 4: a vector is all in or all out)
            const bool zok = mok && n < a.N, xok = mok && k < a.K;
            float4 zz = *reinterpret_cast<const float4*>(a.Z + (zok ? mc * a.N + n : 0));
            float4 xx = *reinterpret_cast<const float4*>(a.X + (xok ? mc * a.K + k : 0));
            if (!zok) zz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!xok) xx = make_float4(0.f, 0.f, 0.f, 0.f);
            z[u] = zz; x[u] = xx;
        }
    };
    auto mfma = [&](const float4 (&z)[U], const float4 (&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float zz[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
            const float xx[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int jk = 0; jk < 4; ++jk) acc[jn][jk] = __builtin_amdgcn_mfma_f32_16x16x4f32(zz[jn], xx[jk], acc[jn][jk], 0, 0, 0);
        }
    };
    long long mb = m_first;
    if (mb >= m_hi) return;
    float4 z0[U], x0[U], z1[U], x1[U];
    load(mb, z0, x0);
    for (;;) {
        const long long mb1 = mb + (long long)m_stride * U;
        if (mb1 < m_hi) load(mb1, z1, x1);
        mfma(z0, x0);
        if (mb1 >= m_hi) break;
        const long long mb2 = mb1 + (long long)m_stride * U;
        if (mb2 < m_hi) load(mb2, z0, x0);
        mfma(z1, x1);
        if (mb2 >= m_hi) break;
        mb = mb2;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_wgrad(Args a) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // mode 0: [4][64][64]
    constexpr int T = MODE == 0 ? 64 : 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int nb = (a.N + T - 1) / T, kb = (a.K + T - 1) / T, tiles = nb * kb;
    const int per = gridDim.x >> 3;
    const int rank = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);  // XCD c owns the ranks [c G/8, (c+1) G/8)
    if (rank >= tiles * a.splits) return;
    const int split = rank / tiles, tile = rank - split * tiles;
    const int n0 = (tile / kb) * T, k0 = (tile % kb) * T;
    const long long m_lo = (long long)split * a.rows_per_wg;
    long long m_hi = m_lo + a.rows_per_wg;
    if (m_hi > a.M) m_hi = a.M;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* slab = a.slab + (long long)split * a.slab_stride;
    if (MODE == 0) {
        wave_loop(a, acc, m_lo + 4 * wave, 16, m_hi, n0, k0, li, lg);
        float* mine = red + wave * 4096;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int jk = 0; jk < 4; ++jk)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(4 * (lg * 4 + r) + jn) * 64 + 4 * li + jk] = acc[jn][jk][r];
        __syncthreads();
        for (int idx = tid; idx < 4096; idx += 256) {
            const int n = n0 + (idx >> 6), k = k0 + (idx & 63);
            if (n < a.N && k < a.K) slab[(long long)n * a.ldk + k] = red[idx] + red[4096 + idx] + red[8192 + idx] + red[12288 + idx];
        }
    } else {
        const int n0w = n0 + 64 * (wave >> 1), k0w = k0 + 64 * (wave & 1);
        if (n0w >= a.N || k0w >= a.K) return;  // a quadrant outside the matrix
        wave_loop(a, acc, m_lo, 4, m_hi, n0w, k0w, li, lg);
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int jk = 0; jk < 4; ++jk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0w + 4 * (lg * 4 + r) + jn, k = k0w + 4 * li + jk;
                    if (n < a.N && k < a.K) slab[(long long)n * a.ldk + k] = acc[jn][jk][r];
                }
    }
}

static void plan(int M, int tiles, int* splits, int* rows) {
    int s = 512 / tiles;
    if (s < 1) s = 1;
    const int max_s = (M + 63) / 64;
    if (s > max_s) s = max_s;
    int r = (M + s - 1) / s;
    r = (r + 15) / 16 * 16;
    *rows = r;
    *splits = (M + r - 1) / r;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 9120, N = argc > 2 ? atoi(argv[2]) : 300, K = argc > 3 ? atoi(argv[3]) : 300;
    if (N % 4 || K % 4) { printf("N and K must be multiples of 4\n"); return 1; }
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hZ((size_t)M * N), hX((size_t)M * K);
    for (auto& v : hZ) v = nd(rng);
    for (auto& v : hX) v = nd(rng);
    float *dZ, *dX, *dS;
    CK(hipMalloc(&dZ, hZ.size() * 4)); CK(hipMalloc(&dX, hX.size() * 4));
    CK(hipMemcpy(dZ, hZ.data(), hZ.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    const int ldk = (K + 3) / 4 * 4;
    const long long stride = (long long)N * ldk;
    CK(hipMalloc(&dS, (size_t)stride * 4 * 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<double>> results;
    for (int mode = 0; mode < 2; ++mode) {
        const int T = mode == 0 ? 64 : 128;
        const int tiles = ((N + T - 1) / T) * ((K + T - 1) / T);
        Args a{dZ, dX, M, N, K, dS, stride, ldk, 0, 0};
        plan(M, tiles, &a.splits, &a.rows_per_wg);
        if (a.splits > 64) { printf("too many splits\n"); return 1; }
        const int total = tiles * a.splits, grid = (total + 7) / 8 * 8;
        CK(hipMemset(dS, 0, (size_t)stride * 4 * a.splits));
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k_wgrad<0>, dim3(grid), dim3(256), 65536, 0, a);
            else hipLaunchKernelGGL(k_wgrad<1>, dim3(grid), dim3(256), 0, 0, a);
        };
        for (int i = 0; i < 5; ++i) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int g = 0; g < 5; ++g) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / 20 < best) best = ms / 20;
        }
        std::vector<float> hS((size_t)stride * a.splits);
        CK(hipMemcpy(hS.data(), dS, hS.size() * 4, hipMemcpyDeviceToHost));
        // sampled check against a double-precision reference
        double worst = 0, scale = 0;
        std::vector<double> sample;
        std::mt19937 pick(7);
        for (int t = 0; t < 64; ++t) {
            const int n = pick() % N, k = pick() % K;
            double ref = 0;
            for (int m = 0; m < M; ++m) ref += (double)hZ[(size_t)m * N + n] * hX[(size_t)m * K + k];
            double got = 0;
            for (int s = 0; s < a.splits; ++s) got += hS[(size_t)s * stride + (size_t)n * ldk + k];
            worst = fmax(worst, fabs(got - ref));
            scale = fmax(scale, fabs(ref));
            sample.push_back(got);
        }
        results.push_back(sample);
        printf("mode %d (%3dx%3d per workgroup): %3d tiles x %2d splits = %4d workgroups, %3d rows each: %8.2f us  %6.1f TFLOP/s  max err %.2e (scale %.1f)\n",
               mode, T, T, tiles, a.splits, total, a.rows_per_wg, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12, worst, scale);
    }
    double d = 0;
    for (size_t i = 0; i < results[0].size(); ++i) d = fmax(d, fabs(results[0][i] - results[1][i]));
    printf("modes agree on the samples to %.2e\n", d);
    return 0;
}
