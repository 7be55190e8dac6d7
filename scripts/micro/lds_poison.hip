// Debug aid (round 6): fill the whole LDS of every CU with a NaN pattern, so that a kernel that reads LDS it never wrote shows it
// (LDS is not cleared between kernels: on a warm box it holds the previous kernels' finite leftovers).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC scripts/micro/lds_poison.hip -o scripts/micro/liblds_poison.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void k_lds_poison(unsigned pat, unsigned* sink) {
    extern __shared__ unsigned lds[];
    const int n = (160 * 1024 - 64) / 4;
    for (int i = threadIdx.x; i < n; i += 1024) lds[i] = pat;
    __syncthreads();
    if (sink && lds[(threadIdx.x * 7) % n] == 0x12345u) sink[0] = 1;   // (keeps the stores)
}
extern "C" int lds_poison(unsigned pat, void* stream) {
    static bool set = false;
    if (!set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_poison), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess) return 1;
        set = true;
    }
    // (one workgroup per CU at this LDS size; three rounds so that every CU is hit whatever the dispatch order)
    hipLaunchKernelGGL(k_lds_poison, dim3(768), dim3(1024), 160 * 1024 - 64, static_cast<hipStream_t>(stream), pat, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
