// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does lane l receive, given per-lane addresses?
// LDS holds u16 element index e at element e.  Every lane passes its own 8-byte-aligned address; prints, per lane, the four
// element indices it got.  Usage: tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr_bytes, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int a = addr_bytes[threadIdx.x];
    typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 f16x4;
    auto p = reinterpret_cast<__attribute__((address_space(3))) f16x4*>((__attribute__((address_space(3))) char*)lds + a);
    f16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    s4 u = __builtin_bit_cast(s4, v);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)u[j];
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    int *d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    // hypothesis A: per 16-lane group, lane i points at piece (row i >> 2, col quad i & 3) of a [4][16] block with row stride RS elements
    for (int RS : {16, 64, 648}) {
        for (int l = 0; l < 64; ++l) { const int i = l & 15, g = l >> 4; h_addr[l] = 2 * ((g * 4 + (i >> 2)) * RS + (i & 3) * 4); }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("RS=%d (element = row*RS + col)\n", RS);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h_out[l * 4 + j] / RS, h_out[l * 4 + j] % RS);
            printf("\n");
        }
    }
    return 0;
}
