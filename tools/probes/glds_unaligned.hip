// Probe: does global_load_lds_dwordx4 accept a global address that is only 4-byte aligned?  (and dwordx1 for reference)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* src, float* out, int off) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    const float* g = src + off + threadIdx.x * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = lds[threadIdx.x * 4 + i];
}
int main() {
    float h[512], *src, *out, r[256];
    for (int i = 0; i < 512; ++i) h[i] = (float)i;
    hipMalloc(&src, sizeof(h)); hipMalloc(&out, sizeof(r));
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int off = 0; off < 4; ++off) {
        hipMemset(out, 0, sizeof(r));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, off);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i] != (float)(i + off);
        printf("glds_b128 global offset %d floats: %s, mismatches %d (r[0..4]=%g %g %g %g %g)\n", off, hipGetErrorString(e), bad, r[0], r[1], r[2], r[3], r[4]);
    }
    return 0;
}
