// Calibration probe: sustained v_mfma_f32_32x32x2_f32 rate with random vs zero operands (DVFS), 1..4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    float a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *in, *out; float h[512];
    hipMalloc(&in, 2048); hipMalloc(&out, 4096 * 256 * 4);
    for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < 512; ++i) h[i] = mode == 0 ? 0.f : mode == 1 ? (rand() / (float)RAND_MAX - 0.5f) * 1e-3f : (rand() / (float)RAND_MAX - 0.5f) * 4.f * (1.0f + (rand() % 7));
        hipMemcpy(in, h, 2048, hipMemcpyHostToDevice);
        for (int blocks = 512; blocks <= 1024; blocks *= 2) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            int iters = 20000;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, 1000);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)blocks * 4 * iters * 4 * 32 * 32 * 2 * 2;
            printf("%s operands, %d blocks (%d waves/SIMD): %.1f TFLOP/s  (%.3f ms)\n", mode == 0 ? "zero" : mode == 1 ? "random 1e-3" : "random O(1) wide", blocks, blocks / 256, fl / ms / 1e9, ms);
        }
    }
    return 0;
}
