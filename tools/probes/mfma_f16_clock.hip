// What the 16-bit matrix pipe sustains on this chip: back-to-back v_mfma_f32_32x32x16_f16 on every SIMD (random operands, 8
// independent accumulators per wave) while wave 0 of workgroup 0 samples s_memtime / s_memrealtime: achieved TFLOP/s and the shader
// clock UNDER that load, for 1 wave per SIMD and for issue densities thinned with s_s_sleep padding.  Stand-alone binary:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f16_clock.hip -o tools/probes/_build/mfma_f16_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NOPS>
__global__ __launch_bounds__(256) void k(const _Float16* in, float* out, int iters, long long* clk) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 1023]; b[i] = in[(threadIdx.x * 8 + i + 512) & 1023]; }
    f32x16 c[8] = {};
    long long c0 = 0, w0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = clock64(); w0 = wall_clock64(); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[q], 0, 0, 0);
            if (NOPS >= 1) __builtin_amdgcn_s_sleep(1);
            if (NOPS >= 2) __builtin_amdgcn_s_sleep(1);
            if (NOPS >= 3) { __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_s_sleep(1); }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    float s = 0;
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NOPS>
void run(const _Float16* in, float* out, long long* clk, int blocks) {
    const int iters = 40000 / (1 + NOPS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NOPS>, dim3(blocks), dim3(256), 0, 0, in, out, 2000, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NOPS>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = (double)blocks * 4 * iters * 8 * 32 * 32 * 16 * 2;
    const double mhz = (double)h[0] / (double)h[1] * 100.0;
    // issue density: MFMA cycles (8 passes x 4 = 32 per instruction... measured instead: cycles per MFMA from the stamps)
    printf("s_sleep padding %d, %d workgroups (%d waves/SIMD): %7.1f TFLOP/s  %.3f ms  shader clock %.0f MHz  %.1f cycles per MFMA\n", NOPS, blocks,
           blocks / 256, fl / ms / 1e9, ms, mhz, (double)h[0] / ((double)iters * 8));
}

int main() {
    _Float16 h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    _Float16* in; float* out; long long* clk;
    hipMalloc(&in, sizeof(h)); hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&clk, 16);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int blocks = 256; blocks <= 512; blocks *= 2) {
        run<0>(in, out, clk, blocks);
        run<1>(in, out, clk, blocks);
        run<2>(in, out, clk, blocks);
        run<3>(in, out, clk, blocks);
    }
    return 0;
}
