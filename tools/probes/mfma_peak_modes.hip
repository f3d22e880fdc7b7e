// Reconciling the matrix-pipe ceiling: the guide quotes 2495 TFLOP/s measured for v_mfma_f32_32x32x16_{bf16,f16};
// tools/probes/mfma_f16_clock.hip measured 1.42-1.55 PFLOP/s with random operands in a 7-14 ms loop.  This probe runs the SAME
// back-to-back MFMA loop (8 independent accumulators, 1 or 2 waves per SIMD, no LDS, no memory) and varies only
//   * the operand DATA: all zero / all 1.0 / random in [-2, 2)   (toggle power of the multiplier array),
//   * the TYPE: f16 / bf16,
//   * the DURATION: one long launch (~8 ms) vs a train of ~200 us launches separated by ~200 us of idle (a sleeping kernel), which is
//     what the training step looks like to the power controller,
// and prints TFLOP/s with the shader clock measured inside the kernel (s_memtime vs the 100 MHz s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak_modes.hip -o tools/probes/_build/mfma_peak_modes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF>
__global__ __launch_bounds__(256) void k(const unsigned short* in, float* out, int iters, long long* clk) {
    union { f16x8 h; bf16x8 b; unsigned short u[8]; } a, b;
    for (int i = 0; i < 8; ++i) { a.u[i] = in[(threadIdx.x * 8 + i) & 1023]; b.u[i] = in[(threadIdx.x * 8 + i + 512) & 1023]; }
    f32x16 c[8] = {};
    long long c0 = 0, w0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = clock64(); w0 = wall_clock64(); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (BF) c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, c[q], 0, 0, 0);
            else    c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, c[q], 0, 0, 0);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    float s = 0;
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void idle_kernel(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127); }

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
static unsigned short f2b(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

template <bool BF>
void run(const char* data, unsigned short* in, float* out, long long* clk, int blocks, int mode /*0 long, 1 bursts*/) {
    unsigned short h[1024];
    for (int i = 0; i < 1024; ++i) {
        float v = data[0] == 'z' ? 0.f : data[0] == 'o' ? 1.f : (rand() / (float)RAND_MAX - 0.5f) * 4.f;
        h[i] = BF ? f2b(v) : f2h(v);
    }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = mode == 0 ? 40000 : 1000;          // 1000 x 8 MFMAs x 32 cycles = 256 k cycles = 110-180 us per launch
    const int reps = mode == 0 ? 1 : 40;
    hipLaunchKernelGGL(k<BF>, dim3(blocks), dim3(256), 0, 0, in, out, 1000, clk);
    hipDeviceSynchronize();
    double tot_ms = 0, clk_c = 0, clk_w = 0;
    for (int r = 0; r < reps; ++r) {
        if (mode == 1) hipLaunchKernelGGL(idle_kernel, dim3(1), dim3(64), 0, 0, 60);      // ~200 us of an almost idle chip
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<BF>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
        tot_ms += ms; clk_c += (double)hc[0]; clk_w += (double)hc[1];
    }
    const double fl = (double)reps * blocks * 4 * iters * 8 * 32 * 32 * 16 * 2;
    printf("%-4s %-6s %-7s %d waves/SIMD: %7.1f TFLOP/s  (%.3f ms per launch, kernel-internal clock %.0f MHz, %.1f cycles per MFMA)\n",
           BF ? "bf16" : "f16", data, mode == 0 ? "long" : "bursts", blocks / 256, fl / tot_ms / 1e9, tot_ms / reps, clk_c / clk_w * 100.0,
           clk_c / ((double)reps * iters * 8));
}

int main() {
    unsigned short* in; float* out; long long* clk;
    hipMalloc(&in, 2048); hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&clk, 16);
    const char* datas[3] = {"zero", "ones", "random"};
    for (int blocks = 256; blocks <= 512; blocks *= 2)
        for (int mode = 0; mode < 2; ++mode)
            for (int d = 0; d < 3; ++d) {
                run<false>(datas[d], in, out, clk, blocks, mode);
                run<true>(datas[d], in, out, clk, blocks, mode);
            }
    return 0;
}
