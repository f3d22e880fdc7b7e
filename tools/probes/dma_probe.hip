// Measurement probe (not part of the product): raw global -> LDS copy throughput per CU on gfx950 for the access patterns of the
// split-16 convolution loaders.  W loader waves per workgroup stream 1 KiB global_load_lds_dwordx4 instructions with at most
// D wave-instructions in flight per wave, from (shared = every workgroup the same 4 MiB, private = its own 4 MiB).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_probe.hip -o tools/probes/_build/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(256) void probe(const u32x4* src, size_t wg_stride16, int iters, int region16, unsigned* sink) {
    __shared__ u32x4 lds[4 * 64 * 16];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const u32x4* s = src + (size_t)blockIdx.x * wg_stride16 + lane;
    int off = wid * 64;
    for (int i = 0; i < iters; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + off),
                                         (__attribute__((address_space(3))) void*)(lds + (wid * 16 + (i & 15)) * 64), 16, 0, 0);
        off += nw * 64;
        if (off >= region16) off -= region16;
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((D - 1) & 15) | (((D - 1) >> 4) << 14));     // keep <= D - 1 older ones in flight
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = lds[5][0];
}

int main() {
    const size_t region = 4u << 20;                    // bytes walked per workgroup
    const int wgs = 256;
    u32x4* buf; unsigned* sink;
    hipMalloc(&buf, region * wgs); hipMalloc(&sink, 4 * wgs);
    hipMemset(buf, 1, region * wgs);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;                            // 1 KiB each per wave
    for (int shared = 1; shared >= 0; --shared)
        for (int waves : {1, 2, 4})
            for (int depth : {8, 16, 32, 60}) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0);
                    const size_t stride16 = shared ? 0 : region / 16;
                    const int r16 = (int)(region / 16);
#define L(DD) hipLaunchKernelGGL((probe<DD>), dim3(wgs), dim3(64 * waves), 0, 0, buf, stride16, iters, r16, sink)
                    if (depth == 8) L(8); else if (depth == 16) L(16); else if (depth == 32) L(32); else L(60);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double bytes = (double)wgs * waves * iters * 1024.0;
                printf("%-7s waves %d depth %2d: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip\n", shared ? "shared" : "private", waves,
                       depth, best * 1e3, bytes / wgs / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
            }
    return 0;
}
