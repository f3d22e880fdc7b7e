// How many bytes per clock can a CU pull from L2 (or HBM) into LDS with global_load_lds_dwordx4, when every CU does it at once?
// The convolution kernels of the 16-bit pipe re-stream their packed weights for every pixel tile (L2 hits) next to the source halo:
// igemm_split16_kernel moves ~100 KB per 16-channel chunk and workgroup (74 % weights), s16g_kernel 36 KB of weights per 41 KB of halo.
// This probe tells whether that traffic is free (far below the L2 -> CU rate) or the bound.
//   region = bytes every workgroup cycles through (weights-like: ALL workgroups read the SAME region -> L2 / MALL hits)
//   private = 1: every workgroup has its own region (HBM stream when blocks * region >> the caches)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l2_to_lds.hip -o tools/probes/_build/l2_to_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const u32x4* g, u32x4* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// each wave copies 1 KiB per instruction; INFL instructions in flight per wave
template <int INFL>
__global__ __launch_bounds__(256) void k(const u32x4* src, long long region16, int priv, int iters, unsigned* out, long long* clk) {
    __shared__ __attribute__((aligned(16))) u32x4 smem[4 * INFL * 64];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4* base = src + (priv ? (long long)blockIdx.x * region16 : 0);
    long long off = (long long)wid * 64 + lane;              // waves interleave 1 KiB pieces
    // de-phase the workgroups over the region so that they do not all hit the same lines at once
    off += ((long long)blockIdx.x * 4099 * 256) % region16;
    long long c0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < INFL; ++q) {
            if (off >= region16) off -= region16;
            glds16(base + off, smem + (wid * INFL + q) * 64);
            off += 256;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70 | (INFL / 2));        // keep half of them in flight
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = clock64() - c0;
    out[blockIdx.x * 256 + threadIdx.x] = smem[threadIdx.x][0];
}

int main() {
    const long long maxbytes = 2ll << 30;
    u32x4* src; unsigned* out; long long* clk;
    hipMalloc(&src, maxbytes); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 64);
    hipMemset(src, 1, maxbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; long long region; int priv; int wgs_per_cu; } cases[] = {
        {"shared 1.2 MB (one layer's packed weights), 1 WG/CU", 1179648, 0, 1},
        {"shared 1.2 MB, 2 WG/CU", 1179648, 0, 2},
        {"shared 1.2 MB, 4 WG/CU", 1179648, 0, 4},
        {"shared 16 MB, 2 WG/CU", 16ll << 20, 0, 2},
        {"shared 128 MB (MALL-sized), 2 WG/CU", 128ll << 20, 0, 2},
        {"private 2 MB per WG (HBM stream, 1 GB total), 2 WG/CU", 2ll << 20, 1, 2},
        {"private 4 MB per WG (HBM stream), 1 WG/CU", 4ll << 20, 1, 1},
    };
    for (auto& c : cases) {
        const int blocks = 256 * c.wgs_per_cu, iters = 2000;
        const int INFL = 8;
        hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, src, c.region / 16, c.priv, 100, out, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, src, c.region / 16, c.priv, iters, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long cyc; hipMemcpy(&cyc, clk, 8, hipMemcpyDeviceToHost);
        const double bytes = (double)blocks * 4 * iters * INFL * 1024.0;
        printf("%-58s %7.2f TB/s  %6.1f B/clk/CU  (%.3f ms, %lld shader cycles, clock %.0f MHz)\n", c.name, bytes / ms / 1e9,
               bytes / 256.0 / (double)cyc, ms, cyc, cyc / (ms * 1e3));
    }
    return 0;
}
