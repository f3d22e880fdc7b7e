// Per-CU rate of global loads into VGPRs by WIDTH (4 / 8 / 16 bytes per lane, consecutive lanes at consecutive addresses), every CU
// loading at once, from an L2-resident region (shared by all workgroups) and from HBM (private regions).  s16g_kernel fetches its
// source halo with 4-byte loads (56 per thread and 16-channel chunk); is the instruction rate of the vector-memory pipe the bound?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/load_width.hip -o tools/probes/_build/load_width
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <typename T, int INFL>
__global__ __launch_bounds__(256) void k(const T* src, long long regionT, int priv, int iters, unsigned* out, long long* clk) {
    const T* base = src + (priv ? (long long)blockIdx.x * regionT : 0);
    long long off = threadIdx.x + ((long long)blockIdx.x * 4099 * 256) % regionT;
    unsigned acc = 0;
    long long c0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        T v[INFL];
#pragma unroll
        for (int q = 0; q < INFL; ++q) {
            if (off >= regionT) off -= regionT;
            v[q] = __builtin_nontemporal_load(base + off) ;
            off += 256;
        }
#pragma unroll
        for (int q = 0; q < INFL; ++q) {
            if constexpr (sizeof(T) == 4) acc ^= v[q];
            else acc ^= v[q][0];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = clock64() - c0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename T>
void run(const char* name, const void* src, long long region, int priv, int wgs_per_cu, unsigned* out, long long* clk) {
    const int blocks = 256 * wgs_per_cu, INFL = 16;
    const int iters = (int)(4096ll * 16 / sizeof(T) / 16) * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<T, INFL>), dim3(blocks), dim3(256), 0, 0, (const T*)src, region / (long long)sizeof(T), priv, 50, out, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, INFL>), dim3(blocks), dim3(256), 0, 0, (const T*)src, region / (long long)sizeof(T), priv, iters, out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 256 * iters * INFL * sizeof(T);
    printf("%-44s %2zu B/lane  %7.2f TB/s  %6.1f B/clk/CU at 2.1 GHz   %.2f G wave-instructions/s per CU\n", name, sizeof(T), bytes / ms / 1e9,
           bytes / ms / 1e3 / 256.0 / 2.1e3 * 1e-3 * 1e3 / 1e3, (double)blocks * 4 * iters * INFL / ms / 1e6 / 256.0);
}

int main() {
    const long long maxbytes = 2ll << 30;
    void* src; unsigned* out; long long* clk;
    hipMalloc(&src, maxbytes); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 64);
    hipMemset(src, 1, maxbytes);
    for (int wg = 2; wg <= 8; wg *= 2) {
        printf("-- %d workgroups of 256 threads per CU, 16 loads in flight per thread\n", wg);
        run<unsigned>("L2-resident (shared 2 MB)", src, 2ll << 20, 0, wg, out, clk);
        run<u32x2>("L2-resident (shared 2 MB)", src, 2ll << 20, 0, wg, out, clk);
        run<u32x4>("L2-resident (shared 2 MB)", src, 2ll << 20, 0, wg, out, clk);
        run<unsigned>("HBM stream (private 1 MB per workgroup)", src, 1ll << 20, 1, wg, out, clk);
        run<u32x4>("HBM stream (private 1 MB per workgroup)", src, 1ll << 20, 1, wg, out, clk);
    }
    return 0;
}
