// Measurement probe (not part of the product): why do the loader waves of the split-16 convolution need ~290 cycles per 1 KiB
// global->LDS copy when a free-running wave needs ~66?  Same copy burst (6 x 1 KiB per iteration from a 3.5 MiB region shared
// by all workgroups), toggling the ingredients: BAR = workgroup barrier per iteration with 4 parked waves, BIG = 152 KiB of LDS,
// and the burst length.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool BAR, bool BIG, int BURST, int MODE = 0>   // MODE bits: 1 = 232 VGPRs, 2 = 3 duplicate copies (same source, same LDS
// destination) per iteration, 4 = 3 copies per iteration from a private 64 MiB stream (HBM misses), 8 = parked waves issue MFMAs,
// 16 = two bursts may stay in flight
__global__ __launch_bounds__(384) void probe(const u32x4* src, int iters, int region16, long long* out, const u32x4* priv) {
    __shared__ u32x4 lds[BIG ? 9728 : 2048];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (MODE & 1) asm volatile("v_mov_b32 v231, 0" ::: "v231");
    if (wid < 4) {
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        f32x16 acc[4] = {};
        bf16x8 a = {}, b = {};
        const long long tm0 = clock64();
        if (MODE & 64) __builtin_amdgcn_s_setprio(0);
        for (int i = 0; i < iters; ++i) {
            if (MODE & 8)
                for (int k = 0; k < 12; ++k)
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
            if (BAR) __builtin_amdgcn_s_barrier();
        }
        if (wid == 0 && lane == 0 && blockIdx.x == 0) out[4] = (clock64() - tm0) / iters;
        if (MODE & 8) { float v = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]; if (v == 123.f) out[7] = 1; }
        return;
    }
    const int kg = wid - 4;
    if (MODE & 32) __builtin_amdgcn_s_setprio(3);          // loader waves win the issue arbitration against the MFMA wave of their SIMD
    const u32x4* s = src + kg * (BURST * 64) + lane;
    int off = 0;
    long long t0 = clock64(), tb = 0;
    for (int i = 0; i < iters; ++i) {
        const long long a = clock64();
#pragma unroll
        for (int q = 0; q < BURST; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + off + q * 64),
                                             (__attribute__((address_space(3))) void*)(lds + ((i & 3) * 2 + kg) * BURST * 64 + q * 64), 16, 0, 0);
        if (MODE & 2)
            for (int q = 0; q < 3; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + off),
                                                 (__attribute__((address_space(3))) void*)(lds + ((i & 3) * 2 + kg) * BURST * 64), 16, 0, 0);
        if (MODE & 4)
            for (int q = 0; q < 3; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(priv + ((size_t)blockIdx.x * 4096 + (size_t)(i % 1300) * 3 + q) * 64 * 2 + kg * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(lds + 3072 + ((i & 1) * 6 + kg * 3 + q) * 64), 16, 0, 0);
        tb += clock64() - a;
        off += 2 * BURST * 64;
        if (off >= region16) off = 0;
        constexpr int PERIT = BURST + ((MODE & 2) ? 3 : 0) + ((MODE & 4) ? 3 : 0), KEEP = (MODE & 16) ? 2 * PERIT : PERIT;
        __builtin_amdgcn_s_waitcnt(0x0F70 | (KEEP & 15) | ((KEEP >> 4) << 14));     // one (two) iterations' copies may stay in flight
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    if (blockIdx.x == 0 && lane == 0) { out[kg * 2] = (t1 - t0) / iters; out[kg * 2 + 1] = tb / iters; }
}

int main() {
    const size_t region = 3u << 20;
    u32x4* buf; long long* out;
    u32x4* priv; hipMalloc(&priv, (size_t)256 * 4096 * 2048 + (1 << 20));
    hipMalloc(&buf, region + (1 << 20)); hipMalloc(&out, 64);
    hipMemset(buf, 1, region);
    const int iters = 2000, r16 = (int)(region / 16);
    long long h[5];
#define RUN(BAR, BIG, BURST, what) RUNM(BAR, BIG, BURST, 0, what)
#define RUNM(BAR, BIG, BURST, MODE, what)                                                                                   \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<BAR, BIG, BURST, MODE>), dim3(256), dim3(384), 0, 0, buf, iters, r16, out, priv); \
    hipDeviceSynchronize(); hipMemcpy(h, out, 40, hipMemcpyDeviceToHost);                                             \
    printf("%-52s iteration %5lld cycles, burst issue %5lld cycles (%4lld per copy), MFMA wave iteration %5lld\n", what, h[0], h[1], h[1] / BURST, h[4]);
    RUN(false, false, 6, "free-running, 32 KiB LDS, burst 6")
    RUN(false, true, 6, "free-running, 152 KiB LDS, burst 6")
    RUN(true, false, 6, "barrier + 4 parked waves, 32 KiB LDS, burst 6")
    RUN(true, true, 6, "barrier + 4 parked waves, 152 KiB LDS, burst 6")
    RUN(true, true, 10, "barrier + 4 parked waves, 152 KiB LDS, burst 10")
    RUN(true, true, 3, "barrier + 4 parked waves, 152 KiB LDS, burst 3")
    RUNM(true, true, 6, 1, "  + 232 VGPRs")
    RUNM(true, true, 6, 2, "  + 3 duplicate copies")
    RUNM(true, true, 6, 4, "  + 3 HBM-miss copies")
    RUNM(true, true, 6, 8, "  + 48 MFMAs per iteration in the other waves")
    RUNM(true, true, 6, 16, "  + two iterations in flight")
    RUNM(true, true, 6, 31, "  + all of the above")
    RUNM(true, true, 6, 9, "  + 232 VGPRs + MFMAs")
    RUNM(true, true, 6, 12, "  + HBM-miss copies + MFMAs")
    RUNM(true, true, 6, 40, "  + MFMAs, loader waves at s_setprio 3")
    RUNM(true, true, 10, 40, "  + MFMAs, loader waves at s_setprio 3, burst 10")
    RUNM(true, true, 10, 44, "  + MFMAs + HBM-miss copies, s_setprio 3, burst 10")
    RUNM(false, true, 10, 40, "  free-running + MFMAs, s_setprio 3, burst 10")
    RUNM(false, true, 10, 8, "  free-running + MFMAs, burst 10")
    return 0;
}
