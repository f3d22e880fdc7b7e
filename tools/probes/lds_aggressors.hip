// Co-runner kernels for tools/diag_wgrad_beside.py: which property of a workgroup that shares a CU with wgrad_split16_kernel (LDS-DMA staged
// operands) makes that kernel read wrong X fragments?  (the second side-stream difference.)
//   0 agg_alloc   16 KiB of LDS allocated, written once, then ~20 us of VALU work: LDS allocation without LDS traffic
//   1 agg_lds     16 KiB of LDS, ds_write / ds_read transposes in a loop, no global traffic in the loop
//   2 agg_glob    streaming global copy, no LDS at all
//   3 agg_lds1k   1 KiB of LDS, the same ds traffic
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o _build/liblds_aggressors.so lds_aggressors.hip
#include <hip/hip_runtime.h>

template <int WORDS>
__global__ __launch_bounds__(256) void agg_lds_kernel(float* out, int iters, int traffic) {
    __shared__ float tile[WORDS];
    const int t = threadIdx.x;
    for (int i = t; i < WORDS; i += 256) tile[i] = (float)(i + blockIdx.x);
    __syncthreads();
    float acc = 0.f;
    if (traffic) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll 8
            for (int j = 0; j < 8; ++j) {
                const int a = (t * 33 + j * 257 + it) % WORDS;
                acc += tile[a];
                tile[(a + 64) % WORDS] = acc;
            }
            __syncthreads();
        }
    } else {
        for (int it = 0; it < iters * 24; ++it) acc = acc * 1.0001f + (float)it;
        acc += tile[t % WORDS];
    }
    if (acc == 12345.678f) out[blockIdx.x * 256 + t] = acc;
}

__global__ __launch_bounds__(256) void agg_glob_kernel(const float4* in, float4* out, long n) {
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += gridDim.x * 256l) out[i] = in[i];
}

extern "C" int launch_aggressor(int kind, int blocks, int iters, const void* in, void* out, long n16, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL((agg_lds_kernel<4096>), dim3(blocks), dim3(256), 0, st, (float*)out, iters, 0);
    else if (kind == 1) hipLaunchKernelGGL((agg_lds_kernel<4096>), dim3(blocks), dim3(256), 0, st, (float*)out, iters, 1);
    else if (kind == 2) hipLaunchKernelGGL(agg_glob_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)in, (float4*)out, n16);
    else if (kind == 3) hipLaunchKernelGGL((agg_lds_kernel<256>), dim3(blocks), dim3(256), 0, st, (float*)out, iters, 1);
    return (int)hipGetLastError();
}
