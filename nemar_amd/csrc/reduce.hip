// Fixed-order second stage of every split reduction in the backward pass (weight gradients, split data gradients, bias
// gradients, the narrow kernels' channel / tile splits): the first-stage kernels store their partial results to per-split
// slabs with plain stores, and this kernel adds the slabs in split order — so the backward pass is bitwise reproducible
// run to run, which fp32 atomics (the round-1 design) are not.  The reference's CPU path is bitwise repeatable
// (SURVEY.md §8c); autograd's AccumulateGrad (reference models/nemar_model.py:223,260 loss.backward()) is the `+=` below.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// dst[i] = (accumulate ? dst[i] : 0) + bias[i / bias_div] + sum_{s < splits} part[s * stride + i]   (s ascending)
template <bool VEC>
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, long long stride, int splits,
                                                           float* __restrict__ dst, long long n, int accumulate) {
    const long long step = (long long)gridDim.x * blockDim.x;
    if (VEC) {
        const long long n4 = n >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(part + 4 * i);
            for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(part + (long long)s * stride + 4 * i);
            f32x4* d = reinterpret_cast<f32x4*>(dst + 4 * i);
            if (accumulate) acc += *d;
            *d = acc;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
            float acc = part[i];
            for (int s = 1; s < splits; ++s) acc += part[(long long)s * stride + i];
            if (accumulate) acc += dst[i];
            dst[i] = acc;
        }
    }
}

}  // namespace

void nemar_sum_partials(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate,
                        hipStream_t st) {
    const bool vec = (n & 3) == 0 && (stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL((sum_partials_kernel<true>), dim3(nemar_stream_grid(n >> 2, 256)), dim3(256), 0, st, part, stride,
                           splits, dst, n, accumulate ? 1 : 0);
    else
        hipLaunchKernelGGL((sum_partials_kernel<false>), dim3(nemar_stream_grid(n, 256)), dim3(256), 0, st, part, stride,
                           splits, dst, n, accumulate ? 1 : 0);
}
