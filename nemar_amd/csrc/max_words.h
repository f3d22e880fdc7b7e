// Per-sample maxima published by a producer kernel (InstanceNorm, dropout): every workgroup STORES the maximum of what it wrote
// into its own partial word, a second tiny launch reduces a sample's partials to the word the fp16 x 3 convolutions scale by.
// (No atomics: device-scope atomicMax of thousands of workgroups on 8 addresses cost 25-90 us per launch — the words bounce between the
// 8 XCDs' L2s; measured in profiles/r3_producer_max_atomics.txt.)
// Buffer layout (NEMAR_MAX_WORDS(samples) 4-byte words, no initialisation needed):  [samples results | samples x partials scratch].
#pragma once
#include "common.h"

#define NEMAR_MAX_PARTIALS 2048
#define NEMAR_MAX_WORDS(samples) ((size_t)(samples) * (1 + NEMAR_MAX_PARTIALS))

// LAZY words (round 6b, nemar_set_max_words_lazy): the producer does not launch the reduction; result word n holds the MARKER
// 0xFFFF0000 | partials instead (not the bit pattern of any finite maximum), and a consumer that understands it — the two InstanceNorm
// producers of the residual blocks: norm_planes.hip — reduces the sample's partial words itself (one load per lane + a wave maximum).  The
// second launch was 4 us on one stream but 12 us beside the side stream, 57 times per step, on the chain everything else waits for.
#define NEMAR_MAX_LAZY_MARK 0xFFFF0000u
bool nemar_max_words_lazy();          // core.hip: this thread's setting
namespace {
__global__ __launch_bounds__(256) void max_words_finalize_kernel(unsigned* __restrict__ w, int samples, int partials) {
    __shared__ unsigned red[4];
    const unsigned* src = w + samples + (size_t)blockIdx.x * partials;
    unsigned m = 0;
    for (int i = threadIdx.x; i < partials; i += 256) m = max(m, src[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) w[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}
inline void max_words_finalize(unsigned* w, int samples, int partials, hipStream_t st) {
    hipLaunchKernelGGL(max_words_finalize_kernel, dim3(samples), dim3(256), 0, st, w, samples, partials);
}
// the per-sample maximum behind word n of a NEMAR_MAX_WORDS(samples) buffer (or of a plain array of finalized words): wave-uniform;
// every lane of the calling wave takes part
__device__ __forceinline__ unsigned sample_max_word(const unsigned* w, int n, int samples) {
    const unsigned v = w[n];
    if ((v & 0xFFFF0000u) != NEMAR_MAX_LAZY_MARK) return v;
    const int partials = (int)(v & 0xFFFFu);
    const unsigned* const src = w + samples + (size_t)n * partials;
    unsigned m = 0;
    for (int i = (int)(threadIdx.x & 63); i < partials; i += 64) m = max(m, src[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    return m;
}
}  // namespace
