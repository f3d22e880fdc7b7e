// Fixed-order second stage of every split reduction in the backward pass (weight gradients, split data gradients, bias
// gradients, the narrow kernels' channel / tile splits): the first-stage kernels store their partial results to per-split
// slabs with plain stores, and this kernel adds the slabs in split order — so the backward pass is bitwise reproducible
// run to run, which fp32 atomics (the round-1 design) are not.  The reference's CPU path is bitwise repeatable
// (SURVEY.md §8c); autograd's AccumulateGrad (reference models/nemar_model.py:223,260 loss.backward()) is the `+=` below.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// dst[i] = (accumulate ? dst[i] : 0) + sum_{s < splits} part[s * stride + i], with a FIXED association order:
// lane y of a column adds slabs y, y+16, y+32, ... in ascending order, then the 16 lane sums are added in ascending y, then
// dst.  Workgroup = 16 x 16 threads: x = 16 consecutive float4 (one 256-byte run per slab row: coalesced), y = slab lane —
// so a reduction over hundreds of small slabs (narrow layers: 341 slabs of 9216 floats) is as parallel as one over a few
// large ones (256->256 3x3: 14 slabs of 590k floats), instead of one thread walking all slabs at memory latency.
// (bx, gx: this job's block index and block count — blockIdx.x / gridDim.x for the single-job kernels, a sub-range of the grid for the
// two-job kernel below)
template <bool VEC>
__device__ __forceinline__ void sum_general_body(f32x4 (*red)[17], const float* __restrict__ part, long long stride, int splits,
                                                 float* __restrict__ dst, long long n, int accumulate, int act, float slope,
                                                 unsigned bx, unsigned gx) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const long long cols = VEC ? (n >> 2) : n;            // columns of 4 floats (VEC) or 1 float
    for (long long c0 = (long long)bx * 16; c0 < cols; c0 += (long long)gx * 16) {
        const long long c = c0 + tx;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (c < cols) {
            for (int s = ty; s < splits; s += 16) {
                const float* q = part + (long long)s * stride;
                if (VEC) acc += *reinterpret_cast<const f32x4*>(q + 4 * c);
                else acc[0] += q[c];
            }
        }
        red[ty][tx] = acc;
        __syncthreads();
        if (ty == 0 && c < cols) {
            f32x4 t = red[0][tx];
#pragma unroll
            for (int y = 1; y < 16; ++y) t += red[y][tx];
            if (VEC) {
                f32x4* d = reinterpret_cast<f32x4*>(dst + 4 * c);
                if (accumulate) t += *d;
                if (act) {                                     // 1 ReLU, 2 LeakyReLU (a split FORWARD convolution: the activation follows the sum)
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = t[e] > 0.f ? t[e] : (act == 1 ? 0.f : t[e] * slope);
                }
                *d = t;
            } else {
                if (accumulate) t[0] += dst[c];
                if (act) t[0] = t[0] > 0.f ? t[0] : (act == 1 ? 0.f : t[0] * slope);
                dst[c] = t[0];
            }
        }
        __syncthreads();
    }
}
template <bool VEC>
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, long long stride, int splits,
                                                           float* __restrict__ dst, long long n, int accumulate, int act,
                                                           float slope) {
    __shared__ f32x4 red[16][17];
    sum_general_body<VEC>(red, part, stride, splits, dst, n, accumulate, act, slope, blockIdx.x, gridDim.x);
}

// Few slabs (<= 16: the 16-slab sums of the wide weight gradient — 18 per step, 37.7 MB each — and most bias / split sums), 16-byte
// columns: one thread = one column, all slabs' words in flight at once, no LDS, no barrier.  SAME association order as the general
// kernel below ((0 + s0) + s1) + ... [+ dst]: bit-identical results.  (Round 6: in the general kernel a thread had ONE 16-byte load in
// flight per 16 columns and 240 of 256 threads idled through the second phase: 48 us = 0.8 TB/s for the wide layers' sums.)
__device__ __forceinline__ void sum_few_body(const float* __restrict__ part, long long stride, int splits, float* __restrict__ dst, long long n,
                                             int accumulate, int act, float slope, unsigned bx, unsigned gx) {
    const long long cols = n >> 2;
    for (long long c = (long long)bx * 256 + threadIdx.x; c < cols; c += (long long)gx * 256) {
        f32x4 v[16];
#pragma unroll
        for (int s = 0; s < 16; ++s)                       // unconditional loads (slots beyond the last slab re-read slab 0)
            v[s] = *reinterpret_cast<const f32x4*>(part + (long long)(s < splits ? s : 0) * stride + 4 * c);
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        t += v[0];
#pragma unroll
        for (int s = 1; s < 16; ++s)
            if (s < splits) t += v[s];
        f32x4* d = reinterpret_cast<f32x4*>(dst + 4 * c);
        if (accumulate) t += *d;
        if (act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = t[e] > 0.f ? t[e] : (act == 1 ? 0.f : t[e] * slope);
        }
        *d = t;
    }
}
__global__ __launch_bounds__(256) void sum_partials_few_kernel(const float* __restrict__ part, long long stride, int splits,
                                                               float* __restrict__ dst, long long n, int accumulate, int act, float slope) {
    sum_few_body(part, stride, splits, dst, n, accumulate, act, slope, blockIdx.x, gridDim.x);
}

// Two reductions in ONE launch (a weight gradient's slabs and its bias gradient's: every weight-gradient call ended with two sum launches):
// blocks [0, a.blocks) run job a, the rest job b, each with the body — and therefore the association order, bit for bit — its own launch
// would have used (kind 0: few-slab form, 1: general 16-byte form, 2: general scalar form).
struct SumJob {
    const float* part; long long stride; int splits; float* dst; long long n; int accumulate; int kind; unsigned blocks;
};
__global__ __launch_bounds__(256) void sum_partials_pair_kernel(SumJob a, SumJob b) {
    __shared__ f32x4 red[16][17];
    const bool first = blockIdx.x < a.blocks;
    const SumJob& j = first ? a : b;
    const unsigned bx = first ? blockIdx.x : blockIdx.x - a.blocks;
    if (j.kind == 0) sum_few_body(j.part, j.stride, j.splits, j.dst, j.n, j.accumulate, 0, 0.f, bx, j.blocks);
    else if (j.kind == 1) sum_general_body<true>(red, j.part, j.stride, j.splits, j.dst, j.n, j.accumulate, 0, 0.f, bx, j.blocks);
    else sum_general_body<false>(red, j.part, j.stride, j.splits, j.dst, j.n, j.accumulate, 0, 0.f, bx, j.blocks);
}

// the slabs hold [N][C0 + C1][HW]; channels < C0 go to d0 [N][C0][HW], the rest to d1 [N][C1][HW] (data gradient of a two-source layer);
// slabs added in ascending order
__global__ __launch_bounds__(256) void sum_partials_two_kernel(const float* __restrict__ part, long long stride, int splits, float* __restrict__ d0,
                                                               float* __restrict__ d1, int C0, int C1, int HW, long long total) {
    const int C = C0 + C1;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long n = idx / ((long long)C * HW);
        const int r = (int)(idx - n * C * HW), c = r / HW, px = r - c * HW;
        float t = 0.f;
        for (int sidx = 0; sidx < splits; ++sidx) t += part[(long long)sidx * stride + idx];
        if (c < C0) d0[((size_t)n * C0 + c) * HW + px] = t;
        else d1[((size_t)n * C1 + (c - C0)) * HW + px] = t;
    }
}

// Second stage of a reduction-split stride-1 REFLECT data gradient computed on the padded domain (conv.hip fold_small): the slabs hold
// [planes][H + 2 pad][W + 2 pad]; gx[plane][h][w] = the slab sums (ascending slab order, ((0 + s0) + s1) + ...) of every padded position that
// mirrors onto (h, w), added in reflect_fold_kernel's order (rows: own, upper mirror, lower mirror; within a row: own, left, right) [+ addend].
// Workgroup = 16 consecutive texels x 16 slab lanes: lane y LOADS slab s0 + y's values of the texel's (up to nine) positions — every load of
// a round independent of the others —, lane 0 ADDS them in the order above.  (One thread per texel walking its slabs and positions in turn
// cost 11.4 us per launch on maps of a few hundred texels: dependent rounds of cache misses; this form 5.6 us, the same bits.  Letting each
// lane sum its own slabs first is as fast but another association order — and the ten-step trajectory test sits close enough to its
// tolerance at steps 7 - 8 to notice.)
__global__ __launch_bounds__(256) void sum_partials_fold_kernel(const float* __restrict__ part, long long stride, int splits, float* __restrict__ gx,
                                                                int H, int W, int pad, long long total, const float* __restrict__ addend) {
    __shared__ float val[16][9][17];
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    for (long long idx0 = (long long)blockIdx.x * 16; idx0 < total; idx0 += (long long)gridDim.x * 16) {
        const long long idx = idx0 + tx;
        const bool live = idx < total;
        const long long cidx = live ? idx : total - 1;
        const int w = (int)(cidx % W);
        const long long t = cidx / W;
        const int h = (int)(t % H);
        const long long nc = t / H;
        const int y0 = h + pad, x0 = w + pad;
        const int y1 = (h >= 1 && h <= pad) ? pad - h : -1, y2 = (h <= H - 2 && h >= H - 1 - pad) ? 2 * (H - 1) - h + pad : -1;
        const int x1 = (w >= 1 && w <= pad) ? pad - w : -1, x2 = (w <= W - 2 && w >= W - 1 - pad) ? 2 * (W - 1) - w + pad : -1;
        // position k = 3 row + column; absent mirrors re-read the own row / column (their values are never added)
        const int ya = y1 >= 0 ? y1 : y0, yb = y2 >= 0 ? y2 : y0, xa = x1 >= 0 ? x1 : x0, xb = x2 >= 0 ? x2 : x0;
        const long long pbase = nc * (long long)Hp * Wp;
        float a00 = 0.f, a01 = 0.f, a02 = 0.f, a10 = 0.f, a11 = 0.f, a12 = 0.f, a20 = 0.f, a21 = 0.f, a22 = 0.f;      // (lane 0: the slab sums per position)
        for (int s0 = 0; s0 < splits; s0 += 16) {
            const int sl = s0 + ty;
            if (sl < splits) {
                const float* q = part + (long long)sl * stride + pbase;
                const float* r0 = q + (long long)y0 * Wp;
                const float* r1 = q + (long long)ya * Wp;
                const float* r2 = q + (long long)yb * Wp;
                val[ty][0][tx] = r0[x0]; val[ty][1][tx] = r0[xa]; val[ty][2][tx] = r0[xb];
                val[ty][3][tx] = r1[x0]; val[ty][4][tx] = r1[xa]; val[ty][5][tx] = r1[xb];
                val[ty][6][tx] = r2[x0]; val[ty][7][tx] = r2[xa]; val[ty][8][tx] = r2[xb];
            }
            __syncthreads();
            if (ty == 0) {
                const int n = splits - s0 < 16 ? splits - s0 : 16;
                for (int y = 0; y < n; ++y) {
                    a00 += val[y][0][tx]; a01 += val[y][1][tx]; a02 += val[y][2][tx];
                    a10 += val[y][3][tx]; a11 += val[y][4][tx]; a12 += val[y][5][tx];
                    a20 += val[y][6][tx]; a21 += val[y][7][tx]; a22 += val[y][8][tx];
                }
            }
            __syncthreads();
        }
        if (ty == 0 && live) {
            auto rowsum = [&](float v, float l, float r) {
                if (x1 >= 0) v += l;
                if (x2 >= 0) v += r;
                return v;
            };
            float sum = rowsum(a00, a01, a02);
            if (y1 >= 0) sum += rowsum(a10, a11, a12);
            if (y2 >= 0) sum += rowsum(a20, a21, a22);
            gx[idx] = addend ? sum + addend[idx] : sum;          // (+ the skip gradient of the ResnetBlock this convolution opens)
        }
    }
}

}  // namespace

void nemar_sum_partials_fold(const float* part, long long stride, int splits, float* gx, long long planes, int H, int W, int pad,
                             const float* addend, hipStream_t st) {
    const long long total = planes * H * W;
    long long blocks = (total + 15) / 16;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sum_partials_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, stride, splits, gx, H, W, pad, total, addend);
}

void nemar_sum_partials_two(const float* part, long long stride, int splits, float* d0, float* d1, int N, int C0, int C1, int HW, hipStream_t st) {
    const long long total = (long long)N * (C0 + C1) * HW;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum_partials_two_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, stride, splits, d0, d1, C0, C1, HW, total);
}

// form and grid of one reduction (what sum_partials_launch picks)
static SumJob sum_job(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate) {
    const bool vec = (n & 3) == 0 && (stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const long long cols = vec ? (n >> 2) : n;
    long long blocks = (cols + 15) / 16;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    SumJob j{part, stride, splits, dst, n, accumulate ? 1 : 0, vec ? 1 : 2, (unsigned)blocks};
    if (vec && splits >= 1 && splits <= 16) {
        long long fb = (cols + 255) / 256;
        if (fb > 2048) fb = 2048;
        if (fb < 1) fb = 1;
        j.kind = 0; j.blocks = (unsigned)fb;
    }
    return j;
}

static void sum_partials_launch(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate, int act, float slope,
                                hipStream_t st) {
    const bool vec = (n & 3) == 0 && (stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const long long cols = vec ? (n >> 2) : n;
    long long blocks = (cols + 15) / 16;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    if (vec && splits >= 1 && splits <= 16) {
        long long fb = (cols + 255) / 256;
        if (fb > 2048) fb = 2048;
        hipLaunchKernelGGL(sum_partials_few_kernel, dim3((unsigned)fb), dim3(256), 0, st, part, stride, splits, dst, n, accumulate ? 1 : 0, act,
                           slope);
    } else if (vec)
        hipLaunchKernelGGL((sum_partials_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, part, stride, splits, dst, n,
                           accumulate ? 1 : 0, act, slope);
    else
        hipLaunchKernelGGL((sum_partials_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, st, part, stride, splits, dst, n,
                           accumulate ? 1 : 0, act, slope);
}

void nemar_sum_partials(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate, hipStream_t st) {
    sum_partials_launch(part, stride, splits, dst, n, accumulate, 0, 0.f, st);
}

// the two reductions that end a weight-gradient call (weights: splits_a slabs of n_a floats; bias: splits_b of n_b) in one launch; b's part
// NULL = the first one alone
void nemar_sum_partials_pair(const float* part_a, long long stride_a, int splits_a, float* dst_a, long long n_a,
                             const float* part_b, long long stride_b, int splits_b, float* dst_b, long long n_b, bool accumulate, hipStream_t st) {
    if (!part_b || !dst_b) { nemar_sum_partials(part_a, stride_a, splits_a, dst_a, n_a, accumulate, st); return; }
    const SumJob a = sum_job(part_a, stride_a, splits_a, dst_a, n_a, accumulate), b = sum_job(part_b, stride_b, splits_b, dst_b, n_b, accumulate);
    hipLaunchKernelGGL(sum_partials_pair_kernel, dim3(a.blocks + b.blocks), dim3(256), 0, st, a, b);
}

// dst = act(sum of the slabs): the second stage of a reduction-split FORWARD convolution (slab 0 carries the bias); act 1 ReLU, 2 LeakyReLU
void nemar_sum_partials_act(const float* part, long long stride, int splits, float* dst, long long n, int act, float slope, hipStream_t st) {
    sum_partials_launch(part, stride, splits, dst, n, false, act, slope, st);
}
