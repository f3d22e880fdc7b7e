// K3 (+K5 fused): InstanceNorm2d(affine=False, track_running_stats=False), forward and backward, with the
// following ReLU / LeakyReLU and the ResnetBlock residual add fused in.
//
// Replaces nn.InstanceNorm2d at reference models/networks.py:24 (norm_layer of ResnetGenerator :351,358,373,
// ResnetBlock :426,439, NLayerDiscriminator :584,592) and models/stn/layers.py:16 (STN ResnetBlocks, affine STN
// convs), the nn.ReLU(True)/nn.LeakyReLU(0.2, True) that follow them, and `x + self.conv_block(x)`
// (models/networks.py:445).
//   fwd:  y = [residual +] act((x - mean_hw) * rsqrt(var_hw(biased) + eps)),   stats[plane] = (mean, rstd)
//   bwd:  g = gy * act'(xhat);  gx = rstd * (g - mean(g) - xhat * mean(g * xhat))
// HBM-bound.  One workgroup per (n,c) plane; planes up to THREADS*PER elements are held in registers so x (and gy)
// are read from memory exactly once and the variance is the exact two-pass form.
#include "common.h"
#include "max_words.h"

namespace {

constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2;

__device__ __forceinline__ float act_f(float v, int act, float slope) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
    return v;
}
__device__ __forceinline__ float act_df(float xhat, int act, float slope) {
    if (act == ACT_RELU) return xhat > 0.f ? 1.f : 0.f;
    if (act == ACT_LRELU) return xhat > 0.f ? 1.f : slope;
    return 1.f;
}

// Per-sample maximum of what a kernel writes, for the consumers that scale by it (the fp16 x 3 convolutions, csrc/conv_split16*.hip):
// the producer has every output element in a register anyway, so the max pass over the tensor disappears.  One word per sample,
// zero on entry; a workgroup only touches the word when its own maximum would change it (16 words x ~256 workgroups each: no hot
// atomic).  Finite magnitudes only, as nemar_absmax_samples.
__device__ __forceinline__ unsigned finite_mag(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
    return u < 0x7f800000u ? u : 0u;
}
// the workgroup's maximum -> its own partial word (max_words.h)
// pps < 0 = LAZY words (max_words.h): no reduction launch follows; the first plane of a sample leaves the marker in the sample's result word
__device__ __forceinline__ void publish_max(unsigned m, unsigned* maxw, int pps, unsigned* red) {
    const int P = pps < 0 ? -pps : pps;
    unsigned* const word = maxw + gridDim.x / P + blockIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 1; i < nw; ++i) m = max(m, red[i]);
        *word = m;
        if (pps < 0 && blockIdx.x % P == 0) maxw[blockIdx.x / P] = NEMAR_MAX_LAZY_MARK | (unsigned)P;
    }
}

// PER > 0: register-cached plane (HW <= THREADS*PER).  PER == 0: streaming (shifted one-pass statistics).
template <int THREADS, int PER>
__global__ __launch_bounds__(THREADS) void instnorm_fwd_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ residual,
                                                               float* __restrict__ y, float* __restrict__ stats, int HW,
                                                               float eps, int act, float slope, unsigned* maxw, int pps) {
    __shared__ float red[16];
    unsigned omax = 0;                                    // max |y| of this plane (maxw != null)
    const size_t base = (size_t)blockIdx.x * HW;
    const float* xp = x + base;
    float* yp = y + base;
    const float* rp = residual ? residual + base : nullptr;
    const float inv = 1.f / (float)HW;
    float mean, rstd;
    if (PER > 0) {
        float v[PER > 0 ? PER : 1];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * THREADS;
            v[k] = i < HW ? xp[i] : 0.f;
            s += v[k];
        }
        mean = block_sum(s, red) * inv;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * THREADS;
            const float d = i < HW ? v[k] - mean : 0.f;
            q += d * d;
        }
        rstd = 1.f / sqrtf(block_sum(q, red) * inv + eps);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * THREADS;
            if (i < HW) {
                float o = act_f((v[k] - mean) * rstd, act, slope);
                if (rp) o += rp[i];
                yp[i] = o;
                omax = max(omax, finite_mag(o));
            }
        }
    } else {
        const float shift = xp[0];
        float s = 0.f, q = 0.f;
        for (int i = threadIdx.x; i < HW; i += THREADS) {
            const float d = xp[i] - shift;
            s += d;
            q += d * d;
        }
        const float ms = block_sum(s, red) * inv;
        const float var = fmaxf(block_sum(q, red) * inv - ms * ms, 0.f);
        mean = shift + ms;
        rstd = 1.f / sqrtf(var + eps);
        for (int i = threadIdx.x; i < HW; i += THREADS) {
            float o = act_f((xp[i] - mean) * rstd, act, slope);
            if (rp) o += rp[i];
            yp[i] = o;
            omax = max(omax, finite_mag(o));
        }
    }
    if (threadIdx.x == 0) {
        stats[2 * (size_t)blockIdx.x] = mean;
        stats[2 * (size_t)blockIdx.x + 1] = rstd;
    }
    if (maxw) publish_max(omax, maxw, pps, reinterpret_cast<unsigned*>(red));
}

// 16-byte form of the register-cached kernels (HW % 4 == 0, 16-byte aligned planes): a thread holds PER4 float4 — a quarter of the load /
// store instructions and address registers of the scalar form (whose <1024, 64> instance spilled 32 registers), unconditional loads
// from clamped addresses with the tail masked afterwards (a conditional load makes hipcc wait for every load on the spot).
typedef float f32x4n __attribute__((ext_vector_type(4)));

// FULL: HW == 4 THREADS PER4 exactly (no clamped addresses, no tail masks: one address register for the whole plane)
template <int THREADS, int PER4, bool FULL>
__global__ __launch_bounds__(THREADS) void instnorm_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ residual,
                                                                float* __restrict__ y, float* __restrict__ stats, int HW, float eps, int act,
                                                                float slope, unsigned* maxw, int pps) {
    __shared__ float red[16];
    unsigned omax = 0;
    const size_t base = (size_t)blockIdx.x * HW;
    const f32x4n* xp = reinterpret_cast<const f32x4n*>(x + base);
    f32x4n* yp = reinterpret_cast<f32x4n*>(y + base);
    const f32x4n* rp = residual ? reinterpret_cast<const f32x4n*>(residual + base) : nullptr;
    const int n4 = HW >> 2;
    const float inv = 1.f / (float)HW;
    f32x4n v[PER4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PER4; ++k) {
        const int j = threadIdx.x + k * THREADS;
        v[k] = xp[(FULL || j < n4) ? j : n4 - 1];
    }
#pragma unroll
    for (int k = 0; k < PER4; ++k) {
        if (!FULL && threadIdx.x + k * THREADS >= n4) v[k] = f32x4n{0.f, 0.f, 0.f, 0.f};
        s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
    const float mean = uniform_f(block_sum(s, red) * inv);        // (wave-uniform values live in scalar registers)
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < PER4; ++k) {
        if (FULL || threadIdx.x + k * THREADS < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mean; q += d * d; }
        }
    }
    const float rstd = uniform_f(1.f / sqrtf(block_sum(q, red) * inv + eps));
    // (the residual is fetched in groups of four float4 per thread: all PER4 at once would double the register footprint)
    constexpr int RG = PER4 >= 16 ? 1 : (PER4 < 4 ? PER4 : (PER4 % 4 == 0 ? 4 : (PER4 % 2 == 0 ? 2 : 1)));
    static_assert(PER4 % RG == 0, "residual groups must tile the float4s of a thread");
#pragma unroll
    for (int k0 = 0; k0 < PER4; k0 += RG) {
        f32x4n r[RG];
        if (rp) {
#pragma unroll
            for (int k = 0; k < RG; ++k) {
                const int j = threadIdx.x + (k0 + k) * THREADS;
                r[k] = rp[(FULL || j < n4) ? j : n4 - 1];
            }
        }
#pragma unroll
        for (int k = 0; k < RG; ++k) {
            const int j = threadIdx.x + (k0 + k) * THREADS;
            f32x4n o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = act_f((v[k0 + k][e] - mean) * rstd, act, slope);
                if (rp) o[e] += r[k][e];
            }
            if (FULL || j < n4) {
                yp[j] = o;
#pragma unroll
                for (int e = 0; e < 4; ++e) omax = max(omax, finite_mag(o[e]));
            }
        }
    }
    if (threadIdx.x == 0) {
        stats[2 * (size_t)blockIdx.x] = mean;
        stats[2 * (size_t)blockIdx.x + 1] = rstd;
    }
    if (maxw) publish_max(omax, maxw, pps, reinterpret_cast<unsigned*>(red));
}

template <int THREADS, int PER4>
__global__ __launch_bounds__(THREADS) void instnorm_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                const float* __restrict__ gy, float* __restrict__ gx, int HW, int act,
                                                                float slope, unsigned* maxw, int pps) {
    __shared__ float red[16];
    unsigned omax = 0;
    const size_t base = (size_t)blockIdx.x * HW;
    const f32x4n* xp = reinterpret_cast<const f32x4n*>(x + base);
    const f32x4n* gp = reinterpret_cast<const f32x4n*>(gy + base);
    f32x4n* op = reinterpret_cast<f32x4n*>(gx + base);
    const float mean = stats[2 * (size_t)blockIdx.x], rstd = stats[2 * (size_t)blockIdx.x + 1];
    const int n4 = HW >> 2;
    const float inv = 1.f / (float)HW;
    f32x4n xh[PER4], g[PER4];
#pragma unroll
    for (int k = 0; k < PER4; ++k) {
        const int j = threadIdx.x + k * THREADS, jc = j < n4 ? j : n4 - 1;
        xh[k] = xp[jc];
        g[k] = gp[jc];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER4; ++k) {
        const bool on = threadIdx.x + k * THREADS < n4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h = on ? (xh[k][e] - mean) * rstd : 0.f;
            const float t = on ? g[k][e] * act_df(h, act, slope) : 0.f;
            xh[k][e] = h;
            g[k][e] = t;
            s1 += t;
            s2 += t * h;
        }
    }
    const float m1 = uniform_f(block_sum(s1, red) * inv);
    const float m2 = uniform_f(block_sum(s2, red) * inv);
#pragma unroll
    for (int k = 0; k < PER4; ++k) {
        const int j = threadIdx.x + k * THREADS;
        f32x4n o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (g[k][e] - m1 - xh[k][e] * m2);
        if (j < n4) {
            op[j] = o;
#pragma unroll
            for (int e = 0; e < 4; ++e) omax = max(omax, finite_mag(o[e]));
        }
    }
    if (maxw) publish_max(omax, maxw, pps, reinterpret_cast<unsigned*>(red));
}

// HW == 4 THREADS PER4 exactly, too large for x AND gy in registers (256 x 256 planes: the registration net's 32-channel layers, the
// translation net's stem): gy stays in registers, x is read twice — the second time from the memory-side cache —
// with four 16-byte loads in flight per thread (g is recomputed in the second pass: gy is never rewritten in its registers).  Four tensor passes of 16-byte accesses instead of the streaming kernel's five passes of
// 4-byte ones (130 us per call on 256 planes of 256 x 256: 1.5 TB/s).
template <int THREADS, int PER4>
__global__ __launch_bounds__(THREADS) void instnorm_bwd4s_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                 const float* __restrict__ gy, float* __restrict__ gx, int HW, int act,
                                                                 float slope, unsigned* maxw, int pps) {
    static_assert(PER4 % 4 == 0, "x is fetched four float4 at a time");
    __shared__ float red[16];
    unsigned omax = 0;
    const size_t base = (size_t)blockIdx.x * HW;
    const f32x4n* xp = reinterpret_cast<const f32x4n*>(x + base);
    const f32x4n* gp = reinterpret_cast<const f32x4n*>(gy + base);
    f32x4n* op = reinterpret_cast<f32x4n*>(gx + base);
    const float mean = stats[2 * (size_t)blockIdx.x], rstd = stats[2 * (size_t)blockIdx.x + 1];
    const float inv = 1.f / (float)HW;
    const float dneg = act == ACT_RELU ? 0.f : (act == ACT_LRELU ? slope : 1.f);       // act'(xhat <= 0), chosen once (act_df per element: a branch each)
    f32x4n g[PER4];
#pragma unroll
    for (int k = 0; k < PER4; ++k) g[k] = gp[threadIdx.x + k * THREADS];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < PER4; k0 += 4) {
        f32x4n xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = xp[threadIdx.x + (k0 + k) * THREADS];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (xv[k][e] - mean) * rstd;
                const float t = g[k0 + k][e] * (h > 0.f ? 1.f : dneg);
                s1 += t;
                s2 += t * h;
            }
            // one float4 at a time: hipcc otherwise hoists all sixteen x loads above the arithmetic, or runs the whole s1 chain before
            // the s2 chain with every xhat alive in between (142 / 39 spilled registers)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifndef NEMAR_HOST_EMULATION
    asm volatile("" : "+v"(s1), "+v"(s2));                 // both sums complete HERE (hipcc otherwise sinks the s2 chain below the first reduction,
#endif                                                     // with every product's operands spilled to scratch on the way)
    const float m1 = uniform_f(block_sum(s1, red) * inv);
    const float m2 = uniform_f(block_sum(s2, red) * inv);
    // the second read of x must BE a read: through an opaque copy of the pointer (hipcc otherwise keeps the first pass's 64 values per
    // thread alive across the reduction — 144 spilled registers)
    const f32x4n* xp2 = xp;
#ifndef NEMAR_HOST_EMULATION
    asm volatile("" : "+s"(xp2));
#endif
#pragma unroll
    for (int k0 = 0; k0 < PER4; k0 += 4) {
        f32x4n xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = xp2[threadIdx.x + (k0 + k) * THREADS];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4n o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (xv[k][e] - mean) * rstd;
                const float t = g[k0 + k][e] * (h > 0.f ? 1.f : dneg);
                o[e] = rstd * (t - m1 - h * m2);
                omax = max(omax, finite_mag(o[e]));
            }
            op[threadIdx.x + (k0 + k) * THREADS] = o;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (maxw) publish_max(omax, maxw, pps, reinterpret_cast<unsigned*>(red));
}

template <int THREADS, int PER>
__global__ __launch_bounds__(THREADS) void instnorm_bwd_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ gy, float* __restrict__ gx,
                                                               int HW, int act, float slope, unsigned* maxw, int pps) {
    __shared__ float red[16];
    unsigned omax = 0;
    const size_t base = (size_t)blockIdx.x * HW;
    const float* xp = x + base;
    const float* gp = gy + base;
    float* op = gx + base;
    const float mean = stats[2 * (size_t)blockIdx.x], rstd = stats[2 * (size_t)blockIdx.x + 1];
    const float inv = 1.f / (float)HW;
    if (PER > 0) {
        float xh[PER > 0 ? PER : 1], g[PER > 0 ? PER : 1];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * THREADS;
            if (i < HW) {
                xh[k] = (xp[i] - mean) * rstd;
                g[k] = gp[i] * act_df(xh[k], act, slope);
            } else {
                xh[k] = 0.f;
                g[k] = 0.f;
            }
            s1 += g[k];
            s2 += g[k] * xh[k];
        }
        const float m1 = block_sum(s1, red) * inv;
        const float m2 = block_sum(s2, red) * inv;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * THREADS;
            if (i < HW) {
                const float o = rstd * (g[k] - m1 - xh[k] * m2);
                op[i] = o;
                omax = max(omax, finite_mag(o));
            }
        }
    } else {
        float s1 = 0.f, s2 = 0.f;
        for (int i = threadIdx.x; i < HW; i += THREADS) {
            const float xh = (xp[i] - mean) * rstd;
            const float g = gp[i] * act_df(xh, act, slope);
            s1 += g;
            s2 += g * xh;
        }
        const float m1 = block_sum(s1, red) * inv;
        const float m2 = block_sum(s2, red) * inv;
        for (int i = threadIdx.x; i < HW; i += THREADS) {
            const float xh = (xp[i] - mean) * rstd;
            const float g = gp[i] * act_df(xh, act, slope);
            const float o = rstd * (g - m1 - xh * m2);
            op[i] = o;
            omax = max(omax, finite_mag(o));
        }
    }
    if (maxw) publish_max(omax, maxw, pps, reinterpret_cast<unsigned*>(red));
}

}  // namespace

// x, y, residual (nullable): [planes, HW] with planes = N*C;  stats: [planes, 2] = (mean, rstd)
static int instnorm_fwd_impl(const float* x, const float* residual, float* y, float* stats, int planes, int HW, float eps, int act,
                             float slope, unsigned* maxw, int pps, void* stream);

NEMAR_API int nemar_instnorm_fwd(const float* x, const float* residual, float* y, float* stats, int planes, int HW,
                                 float eps, int act, float slope, void* stream) {
    return instnorm_fwd_impl(x, residual, y, stats, planes, HW, eps, act, slope, nullptr, 1, stream);
}

// ... and max |y| per sample into max_words[plane / planes_per_sample] (words zero on entry; see publish_max)
NEMAR_API int nemar_instnorm_fwd_max(const float* x, const float* residual, float* y, float* stats, int planes, int HW,
                                     float eps, int act, float slope, void* max_words, int planes_per_sample, void* stream) {
    NEMAR_REQUIRE(max_words && planes_per_sample > 0 && planes_per_sample <= NEMAR_MAX_PARTIALS && planes % planes_per_sample == 0,
                  "instnorm_fwd_max: bad max words");
    return instnorm_fwd_impl(x, residual, y, stats, planes, HW, eps, act, slope, (unsigned*)max_words, planes_per_sample, stream);
}

static int instnorm_fwd_impl(const float* x, const float* residual, float* y, float* stats, int planes, int HW, float eps, int act,
                             float slope, unsigned* maxw, int pps, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && y && stats, "instnorm_fwd: null pointer");
    NEMAR_REQUIRE(planes > 0 && HW > 0, "instnorm_fwd: bad shape planes=%d HW=%d", planes, HW);
    NEMAR_REQUIRE(act == ACT_NONE || act == ACT_RELU || act == ACT_LRELU, "instnorm_fwd: unsupported act %d", act);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(planes);
    // LAZY words (nemar_set_max_words_lazy): no reduction launch — the kernel leaves the marker and whoever needs the word reduces the
    // partials (the InstanceNorm producers of norm_planes.hip in their prologue, anyone else through nemar_max_words_finalize)
    const bool lazy = maxw && nemar_max_words_lazy() && pps <= 0xFFFF;
    const int pps_k = lazy ? -pps : pps;
    const bool vec = HW % 4 == 0 && HW >= 1024 && HW <= 1024 * 64 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
    if (vec && HW <= 256 * 16)
        hipLaunchKernelGGL((instnorm_fwd4_kernel<256, 4, false>), grid, dim3(256), 0, st, x, residual, y, stats, HW, eps, act, slope, maxw, pps_k);
    else if (vec && HW <= 1024 * 16)
        hipLaunchKernelGGL((instnorm_fwd4_kernel<1024, 4, false>), grid, dim3(1024), 0, st, x, residual, y, stats, HW, eps, act, slope, maxw, pps_k);
    else if (vec && HW <= 1024 * 32)
        hipLaunchKernelGGL((instnorm_fwd4_kernel<1024, 8, false>), grid, dim3(1024), 0, st, x, residual, y, stats, HW, eps, act, slope, maxw, pps_k);
    else if (vec && HW == 1024 * 64)
        hipLaunchKernelGGL((instnorm_fwd4_kernel<1024, 16, true>), grid, dim3(1024), 0, st, x, residual, y, stats, HW, eps, act, slope, maxw, pps_k);
    else if (vec && HW <= 1024 * 48)        // 32768 < HW <= 49152 (200 x 200, 208 x 208 ...): register-cached as well
        hipLaunchKernelGGL((instnorm_fwd4_kernel<1024, 12, false>), grid, dim3(1024), 0, st, x, residual, y, stats, HW, eps, act, slope, maxw, pps_k);
    else if (vec && HW <= 1024 * 56)        // ... <= 57344 (224 x 224, 232 x 232); a <1024, 16, false> instance spills 20 registers
        hipLaunchKernelGGL((instnorm_fwd4_kernel<1024, 14, false>), grid, dim3(1024), 0, st, x, residual, y, stats, HW, eps, act, slope, maxw, pps_k);
    else if (HW <= 64 * 8)
        hipLaunchKernelGGL((instnorm_fwd_kernel<64, 8>), grid, dim3(64), 0, st, x, residual, y, stats, HW, eps, act, slope, maxw, pps_k);
    else if (HW <= 256 * 16)
        hipLaunchKernelGGL((instnorm_fwd_kernel<256, 16>), grid, dim3(256), 0, st, x, residual, y, stats, HW, eps, act, slope, maxw, pps_k);
    else if (HW <= 1024 * 16)
        hipLaunchKernelGGL((instnorm_fwd_kernel<1024, 16>), grid, dim3(1024), 0, st, x, residual, y, stats, HW, eps, act, slope, maxw, pps_k);
    else
        hipLaunchKernelGGL((instnorm_fwd_kernel<1024, 0>), grid, dim3(1024), 0, st, x, residual, y, stats, HW, eps, act, slope, maxw, pps_k);
    if (maxw && !lazy) max_words_finalize(maxw, planes / pps, pps, st);
    NEMAR_CHECK_LAUNCH("instnorm_fwd");
    return NEMAR_OK;
}

static int instnorm_bwd_impl(const float* x, const float* stats, const float* gy, float* gx, int planes, int HW, int act, float slope,
                             unsigned* maxw, int pps, void* stream);

NEMAR_API int nemar_instnorm_bwd(const float* x, const float* stats, const float* gy, float* gx, int planes, int HW,
                                 int act, float slope, void* stream) {
    return instnorm_bwd_impl(x, stats, gy, gx, planes, HW, act, slope, nullptr, 1, stream);
}

NEMAR_API int nemar_instnorm_bwd_max(const float* x, const float* stats, const float* gy, float* gx, int planes, int HW,
                                     int act, float slope, void* max_words, int planes_per_sample, void* stream) {
    NEMAR_REQUIRE(max_words && planes_per_sample > 0 && planes_per_sample <= NEMAR_MAX_PARTIALS && planes % planes_per_sample == 0,
                  "instnorm_bwd_max: bad max words");
    return instnorm_bwd_impl(x, stats, gy, gx, planes, HW, act, slope, (unsigned*)max_words, planes_per_sample, stream);
}

static int instnorm_bwd_impl(const float* x, const float* stats, const float* gy, float* gx, int planes, int HW, int act, float slope,
                             unsigned* maxw, int pps, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && stats && gy && gx, "instnorm_bwd: null pointer");
    NEMAR_REQUIRE(planes > 0 && HW > 0, "instnorm_bwd: bad shape planes=%d HW=%d", planes, HW);
    NEMAR_REQUIRE(act == ACT_NONE || act == ACT_RELU || act == ACT_LRELU, "instnorm_bwd: unsupported act %d", act);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(planes);
    // LAZY words (nemar_set_max_words_lazy): no reduction launch — the kernel leaves the marker and whoever needs the word reduces the
    // partials (the InstanceNorm producers of norm_planes.hip in their prologue, anyone else through nemar_max_words_finalize)
    const bool lazy = maxw && nemar_max_words_lazy() && pps <= 0xFFFF;
    const int pps_k = lazy ? -pps : pps;
    const bool vec = HW % 4 == 0 && HW >= 1024 && HW <= 1024 * 32 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx)) & 15) == 0;
    if (vec && HW <= 256 * 16)
        hipLaunchKernelGGL((instnorm_bwd4_kernel<256, 4>), grid, dim3(256), 0, st, x, stats, gy, gx, HW, act, slope, maxw, pps_k);
    else if (vec && HW <= 1024 * 16)
        hipLaunchKernelGGL((instnorm_bwd4_kernel<1024, 4>), grid, dim3(1024), 0, st, x, stats, gy, gx, HW, act, slope, maxw, pps_k);
    else if (vec)
        hipLaunchKernelGGL((instnorm_bwd4_kernel<1024, 8>), grid, dim3(1024), 0, st, x, stats, gy, gx, HW, act, slope, maxw, pps_k);
    else if (HW == 1024 * 64 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx)) & 15) == 0)
        hipLaunchKernelGGL((instnorm_bwd4s_kernel<1024, 16>), grid, dim3(1024), 0, st, x, stats, gy, gx, HW, act, slope, maxw, pps_k);
    else if (HW <= 64 * 8)
        hipLaunchKernelGGL((instnorm_bwd_kernel<64, 8>), grid, dim3(64), 0, st, x, stats, gy, gx, HW, act, slope, maxw, pps_k);
    else if (HW <= 256 * 16)
        hipLaunchKernelGGL((instnorm_bwd_kernel<256, 16>), grid, dim3(256), 0, st, x, stats, gy, gx, HW, act, slope, maxw, pps_k);
    else if (HW <= 1024 * 16)
        hipLaunchKernelGGL((instnorm_bwd_kernel<1024, 16>), grid, dim3(1024), 0, st, x, stats, gy, gx, HW, act, slope, maxw, pps_k);
    else if (HW <= 1024 * 32)
        hipLaunchKernelGGL((instnorm_bwd_kernel<1024, 32>), grid, dim3(1024), 0, st, x, stats, gy, gx, HW, act, slope, maxw, pps_k);
    else
        hipLaunchKernelGGL((instnorm_bwd_kernel<1024, 0>), grid, dim3(1024), 0, st, x, stats, gy, gx, HW, act, slope, maxw, pps_k);
    if (maxw && !lazy) max_words_finalize(maxw, planes / pps, pps, st);
    NEMAR_CHECK_LAUNCH("instnorm_bwd");
    return NEMAR_OK;
}

// The on-demand form of the reduction a lazy producer skipped (nemar_set_max_words_lazy): for every sample whose result word still holds the
// marker, reduce its partial words into it; words that are maxima already are left alone.  One launch, only when somebody asks.
namespace {
__global__ __launch_bounds__(256) void max_words_finalize_lazy_kernel(unsigned* __restrict__ w, int samples) {
    __shared__ unsigned red[4];
    const unsigned v = w[blockIdx.x];
    if ((v & 0xFFFF0000u) != NEMAR_MAX_LAZY_MARK) return;          // (uniform per workgroup)
    const int partials = (int)(v & 0xFFFFu);
    const unsigned* src = w + samples + (size_t)blockIdx.x * partials;
    unsigned m = 0;
    for (int i = threadIdx.x; i < partials; i += 256) m = max(m, src[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) w[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}
}  // namespace

NEMAR_API int nemar_max_words_finalize(void* max_words, int samples, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(max_words && samples > 0 && samples <= 65535, "max_words_finalize: bad arguments");
    hipLaunchKernelGGL(max_words_finalize_lazy_kernel, dim3(samples), dim3(256), 0, (hipStream_t)stream, (unsigned*)max_words, samples);
    NEMAR_CHECK_LAUNCH("max_words_finalize");
    return NEMAR_OK;
}
