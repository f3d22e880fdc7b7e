// nemar_amd — InstanceNorm2d (+ ReLU / LeakyReLU, + Dropout, + the ResnetBlock skip) that ALSO emits its output in the form the next
// 3x3 reflect-padded convolution of the fp16 x 3 route consumes: the padded, channel-blocked hi / lo planes of conv_split16.hip
// (plane[t][n][c/8][row 0..H+3][slot 0..W+3][8 channels], mirrored border materialised).  Reference: the
// conv -> InstanceNorm -> ReLU -> [Dropout] -> ReflectionPad -> conv chain of ResnetBlock, models/networks.py:418-446.
//
// Why this is possible in ONE pass: the split needs a power-of-two scale that keeps the largest magnitude of the sample inside fp16's
// range, and a separate max pass (or the producer's max words) only exists AFTER the producer has finished.  But the scale does not
// have to come from the actual maximum: with h = RN16(v s), l = RN16(v s - h) the pair (h, l) carries 22 significant bits of v as long
// as l is a normal fp16 number, i.e. for |v s| >= 2^-3, and has an absolute error <= 2^-25 below that.  A scale derived from an a-priori
// BOUND B >= max |v| (B = 2^k x the actual maximum) puts the maximum at 2^(11-k) .. 2^(12-k) instead of 2^11 .. 2^12: every element
// larger than 2^(k-14) x max keeps its 22 bits, smaller ones are off by <= 2^(k-36) x max — for k <= 8 far below the 2^-22 of the split
// itself.  And InstanceNorm has a bound: |xhat| <= sqrt(HW - 1) for every element of a plane (Samuelson), so
//     B = sqrt(HW) [x 1/(1-p) under dropout]  (+ max |skip| of the sample, a word the previous producer published)
// is known before the first element is written.  The word pair (bound for the scale, actual maximum for the next bound) travels with
// the tensor; the consuming convolution takes the planes through nemar_planes_hint and skips its absmax + split passes.
//
// One workgroup = one (sample, 8-channel group): 1024 threads x (4 consecutive pixels x 8 channels) — a thread owns whole 16-byte plane
// words, the plane is read once (float4 per channel), statistics are the exact two-pass form of norm.hip from registers.
#include "common.h"
#include "max_words.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2;

__device__ __forceinline__ float np_act(float v, int act, float slope) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
    return v;
}
__device__ __forceinline__ float np_pow2_scale(unsigned maxbits) {       // conv_split16.hip pow2_scale: 2^(11 - floor(log2 max))
    const int e = (int)((maxbits >> 23) & 255u);
    if (e == 0 || e == 255) return 1.f;
    const int se = 127 + 11 - (e - 127);
    if (se < 1 || se > 254) return 1.f;
    return __builtin_bit_cast(float, (unsigned)se << 23);
}
__device__ __forceinline__ unsigned np_f16(float v) {
    const _Float16 h = (_Float16)v;
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ void np_split(float v, unsigned& h, unsigned& l) {
    h = np_f16(v);
    l = np_f16(v - (float)__builtin_bit_cast(_Float16, (unsigned short)h));
}
__device__ __forceinline__ unsigned np_finite_mag(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
    return u < 0x7f800000u ? u : 0u;
}
// Philox4x32-10 exactly as pointwise.hip's dropout_kernel draws it (counter = float4 index over the [N,C,H,W] tensor)
__device__ __forceinline__ void np_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct NormPlanesParams {
    const float* x;            // [N, C, H, W] the convolution's output
    const float* residual;     // [N, C, H, W] or null
    const unsigned* resmax;    // per-sample max |residual| words (required with a residual)
    float* y;                  // [N, C, H, W] fp32 output or null (planes only)
    float* stats;              // [N*C, 2] (mean, rstd)
    u32x4* planes;             // hi plane, lo plane at + plane16 words
    unsigned* scale_words;     // [N] out: the bound the planes were scaled by (what the consumer's epilogue unscales with)
    unsigned* maxw;            // NEMAR_MAX_WORDS(N) buffer for the ACTUAL maxima, or null
    long long plane16;
    int N, C, H, W;
    float eps, slope, bound0;  // bound0 = sqrt(HW) [/ (1 - p)]
    int act;
    int dropout;               // 1: dropout, keep when the random word >= thresh (pointwise.hip's rule)
    unsigned thresh;
    float dscale;
    unsigned seed_lo, seed_hi, offset;
    const unsigned* obase;     // nemar_set_dropout_base word (added to offset) or null
    int dbg;                   // measurement only (nemar_tune(31, bits)): 1 no plane stores, 2 no LDS transpose, 4 no statistics, 8 no fp32 stores
};

// sums of eight per-thread values over the workgroup -> tot[0..7] (all threads)
__device__ __forceinline__ void block_sum8(float* s, float* red, float* tot) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = wave_sum(s[j]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wid * 8 + j] = s[j];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w * 8 + threadIdx.x];
        red[128 + threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) tot[j] = red[128 + j];
}

__global__ __launch_bounds__(1024) void instnorm_planes_kernel(NormPlanesParams p) {
    __shared__ float red[136];
    __shared__ unsigned mred[16];
    __shared__ u32x4 xpose[16][256];                       // per wave: 256 plane words in flight between the two orders
    const int CG = p.C >> 3;
    const int n = blockIdx.x / CG, cg = blockIdx.x - n * CG;
    const int HW = p.H * p.W, W = p.W, H = p.H;
    const int t = threadIdx.x;
    const bool active = 4 * t < HW;
    const size_t cbase = ((size_t)n * p.C + (size_t)cg * 8) * HW;
    float v[8][4];
    float s[8], tot[8], mean[8], rstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) q = *reinterpret_cast<const float4*>(p.x + cbase + (size_t)j * HW + 4 * t);
        v[j][0] = q.x; v[j][1] = q.y; v[j][2] = q.z; v[j][3] = q.w;
        s[j] = (q.x + q.y) + (q.z + q.w);
    }
    const float inv = 1.f / (float)HW;
    if (!(p.dbg & 4)) block_sum8(s, red, tot); else { for (int j = 0; j < 8; ++j) tot[j] = s[j] * 1024.f; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mean[j] = tot[j] * inv;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = active ? v[j][e] - mean[j] : 0.f;
            v[j][e] = d;
            q += d * d;
        }
        s[j] = q;
    }
    if (!(p.dbg & 4)) block_sum8(s, red, tot); else { for (int j = 0; j < 8; ++j) tot[j] = s[j] * 1024.f; }
#pragma unroll
    for (int j = 0; j < 8; ++j) rstd[j] = 1.f / sqrtf(tot[j] * inv + p.eps);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (t == j) {
            const size_t pl = (size_t)n * p.C + (size_t)cg * 8 + j;
            p.stats[2 * pl] = mean[j];
            p.stats[2 * pl + 1] = rstd[j];
        }
    // the bound this sample's planes are scaled by
    float bound = p.bound0;
    if (p.residual) bound += __builtin_bit_cast(float, p.resmax[n]);
    const unsigned bound_bits = __builtin_bit_cast(unsigned, bound);
    if (cg == 0 && t == 0) p.scale_words[n] = bound_bits;
    const float scale = np_pow2_scale(bound_bits);
    unsigned omax = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = np_act(v[j][e] * rstd[j], p.act, p.slope);
        if (p.dropout) {
            const unsigned long long q = ((unsigned long long)n * p.C + (unsigned long long)cg * 8 + j) * (unsigned long long)(HW >> 2) + t;
            unsigned r[4];
            np_philox((unsigned)q, (unsigned)(q >> 32), p.offset + (p.obase ? *p.obase : 0u), 0u, p.seed_lo, p.seed_hi, r);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = r[e] >= p.thresh ? o[e] * p.dscale : 0.f;
        }
        if (p.residual && active) {
            const float4 q = *reinterpret_cast<const float4*>(p.residual + cbase + (size_t)j * HW + 4 * t);
            o[0] += q.x; o[1] += q.y; o[2] += q.z; o[3] += q.w;
        }
        if (p.y && active && !(p.dbg & 8)) *reinterpret_cast<float4*>(p.y + cbase + (size_t)j * HW + 4 * t) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            omax = max(omax, np_finite_mag(o[e]));
            v[j][e] = o[e];
        }
    }
    // v s = h + l: the eight channels of a pixel as one hi and one lo word (v * s is exact, fmaf(v, s, -h) the exact residual: v_fma_mix)
    u32x4 hw[4], lw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f16x2 h, l;
            h[0] = (_Float16)(v[2 * k][e] * scale);
            h[1] = (_Float16)(v[2 * k + 1][e] * scale);
            l[0] = (_Float16)__builtin_fmaf(v[2 * k][e], scale, -(float)h[0]);
            l[1] = (_Float16)__builtin_fmaf(v[2 * k + 1][e], scale, -(float)h[1]);
            hw[e][k] = __builtin_bit_cast(unsigned, h);
            lw[e][k] = __builtin_bit_cast(unsigned, l);
        }
    }
    // ---- plane words: image pixel (row, col) -> plane (row + 1, col + 1); mirrored border rows 0 / H+1 and slots 0 / W+1; the two
    // extra rows and slots of the layout (the reflect data gradient's folded sums: unused by the forward convolution) are zero.
    // A thread owns 4 CONSECUTIVE pixels (float4 loads, one Philox draw per channel); stored like that a wave's 16-byte stores would
    // be 64 bytes apart.  Each wave transposes its 256 words through LDS so that store e of lane L is pixel 256 wave + 64 e + L.
    {
        const int Ws = W + 4;
        u32x4* const hp = p.planes + ((size_t)n * CG + cg) * (size_t)(H + 4) * Ws;
        const u32x4 z = u32x4{0u, 0u, 0u, 0u};
        const int wid = t >> 6, lane = t & 63;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            if (p.dbg & 1) break;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) xpose[wid][(p.dbg & 2) ? e * 64 + lane : lane * 4 + e] = pl ? lw[e] : hw[e];
            __syncthreads();
            u32x4* const dst = hp + (pl ? p.plane16 : 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int P = wid * 256 + e * 64 + lane;
                if (P >= HW) continue;
                const u32x4 word = xpose[wid][e * 64 + lane];
                const int row = P / W, col = P - row * W;
                // (no row list in a private array: a dynamically indexed one lives in scratch memory — 50 us per launch)
                auto put = [&](int prow) {
                    u32x4* const r = dst + (size_t)prow * Ws;
                    r[col + 1] = word;
                    if (col == 1) r[0] = word;
                    if (col == W - 2) r[W + 1] = word;
                    if (col == W - 1) { r[W + 2] = z; r[W + 3] = z; }
                };
                put(row + 1);
                if (row == 1) put(0);
                if (row == H - 2) put(H + 1);
                if (row == 0 || row == H - 1) {
                    u32x4* const r = dst + (size_t)(row == 0 ? H + 2 : H + 3) * Ws;
                    r[col + 1] = z;
                    if (col == 0) r[0] = z;
                    if (col == W - 1) { r[W + 1] = z; r[W + 2] = z; r[W + 3] = z; }
                }
            }
        }
    }
    if (p.maxw) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
        if ((t & 63) == 0) mred[t >> 6] = omax;
        __syncthreads();
        if (t == 0) {
            unsigned m = mred[0];
            for (int i = 1; i < 16; ++i) m = max(m, mred[i]);
            p.maxw[p.N + blockIdx.x] = m;
        }
    }
}

NEMAR_SWITCH(int, g_norm_planes_dbg, 0);
}  // namespace

extern const unsigned* g_dropout_base;          // pointwise.hip

#ifdef NEMAR_AB
void nemar_norm_planes_debug(int bits) { g_norm_planes_dbg = bits; }
#endif

// y (optional) = [residual +] dropout(act(InstanceNorm(x))), stats, AND the fp16 x 3 planes of y for a 3x3 / pad-1 reflect convolution
// (conv_split16.hip layout, 2 * N * (C/8) * (H+4) * (W+4) 16-byte words), scaled by the a-priori bound written to scale_words[n].
NEMAR_API int nemar_instnorm_fwd_planes(const float* x, const float* residual, const void* residual_max_words, float* y, float* stats,
                                        int N, int C, int H, int W, float eps, int act, float slope, float dropout_p,
                                        unsigned long long seed, unsigned offset, void* planes, void* scale_words, void* max_words,
                                        void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && stats && planes && scale_words, "instnorm_fwd_planes: null pointer");
    NEMAR_REQUIRE(N > 0 && N <= 256 && C > 0 && C % 8 == 0 && C / 8 <= NEMAR_MAX_PARTIALS, "instnorm_fwd_planes: bad N=%d C=%d", N, C);
    NEMAR_REQUIRE(H >= 4 && W >= 4 && W % 4 == 0 && H * W <= 4096, "instnorm_fwd_planes: unsupported plane %dx%d (W %% 4 == 0, HW <= 4096)", H, W);
    NEMAR_REQUIRE(act == ACT_NONE || act == ACT_RELU || act == ACT_LRELU, "instnorm_fwd_planes: unsupported act %d", act);
    NEMAR_REQUIRE(!residual || residual_max_words, "instnorm_fwd_planes: a residual needs its per-sample max words (the bound of the sum)");
    NEMAR_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "instnorm_fwd_planes: bad dropout p");
    NEMAR_REQUIRE((((uintptr_t)x | (uintptr_t)residual | (uintptr_t)y | (uintptr_t)planes) & 15) == 0, "instnorm_fwd_planes: pointers must be 16-byte aligned");
    NormPlanesParams p;
    p.x = x; p.residual = residual; p.resmax = (const unsigned*)residual_max_words; p.y = y; p.stats = stats;
    p.planes = (u32x4*)planes; p.scale_words = (unsigned*)scale_words; p.maxw = (unsigned*)max_words;
    p.plane16 = (long long)N * (C / 8) * (H + 4) * (W + 4);
    p.N = N; p.C = C; p.H = H; p.W = W;
    p.eps = eps; p.slope = slope; p.act = act;
    p.bound0 = sqrtf((float)(H * W)) * (dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f);
    const double t = (double)dropout_p * 4294967296.0;
    p.dropout = dropout_p > 0.f ? 1 : 0;
    p.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    p.dscale = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
    p.dbg = g_norm_planes_dbg;
    p.obase = g_dropout_base;
    p.seed_lo = (unsigned)(seed & 0xffffffffu); p.seed_hi = (unsigned)(seed >> 32); p.offset = offset;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(instnorm_planes_kernel, dim3(N * (C / 8)), dim3(1024), 0, st, p);
    if (max_words) max_words_finalize((unsigned*)max_words, N, C / 8, st);
    NEMAR_CHECK_LAUNCH("instnorm_fwd_planes");
    return NEMAR_OK;
}
