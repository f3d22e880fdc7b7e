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
#include "conv_split16.h"

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
    int lazy;                  // the consumer reduces the partial maxima (nemar_set_max_words_lazy)
    int dbg;                   // measurement only (nemar_tune(31, bits)): 1 no plane stores, 2 no LDS transpose, 4 no statistics, 8 no fp32 stores
    u32x4* xw;                 // the weight gradient's pixel-major X planes of y (conv_split16_wgrad.hip layout, reflect border) or null
    long long xplane16;        // 16-byte words per X plane
    int CPR, Hx;               // 8-pixel chunks per padded row, plane rows (>= H + 2; rows beyond the padded image are zero)
};

// the X planes' staging tile: [8 channels][H rows][CPR * 8 padded pixels] 16-bit values (+ 8 per channel: the 16-byte reads of the 8
// channels of one chunk fall into different LDS banks); one plane (hi, then lo) at a time, in the memory of the transpose buffer
constexpr int XT_MAX = 64 * 72;                    // H * CPR * 8 of the largest plane served (64 x 64)
constexpr int NP_LDS_BYTES = 8 * (XT_MAX + 8) * 2 > 16 * 256 * 16 ? 8 * (XT_MAX + 8) * 2 : 16 * 256 * 16;

// sums of eight per-thread values over the workgroup -> tot[0..7] (all threads; wave-uniform: they live in scalar registers)
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
    for (int j = 0; j < 8; ++j) tot[j] = uniform_f(red[128 + j]);
}

// Workgroup id -> (sample, 8-channel group).  Consecutive workgroup ids sit on consecutive XCDs, and the eight channel groups of one
// 64-channel block each write a 128-byte piece of every 1 KiB run of the pixel-major planes ([flat chunk][64 channels][8 pixels]): with the
// plain order those eight pieces leave eight different L2s as eight separate 128-byte write-backs.  Eight consecutive ids OF ONE XCD get the
// eight groups of one block instead, so the pieces meet in that XCD's L2 (round 6b; needs a whole number of 64-unit rounds).
__device__ __forceinline__ int np_unit_of_block(int b, int total) {
    if (total & 63) return b;
    const int x = b & 7, r = b >> 3;
    return (((r >> 3) << 3) + x) * 8 + (r & 7);
}

#ifdef NEMAR_NP_FWD_OCC2          /* variant build (tools/build_variant.py): 8 waves per SIMD = two workgroups per CU, whatever it spills */
__global__ __launch_bounds__(1024, 8) void instnorm_planes_kernel(NormPlanesParams p) {
#else
__global__ __launch_bounds__(1024) void instnorm_planes_kernel(NormPlanesParams p) {
#endif
    __shared__ float red[136];
    __shared__ unsigned mred[16];
    __shared__ __attribute__((aligned(16))) unsigned char np_lds[NP_LDS_BYTES];
    u32x4 (*const xpose)[256] = reinterpret_cast<u32x4 (*)[256]>(np_lds);      // per wave: 256 plane words in flight between the two orders
    const int CG = p.C >> 3;
    const int unit = np_unit_of_block((int)blockIdx.x, (int)gridDim.x);
    const int n = unit / CG, cg = unit - n * CG;
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
    if (p.residual) bound += __builtin_bit_cast(float, sample_max_word(p.resmax, n, p.N));
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
    // ---- the weight gradient's X planes (conv_split16_wgrad.hip: word ((pl N + n) CBLK + cblk) FX + yp CPR + q) 64 + cc = padded pixels
    // 8 q .. 8 q + 7 of channel cc in padded row yp, mirrored border materialised, zero beyond column W + 1 / row H + 1).  A word runs
    // along the row of ONE channel: the 16-bit halves go through an LDS tile [channel][row][padded pixel] and come back as 16-byte reads.
    if (p.xw) {
        unsigned short* const xt = reinterpret_cast<unsigned short*>(np_lds);
        const int RS = p.CPR * 8, CS = H * RS + 8;
        const int row = (4 * t) / W, c0 = 4 * t - row * W;
        const int CBLK = p.C >> 6, FX = p.Hx * p.CPR;
        const int nwords = 8 * FX;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            __syncthreads();
            if (active) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned short u[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) u[e] = (unsigned short)(((pl ? lw[e][j >> 1] : hw[e][j >> 1]) >> (16 * (j & 1))) & 0xffffu);
                    unsigned short* const r = xt + j * CS + row * RS + c0 + 1;       // padded pixel = column + 1
                    r[0] = u[0];
                    *reinterpret_cast<unsigned*>(r + 1) = (unsigned)u[1] | ((unsigned)u[2] << 16);
                    r[3] = u[3];
                    if (c0 == 0) r[-1] = u[1];                                       // padded pixel 0 mirrors column 1
                    if (c0 == W - 4) {
                        r[4] = u[2];                                                 // padded pixel W + 1 mirrors column W - 2
                        for (int z = W + 2; z < RS; ++z) xt[j * CS + row * RS + z] = 0;
                    }
                }
            }
            __syncthreads();
            u32x4* const dst = p.xw + (pl ? p.xplane16 : 0) + (((size_t)n * CBLK + (cg >> 3)) * FX) * 64 + (cg & 7) * 8;
            for (int i = t; i < nwords; i += 1024) {
                const int cc = i & 7, f = i >> 3;
                const int yp = f / p.CPR, q = f - yp * p.CPR;
                u32x4 word = u32x4{0u, 0u, 0u, 0u};
                if (yp <= H + 1) {
                    const int sr = yp == 0 ? 1 : (yp == H + 1 ? H - 2 : yp - 1);
                    word = *reinterpret_cast<const u32x4*>(xt + cc * CS + sr * RS + q * 8);
                }
                dst[(size_t)f * 64 + cc] = word;
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
            p.maxw[p.N + unit] = m;
            if (p.lazy && cg == 0) p.maxw[n] = NEMAR_MAX_LAZY_MARK | (unsigned)CG;       // (no reduction launch: max_words.h)
        }
    }
}


// ---- backward: InstanceNorm (+ activation, + dropout) gradient that writes the OPERAND PLANES of the two gradient calls of the 3x3
// convolution in front of it instead of (or besides) the fp32 tensor ------------------------------------------------------------------
//     g = gy [* dropout mask / (1 - p)] * act'(xhat);   gx = rstd (g - mean(g) - xhat mean(g xhat))          (norm.hip's formula)
//     -> the data gradient's channel-blocked padded planes of gx (conv_split16.hip, SPLIT16_ZERO or SPLIT16_DGRAD_REFLECT content:
//        what split_dual_kernel / split_planes_kernel make of the fp32 tensor) and the weight gradient's G_0 planes
//        (conv_split16_wgrad.hip), both scaled by a power of two from the a-priori bound
//            |gx| <= rstd_max(sample) (2 + sqrt(HW)) max |g|        (|mean g| <= max |g|, |mean(g xhat)| <= max |g|, |xhat| <= sqrt(HW))
//        written to scale_words[n]: the consumer's epilogue unscales with it, exactly as with the forward producer's bound;
//     -> per-plane sums of gx (the convolution's bias gradient is their sum over the batch; mathematically zero, rounding in practice).
// One workgroup = (sample, 8 channels), 1024 threads x (4 pixels x 8 channels) like the forward kernel; x and gy are read once, the
// fp32 result goes through an LDS tile [8][HW + 4] from which both layouts (borders, folded rows of the reflect data gradient) are read.
struct NormBwdPlanesParams {
    const float* x;            // [N, C, H, W] the forward's input (the convolution's output)
    const float* stats;        // [N*C, 2]
    const float* gy;           // [N, C, H, W]
    const unsigned* gymax;     // per-sample max |gy| words
    float* gx;                 // fp32 copy or null
    u32x4* dpl;                // data-gradient planes (hi; lo at + dplane16) or null
    u32x4* gpl;                // weight-gradient G_0 planes (hi; lo at + gplane16) or null
    unsigned* scale_words;     // [N] out
    float* bsum;               // [N * C] out: sum of gx over each plane, or null
    long long dplane16, gplane16;
    int N, C, H, W;
    int CPR, Hg;
    int reflect;               // data-gradient planes: 1 SPLIT16_DGRAD_REFLECT content, 0 SPLIT16_ZERO
    int act;
    float slope, bmul;         // bmul = (2 + sqrt(HW)) [/ (1 - p)]
    int dropout;
    unsigned thresh;
    float dscale;
    unsigned seed_lo, seed_hi, offset;
    const unsigned* obase;
};

__device__ __forceinline__ float np_act_df(float xhat, int act, float slope) {
    if (act == ACT_RELU) return xhat > 0.f ? 1.f : 0.f;
    if (act == ACT_LRELU) return xhat > 0.f ? 1.f : slope;
    return 1.f;
}

// LDS bytes of the backward producer: the 16-bit tile [8][HW + 8], or the fp32 border rows / columns of the reflect fold where those are larger
// (planes of a few rows only)
static size_t np_bwd_lds_bytes(int H, int W) {
    const size_t tile = (size_t)8 * (H * W + 8) * 2, side = (size_t)4 * 8 * (W + H) * sizeof(float);
    return tile > side ? tile : side;
}

// Round 6b: the result used to go through an fp32 tile [8][HW + 4] (131 KB: ONE workgroup per CU, its load / reduce / store phases exposed
// — 74 us stand-alone, 2.7 x that beside the side stream's weight gradients).  Now each value is split ONCE in the registers of the thread
// that computed it and the tile holds ONE 16-bit plane at a time ([8][HW + 8] halves, 66 KB: two workgroups per CU): hi plane -> both
// layouts, then lo plane -> both layouts.  The folded border sums of the reflect data gradient need fp32 operands: the four border rows
// and columns they are made of go through the same memory first (16 KB), the sums in the order plane_value (conv_split16.hip) adds them —
// the planes are bit for bit what the fp32-tile form wrote.
// 4 * threadIdx.x, recomputed where it is used (one shift): as one value hipcc parks it in scratch across the kernel — the one register
// the 64 of two workgroups per CU do not have
__device__ __forceinline__ unsigned np_t4() {
#ifdef NEMAR_HOST_EMULATION
    return 4u * threadIdx.x;
#else
    unsigned v;
    asm volatile("v_lshlrev_b32 %0, 2, %1" : "=v"(v) : "v"((unsigned)threadIdx.x));
    return v;
#endif
}

// Registers (round 6b): a thread keeps ONE array of 32 values — g = gy [x mask] act'(xhat), later the scaled result — and reads x twice (the
// second time from the caches: its own workgroup fetched those 128 KB microseconds ago); statistics, means and scales are wave-uniform
// (scalar registers); every per-channel sum goes wave sum -> LDS at once instead of waiting in eight registers.  <= 64 registers x 1024
// threads and 66 KB of LDS: two of these workgroups share a CU, or one shares it with a 64 KB / ~300-register weight-gradient workgroup
// of the side stream (the fp32-tile form, 131 KB / 107 registers, could do neither: 74 us stand-alone, 203 us beside the side stream).
__global__ __launch_bounds__(1024, 8) void instnorm_bwd_planes_kernel(NormBwdPlanesParams p) {
#ifdef NEMAR_HOST_EMULATION
    __shared__ __attribute__((aligned(16))) unsigned short tile16[8 * (4096 + 8)];
#else
    extern __shared__ __attribute__((aligned(16))) unsigned short tile16[];       // 8 (HW + 8) halves (np_bwd_lds_bytes)
#endif
    __shared__ float red[16 * 16 + 16];                    // [wave][sum 1 of channel 0..7 | sum 2 of channel 0..7], then the 16 totals
    __shared__ float rmax[16];
    const int CG = p.C >> 3;
    const int unit = np_unit_of_block((int)blockIdx.x, (int)gridDim.x);
    const int n = unit / CG, cg = unit - n * CG;
    const int HW = p.H * p.W, W = p.W, H = p.H;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const bool active = 4 * t < HW;
    const size_t cbase = ((size_t)n * p.C + (size_t)cg * 8) * HW;
    #define NP_TOFF (active ? np_t4() : 0u)                 /* (inactive threads load element 0 and use nothing of it) */
    float g[8][4];
    // largest rstd of the sample (all C planes): part of the bound
    {
        float m = 0.f;
        for (int c = t; c < p.C; c += 1024) m = fmaxf(m, p.stats[2 * ((size_t)n * p.C + c) + 1]);
        m = wave_max(m);
        if (lane == 0) rmax[wid] = m;
    }
    const float dneg = p.act == ACT_RELU ? 0.f : (p.act == ACT_LRELU ? p.slope : 1.f);      // act'(xhat <= 0)
    // ---- pass 1: g and the two sums of every channel ----
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const size_t pl = (size_t)n * p.C + (size_t)cg * 8 + j;
        const float mean = p.stats[2 * pl], rstd = p.stats[2 * pl + 1];
        const float4 a = *reinterpret_cast<const float4*>(p.x + cbase + (size_t)j * HW + NP_TOFF);
        const float4 b = *reinterpret_cast<const float4*>(p.gy + cbase + (size_t)j * HW + NP_TOFF);
        g[j][0] = b.x; g[j][1] = b.y; g[j][2] = b.z; g[j][3] = b.w;
        if (p.dropout) {
            const unsigned long long q = ((unsigned long long)n * p.C + (unsigned long long)cg * 8 + j) * (unsigned long long)(HW >> 2) + t;
            unsigned r[4];
            np_philox((unsigned)q, (unsigned)(q >> 32), p.offset + (p.obase ? *p.obase : 0u), 0u, p.seed_lo, p.seed_hi, r);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[j][e] = r[e] >= p.thresh ? g[j][e] * p.dscale : 0.f;
        }
        const float xv[4] = {a.x, a.y, a.z, a.w};
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h = active ? (xv[e] - mean) * rstd : 0.f;
            const float u = active ? g[j][e] * (h > 0.f ? 1.f : dneg) : 0.f;
            g[j][e] = u;
            s1 += u;
            s2 += u * h;
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { red[wid * 16 + j] = s1; red[wid * 16 + 8 + j] = s2; }
    }
    __syncthreads();
    if (t < 16) {
        float tt = 0.f;
        for (int w = 0; w < 16; ++w) tt += red[w * 16 + t];          // wave order: a fixed association
        red[256 + t] = tt;
    }
    __syncthreads();
    // the bound this sample's planes are scaled by (rmax was written before the first barrier)
    float rm = rmax[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) rm = fmaxf(rm, rmax[w]);
    const float bound = rm * p.bmul * __builtin_bit_cast(float, sample_max_word(p.gymax, n, p.N));
    const unsigned bound_bits = __builtin_bit_cast(unsigned, bound);
    if (cg == 0 && t == 0) p.scale_words[n] = bound_bits;
    const float scale = uniform_f(np_pow2_scale(bound_bits));
    const float inv = 1.f / (float)HW;
    // ---- pass 2: x again (a real read: through an opaque copy of the pointer), the result, its per-plane sums; g <- scaled result ----
    const float* x2 = p.x;
#ifndef NEMAR_HOST_EMULATION
    asm volatile("" : "+s"(x2));
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const size_t pl = (size_t)n * p.C + (size_t)cg * 8 + j;
        const float mean = p.stats[2 * pl], rstd = p.stats[2 * pl + 1];
        const float m1 = uniform_f(red[256 + j] * inv), m2 = uniform_f(red[256 + 8 + j] * inv);
        const float4 a = *reinterpret_cast<const float4*>(x2 + cbase + (size_t)j * HW + NP_TOFF);
        const float xv[4] = {a.x, a.y, a.z, a.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h = active ? (xv[e] - mean) * rstd : 0.f;
            o[e] = rstd * (g[j][e] - m1 - h * m2);
        }
        if (active && p.gx) *reinterpret_cast<float4*>(p.gx + cbase + (size_t)j * HW + NP_TOFF) = make_float4(o[0], o[1], o[2], o[3]);
        if (p.bsum) {                                      // (red[0 .. 255] was last read before the barrier above; the totals sit behind it)
            const float bsj = wave_sum(active ? (o[0] + o[1]) + (o[2] + o[3]) : 0.f);
            if (lane == 0) red[wid * 16 + j] = bsj;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) g[j][e] = active ? o[e] * scale : 0.f;
    }
    if (p.bsum) {
        __syncthreads();
        if (t < 8) {
            float tt = 0.f;
            for (int w = 0; w < 16; ++w) tt += red[w * 16 + t];
            p.bsum[(size_t)n * p.C + (size_t)cg * 8 + t] = tt;
        }
    }
    const int Hp = H + 4, Ws = W + 4;
    u32x4* const hp = p.dpl ? p.dpl + ((size_t)n * CG + cg) * (size_t)Hp * Ws : nullptr;
    const int row = active ? (int)np_t4() / W : 0, c0 = (int)np_t4() - row * W;
    // ---- phase 0 (reflect data gradient only): the folded rows H + 2, H + 3 and slots W + 2, W + 3 of the data-gradient planes from the fp32
    // border rows {0, 2, H - 3, H - 1} and columns {0, 2, W - 3, W - 1}: rb[4][8][W], cb[4][8][H] ----
    if (p.dpl && p.reflect) {
        float* const rb = reinterpret_cast<float*>(tile16);
        float* const cb = rb + 4 * 8 * W;
        __syncthreads();
        if (active) {
            const int ri = row == 0 ? 0 : (row == 2 ? 1 : (row == H - 3 ? 2 : (row == H - 1 ? 3 : -1)));
            // (first match, as the reader below looks them up: H == 5 / W == 5 name one row / column twice)
            if (ri >= 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<float4*>(rb + (ri * 8 + j) * W + c0) = make_float4(g[j][0], g[j][1], g[j][2], g[j][3]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = c0 + e;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int want = k == 0 ? 0 : (k == 1 ? 2 : (k == 2 ? W - 3 : W - 1));
                    if (col == want) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) cb[(k * 8 + j) * H + row] = g[j][e];
                    }
                }
            }
        }
        __syncthreads();
        // aux positions: (H + 2 | H + 3, slot 1 .. W and W + 2, W + 3) and (row 1 .. H, slot W + 2 | W + 3)
        const int naux = 2 * (W + 2) + 2 * H;
        for (int i = t; i < naux; i += 1024) {
            int prow, pslot;
            if (i < 2 * (W + 2)) {
                prow = H + 2 + i / (W + 2);
                const int q = i - (i / (W + 2)) * (W + 2);
                pslot = q < W ? q + 1 : (q == W ? W + 2 : W + 3);
            } else {
                const int q = i - 2 * (W + 2);
                prow = 1 + (q >> 1);
                pslot = W + 2 + (q & 1);
            }
            // operands as plane_value names them: rows ya (+ yb), columns xa (+ xb)
            const int ya = prow == H + 2 ? 0 : (prow == H + 3 ? H - 3 : prow - 1), yb = prow == H + 2 ? 2 : (prow == H + 3 ? H - 1 : -1);
            const int xa = pslot == W + 2 ? 0 : (pslot == W + 3 ? W - 3 : pslot - 1), xb = pslot == W + 2 ? 2 : (pslot == W + 3 ? W - 1 : -1);
            auto at = [&](int ch, int y, int x) -> float {
                // (y, x) is in a border row or a border column by construction
                if (yb >= 0) {          // folded row: both rows are border rows
                    const int ri = y == 0 ? 0 : (y == 2 ? 1 : (y == H - 3 ? 2 : 3));
                    return rb[(ri * 8 + ch) * W + x];
                }
                const int ci = x == 0 ? 0 : (x == 2 ? 1 : (x == W - 3 ? 2 : 3));
                return cb[(ci * 8 + ch) * H + y];
            };
            u32x4 hwd, lwd;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float vv[2];
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                    const int ch = 2 * k + z;
                    float v = at(ch, ya, xa);
                    if (yb >= 0) v += at(ch, yb, xa);
                    if (xb >= 0) {
                        float u = at(ch, ya, xb);
                        if (yb >= 0) u += at(ch, yb, xb);
                        v += u;
                    }
                    vv[z] = v;
                }
                f16x2 h, l;
                h[0] = (_Float16)vv[0];
                h[1] = (_Float16)vv[1];
                l[0] = (_Float16)(vv[0] - (float)h[0]);
                l[1] = (_Float16)(vv[1] - (float)h[1]);
                hwd[k] = __builtin_bit_cast(unsigned, h);
                lwd[k] = __builtin_bit_cast(unsigned, l);
            }
            hp[(size_t)prow * Ws + pslot] = hwd;
            hp[p.dplane16 + (size_t)prow * Ws + pslot] = lwd;
        }
    }
    // ---- the split in the registers that hold the values: element pairs (e, e + 1) of channel j as one dword of plane pl ----
    auto halves = [&](int pl, int j, int k) -> unsigned {
        f16x2 h;
        h[0] = (_Float16)g[j][2 * k];
        h[1] = (_Float16)g[j][2 * k + 1];
        if (pl) {
            f16x2 l;
            l[0] = (_Float16)(g[j][2 * k] - (float)h[0]);
            l[1] = (_Float16)(g[j][2 * k + 1] - (float)h[1]);
            return __builtin_bit_cast(unsigned, l);
        }
        return __builtin_bit_cast(unsigned, h);
    };
    const int TS = HW + 8;
    const int KBLK = p.C >> 6, F = p.Hg * p.CPR;
    u32x4* const gdst = p.gpl ? p.gpl + (((size_t)n * KBLK + (cg >> 3)) * F) * 64 + (cg & 7) * 8 : nullptr;
#pragma unroll 1                                           // (side by side, hipcc keeps the hi halves of the first pass for the second)
    for (int pl = 0; pl < 2; ++pl) {
        __syncthreads();                                   // (the previous phase's readers are done with the memory)
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<uint2*>(tile16 + j * TS + np_t4()) = make_uint2(halves(pl, j, 0), halves(pl, j, 1));
        }
        __syncthreads();
        // data-gradient planes: word (row, slot) = the 8 channels of one plane position; the folded rows / slots were written in phase 0
        if (p.dpl) {
            u32x4* const dst = hp + (pl ? p.dplane16 : 0);
            for (int i = t; i < Hp * Ws; i += 1024) {
                const int prow = i / Ws, pslot = i - prow * Ws;
                const bool inner = prow >= 1 && prow <= H && pslot >= 1 && pslot <= W;
                if (!inner && p.reflect && (prow >= H + 2 ? (pslot >= 1 && pslot != W + 1) : (prow >= 1 && prow <= H && pslot >= W + 2))) continue;
                u32x4 word = u32x4{0u, 0u, 0u, 0u};
                if (inner) {
                    const unsigned short* const src = tile16 + (prow - 1) * W + (pslot - 1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) word[k] = (unsigned)src[(2 * k) * TS] | ((unsigned)src[(2 * k + 1) * TS] << 16);
                }
                dst[i] = word;
            }
        }
        // weight-gradient G_0 planes: word ((n KBLK + kblk) F + y CPR + q) 64 + kk = pixels 8 q .. 8 q + 7 of row y of channel kk
        if (p.gpl) {
            u32x4* const dst = gdst + (pl ? p.gplane16 : 0);
            for (int i = t; i < 8 * F; i += 1024) {
                const int cc = i & 7, f = i >> 3;
                const int y = f / p.CPR, q = f - y * p.CPR;
                u32x4 word = u32x4{0u, 0u, 0u, 0u};
                if (y < H && q * 8 < W) word = *reinterpret_cast<const u32x4*>(tile16 + cc * TS + y * W + q * 8);
                dst[(size_t)f * 64 + cc] = word;
            }
        }
    }
}
#undef NP_TOFF

NEMAR_SWITCH(int, g_norm_planes_dbg, 0);
}  // namespace

extern const unsigned* g_dropout_base;          // pointwise.hip

#ifdef NEMAR_AB
void nemar_norm_planes_debug(int bits) { g_norm_planes_dbg = bits; }
#endif

// y (optional) = [residual +] dropout(act(InstanceNorm(x))), stats, AND the fp16 x 3 planes of y for a 3x3 / pad-1 reflect convolution
// (conv_split16.hip layout, 2 * N * (C/8) * (H+4) * (W+4) 16-byte words), scaled by the a-priori bound written to scale_words[n].
// wgrad_planes (optional): ALSO the pixel-major X planes the weight gradient of that convolution reads (conv_split16_wgrad.hip layout,
// nemar_conv2d_x_planes_bytes(N, C, H, W, 3) bytes, same scale) — the layer then needs neither split pass.
NEMAR_API size_t nemar_conv2d_x_planes_bytes(int N, int C, int H, int W, int KS) {
    if (KS != 3 || N <= 0 || N > 256 || C <= 0 || C % 64 || H < 4 || W < 8 || W % 8 || H * W > 4096) return 0;
    const int CPR = (W + 2 + 7) / 8, Hg = (H + 3) / 4 * 4;
    if (H * CPR * 8 > XT_MAX) return 0;
    return (size_t)2 * N * C * (Hg + 2) * CPR * 16;
}

NEMAR_API int nemar_instnorm_fwd_planes(const float* x, const float* residual, const void* residual_max_words, float* y, float* stats,
                                        int N, int C, int H, int W, float eps, int act, float slope, float dropout_p,
                                        unsigned long long seed, unsigned offset, void* planes, void* scale_words, void* max_words,
                                        void* wgrad_planes, void* stream) {
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
    p.lazy = (max_words && nemar_max_words_lazy() && C / 8 <= 0xFFFF) ? 1 : 0;
    p.obase = g_dropout_base;
    p.seed_lo = (unsigned)(seed & 0xffffffffu); p.seed_hi = (unsigned)(seed >> 32); p.offset = offset;
    p.xw = (u32x4*)wgrad_planes;
    p.CPR = (W + 2 + 7) / 8;
    p.Hx = (H + 3) / 4 * 4 + 2;
    p.xplane16 = (long long)N * C * p.Hx * p.CPR;
    NEMAR_REQUIRE(!wgrad_planes || (nemar_conv2d_x_planes_bytes(N, C, H, W, 3) > 0 && ((uintptr_t)wgrad_planes & 15) == 0),
                  "instnorm_fwd_planes: this shape has no weight-gradient planes (N=%d C=%d %dx%d)", N, C, H, W);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(instnorm_planes_kernel, dim3(N * (C / 8)), dim3(1024), 0, st, p);
    if (max_words && !p.lazy) max_words_finalize((unsigned*)max_words, N, C / 8, st);
    NEMAR_CHECK_LAUNCH("instnorm_fwd_planes");
    return NEMAR_OK;
}

// gx (optional fp32) = InstanceNorm backward of gy through [dropout ->] act -> InstanceNorm (the order of nemar_instnorm_fwd_planes), AND the
// operand planes of gx for the two gradient calls of the 3x3 / pad-1 convolution that produced x: `dgrad_planes`
// (2 N (C/8) (H+4) (W+4) 16-byte words; pad_mode 1: the reflect data gradient's folded content, 0: zero padding) and `wgrad_planes`
// (nemar_conv2d_gy_planes_bytes: the G_0 planes), scaled by the a-priori bound written to scale_words[n] (pass it as the max words of
// gx to the two calls).  gy_max_words: per-sample max |gy| (N words).  bias_partials (optional, [N, C]): sum of gx over each plane.
NEMAR_API int nemar_instnorm_bwd_planes(const float* x, const float* stats, const float* gy, const void* gy_max_words, int N, int C,
                                        int H, int W, int act, float slope, float dropout_p, unsigned long long seed, unsigned offset,
                                        int pad_mode, float* gx, void* dgrad_planes, void* wgrad_planes, void* scale_words,
                                        float* bias_partials, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && stats && gy && gy_max_words && scale_words, "instnorm_bwd_planes: null pointer");
    NEMAR_REQUIRE(dgrad_planes || wgrad_planes, "instnorm_bwd_planes: no planes requested (use nemar_instnorm_bwd)");
    NEMAR_REQUIRE(N > 0 && N <= 256 && C > 0 && C % 8 == 0, "instnorm_bwd_planes: bad N=%d C=%d", N, C);
    NEMAR_REQUIRE(H >= 4 && W >= 4 && W % 4 == 0 && H * W <= 4096, "instnorm_bwd_planes: unsupported plane %dx%d (W %% 4 == 0, HW <= 4096)", H, W);
    NEMAR_REQUIRE(!wgrad_planes || (C % 64 == 0 && W % 8 == 0), "instnorm_bwd_planes: weight-gradient planes need C %% 64 == 0 and W %% 8 == 0");
    NEMAR_REQUIRE(act == ACT_NONE || act == ACT_RELU || act == ACT_LRELU, "instnorm_bwd_planes: unsupported act %d", act);
    NEMAR_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "instnorm_bwd_planes: bad dropout p");
    NEMAR_REQUIRE(pad_mode == 0 || pad_mode == 1, "instnorm_bwd_planes: bad pad_mode %d", pad_mode);
    NEMAR_REQUIRE((((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gx | (uintptr_t)dgrad_planes | (uintptr_t)wgrad_planes) & 15) == 0,
                  "instnorm_bwd_planes: pointers must be 16-byte aligned");
    NormBwdPlanesParams p;
    p.x = x; p.stats = stats; p.gy = gy; p.gymax = (const unsigned*)gy_max_words; p.gx = gx;
    p.dpl = (u32x4*)dgrad_planes; p.gpl = (u32x4*)wgrad_planes; p.scale_words = (unsigned*)scale_words; p.bsum = bias_partials;
    p.N = N; p.C = C; p.H = H; p.W = W;
    p.CPR = (W + 2 + 7) / 8;
    p.Hg = (H + 3) / 4 * 4;
    p.dplane16 = (long long)N * (C / 8) * (H + 4) * (W + 4);
    p.gplane16 = (long long)N * C * p.Hg * p.CPR;
    p.reflect = pad_mode;
    p.act = act; p.slope = slope;
    p.dscale = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
    p.bmul = (2.f + sqrtf((float)(H * W))) * p.dscale;
    const double t = (double)dropout_p * 4294967296.0;
    p.dropout = dropout_p > 0.f ? 1 : 0;
    p.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    p.obase = g_dropout_base;
    p.seed_lo = (unsigned)(seed & 0xffffffffu); p.seed_hi = (unsigned)(seed >> 32); p.offset = offset;
    const size_t lds = nemar_lds_bytes(reinterpret_cast<const void*>(&instnorm_bwd_planes_kernel), np_bwd_lds_bytes(H, W), false);
#ifdef NEMAR_HOST_EMULATION
    hipLaunchKernelGGL(instnorm_bwd_planes_kernel, dim3(N * (C / 8)), dim3(1024), 0, (hipStream_t)stream, p);
    (void)lds;
#else
    hipLaunchKernelGGL(instnorm_bwd_planes_kernel, dim3(N * (C / 8)), dim3(1024), lds, (hipStream_t)stream, p);
#endif
    NEMAR_CHECK_LAUNCH("instnorm_bwd_planes");
    return NEMAR_OK;
}

// gb[C] += sum over the batch of bias_partials [N, C] (fixed order: bitwise reproducible)
NEMAR_API int nemar_bias_from_partials(const float* bias_partials, int N, int C, float* gb, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(bias_partials && gb && N > 0 && C > 0, "bias_from_partials: bad arguments");
    nemar_sum_partials(bias_partials, C, N, gb, C, true, (hipStream_t)stream);
    NEMAR_CHECK_LAUNCH("bias_from_partials");
    return NEMAR_OK;
}
