// nemar_amd — weight (+ bias) gradient of the general convolutions on the 16-bit matrix pipe at fp32 accuracy, operand split inside
// the kernel (companion of conv_s16g.hip; reference: autograd of nn.Conv2d / nn.ConvTranspose2d at models/networks.py:355-374,
// 576-593 and models/stn/unet_stn.py:28-102, layers.py:73-106 — on the exact-fp32 MFMA these ran at 60-100 TF, 5.4 ms of a step).
//
//     dW[k][c][r][s] = sum_{n, oy, ox} gy[n][k][oy][ox] * xpad[n][c][oy st + r - p][ox st + s - p],      gb[k] = sum gy[n][k][oy][ox]
//
// GEMM view: rows = k (from gy), columns = c (from x), one accumulator tile per tap, reduction = pixels.  An MFMA operand is 8
// consecutive reduction elements, i.e. 8 consecutive pixels of one channel row — contiguous in NCHW memory, so the loader needs no
// transposition: a thread takes (channel, 8-pixel chunk), loads two (stride 2: four) 16-byte words, converts, writes one LDS word
// per plane.  The taps' horizontal shifts would misalign every operand read; instead the reduction runs over SOURCE positions
//     j  (x' = st j + par is the source column, par = (s - p) mod st),   ox = j + d(s),  d(s) = (par - s + p) / st
// so that the x operand is always the aligned chunk [8q, 8q + 8) of the (parity-de-interleaved) source row and the gy operand is
// the chunk shifted by d(s): the loader builds the few shifted copies G_d (3x3 stride 1: d = +1, 0, -1) from one register window
// gy[8q + dmin .. 8q + 7 + dmax] — conversions happen once per element, a shifted copy costs four pack instructions per plane.
// Reflect padding folds into the copies (G_{+1} gets gy[0] added at x' = 1, G_{-1} gets gy[W-1] at x' = W - 2; mirrored rows are a
// row index), zero padding is absent source rows / zero window elements.
//
// Scaling without any cross-wave agreement: the power-of-two scale of an operand ROW (one k of gy, one c of x) may be anything as
// long as the accumulators know it; a row is owned by the NCH adjacent lanes that load its chunks, which keep a running maximum
// exponent (DPP max over the lane group), convert with 2^(14 - E), and publish E as one byte per row next to the step's operand
// words.  The MFMA waves compare the bytes of their 32 rows / 32 columns with the values their accumulators are scaled by and, when
// a row's exponent grew (rare after the first steps), multiply the affected accumulator registers by the exact power-of-two ratio
// before accumulating the step.  Each output element's error is relative to ITS OWN (k, c) rows over the slab.
//
// Work split: workgroup = TB x TB channels (64 x 64: waves 2 x 2; 32 x 32: the four waves take the four MFMA k-steps of a step) x ALL
// taps x a slab of RB output rows of one image; slabs are summed in order (nemar_sum_partials): bitwise reproducible.  A step = P
// source positions (32 / 64) of one output row: double-buffered LDS, one barrier per step; the loads of step t + 2 are in flight
// while step t + 1 is converted and step t multiplied.  The bias gradient is the loader threads' own sum of what they read.
#include "common.h"
#include "conv_s16g.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TEXP = 14;

__device__ __forceinline__ float pow2f(int biased) {
    return biased < 1 ? 0.f : __builtin_bit_cast(float, (unsigned)(biased > 254 ? 254 : biased) << 23);
}
__device__ __forceinline__ int mirror(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ unsigned finite_abs(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
    return u < 0x7f800000u ? u : 0u;
}
// max over the G lanes (4 or 8, aligned) that share a channel row
template <int G>
__device__ __forceinline__ unsigned group_max(unsigned x) {
#ifdef NEMAR_HOST_EMULATION
#pragma unroll
    for (int o = 1; o < G; o <<= 1) x = max(x, (unsigned)__shfl_xor((int)x, o, 64));
    return x;
#else
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    if (G == 8) x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false));   // row_half_mirror: quad 0 <-> quad 1
    return x;
#endif
}

// tap s of a KS-wide filter with padding PADC, stride ST:  source column x' = ST j + par(s),  output column ox = j + d(s)
constexpr int tap_par(int s, int padc, int st) { return ((s - padc) % st + st) % st; }
constexpr int tap_d(int s, int padc, int st) { return (tap_par(s, padc, st) - s + padc) / st; }
constexpr int tap_dmin(int ks, int padc, int st) {
    int m = 1 << 20;
    for (int s = 0; s < ks; ++s) m = tap_d(s, padc, st) < m ? tap_d(s, padc, st) : m;
    return m;
}
constexpr int tap_dmax(int ks, int padc, int st) {
    int m = -(1 << 20);
    for (int s = 0; s < ks; ++s) m = tap_d(s, padc, st) > m ? tap_d(s, padc, st) : m;
    return m;
}

struct WgGParams {
    const float* x0; const float* x1; int C0, C1;
    const float* gy;
    float* part;                 // slabs [nslab][K][C][KS KS]
    float* partb;                // bias partials [nslab][K] or null
    int N, H, W, C, K, OH, OW;
    int reflect, pad;
    int RB, spi;                 // output rows per slab, slabs per image
    int spr;                     // steps per row
    int KT, CT;                  // channel tiles
    int xcd;                     // XCD-aware workgroup -> (slab, tile) mapping (grid % 8 == 0)
    int dbg;                     // ablation bits (tools/): 1 no MFMAs, 2 no loads, 4 no LDS writes, 8 no exponent check, 16 no barrier
};

// KS x KS taps, stride ST, pad 1 (KS = 1: pad 0).  TB = 64: waves 2 x 2 over (k, c), P = 32 positions per tick; TB = 32: one channel
// tile, the four waves split the four k-steps of a P = 64 tick and add their accumulators at the end.
//
// Order of the reduction inside a slab: column strips of P positions outermost, output rows inside — so that consecutive ticks
// move DOWN the image and the KS source rows an output row needs are a ROLLING WINDOW: every source row segment is loaded and
// converted once, into a ring of LDS row slots (RING = what the taps read + what the next tick writes); only the gy row (with its
// shifted copies) is per tick, double-buffered.  Tick tau of a strip loads source rows ST (oy0 + tau) - 1 + e (e < ST) and
// multiplies output row oy0 + tau - LAG (its last source row has just arrived); the first LAG ticks of a strip only load.
// Each ring row carries its own exponent byte per channel (it was converted with the scale current at its tick), so the
// accumulator tiles of filter row r keep their own column scale.
template <int KS, int ST, int TB>
__global__ __launch_bounds__(256) void s16g_wgrad_kernel(WgGParams p) {
    constexpr int PADC = KS == 1 ? 0 : 1;
    constexpr int NCH = TB == 64 ? 4 : 8;                 // 8-position chunks per tick
    constexpr int TP = TB + 4;                            // LDS row pitch (words): the NCH lanes of a row hit distinct 16-byte slots
    constexpr int NT = KS * KS;
    constexpr int DMIN = tap_dmin(KS, PADC, ST), DMAX = tap_dmax(KS, PADC, ST);       // 3x3 stride 1: -1 .. +1; 3x3 stride 2: 0 .. +1
    constexpr int ND = DMAX - DMIN + 1;                   // shifted copies of gy
    constexpr int WIN = 8 + DMAX - DMIN;                  // register window of gy
    constexpr int LAG = (KS - 1) / ST;
    constexpr int RING = ST * LAG + 2 * ST;               // source-row slots: read ST LAG + KS - 1 back, written ST ahead
    constexpr int GW = ND * 2 * NCH * TP;                 // words of the gy copies of one tick: [copy][plane][chunk][row]
    constexpr int XR = ST * 2 * NCH * TP;                 // words of one source row slot: [parity][plane][chunk][row]
    constexpr int EW = TB / 16;                           // words of one exponent byte array
    constexpr int GBASE = 0, XBASE = 2 * GW, EGBASE = XBASE + RING * XR, EXBASE = EGBASE + 2 * EW;
    __shared__ __attribute__((aligned(16))) u32x4 smem[EXBASE + RING * EW];

    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    // consecutive workgroup ids sit on consecutive XCDs: give every XCD a contiguous run of (slab, tile) pairs, so that the tiles of
    // a slab — which read the same gy / x rows — share one L2
    int t = blockIdx.x;
    if (p.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);
    const int tiles = p.KT * p.CT;
    const int slab = t / tiles, tile = t - slab * tiles;
    const int kt = tile / p.CT, ct = tile - kt * p.CT;
    // slab = (image, row block, column strip)
    const int strip = slab % p.spr, rbi = (slab / p.spr) % p.spi, n = slab / (p.spr * p.spi);
    const int oy0 = rbi * p.RB;
    const int oy1 = min(oy0 + p.RB, p.OH);
    const int nticks = oy1 - oy0 + LAG;

    // ---- loader role: thread = (channel row, chunk of the tick) for gy row k and x row c ----
    const int lch = tid % NCH, lrow = tid / NCH;          // lrow < TB
    const int kg = kt * TB + lrow, cg = ct * TB + lrow;
    const float* const gyrow = p.gy + ((size_t)n * p.K + kg) * p.OH * p.OW + lch * 8;
    const float* const xrow = (cg < p.C0 ? p.x0 + ((size_t)n * p.C0 + cg) * p.H * p.W
                                         : p.x1 + ((size_t)n * p.C1 + (cg - p.C0)) * p.H * p.W) + ST * lch * 8;
    constexpr int DEPTH = 4;                              // register sets: the loads of tick g + DEPTH are issued at tick g (a tick is
                                                          // shorter than the memory latency under load: ~3 us measured)
    float gwin[DEPTH][WIN + 1];                           // (+ 1: the pair conversion reads one element beyond)
    float xv[DEPTH][ST][8 * ST];
    int Eg = TEXP + 2, Ex = TEXP + 2;                      // running biased exponents of this thread's rows
    float bsum = 0.f;
    int ltau = 0;                                          // the next tick to LOAD ...
    int ctau = 0;                                          // ... and the next one to CONVERT (ticks are issued in order)

    // issue the loads of the next tick into register set set_: unconditional loads from clamped addresses (masked at conversion).
    // Every chunk of a tick is inside the row (eligibility: the row is a whole number of ticks, OW == W / ST).
#define WG_LOAD(set_)                                                                                                   \
    {   /* always the same number of load instructions, no branches: hipcc then counts its vmcnt waits exactly and the three ticks   \
           issued after the one being converted stay in flight (with conditional loads it fell back to vmcnt(0) every tick) */      \
        const int tl_ = min(ltau, nticks - 1);                                                                          \
        _Pragma("unroll") for (int e = 0; e < ST; ++e) {                                                                \
            int iy_ = ST * (oy0 + tl_) - PADC + e;                                                                      \
            iy_ = p.reflect ? mirror(iy_, p.H) : iy_;                           /* (zero padding: zeroed at conversion) */ \
            iy_ = min(max(iy_, 0), p.H - 1);                                                                            \
            const float* const x_ = xrow + (size_t)iy_ * p.W + strip * (NCH * 8 * ST);                                  \
            _Pragma("unroll") for (int i = 0; i < 8 * ST; i += 4) {                                                     \
                const f32x4 q_ = *reinterpret_cast<const f32x4*>(x_ + i);                                               \
                xv[set_][e][i] = q_[0]; xv[set_][e][i + 1] = q_[1]; xv[set_][e][i + 2] = q_[2]; xv[set_][e][i + 3] = q_[3]; \
            }                                                                                                           \
        }                                                                                                               \
        {                                                                                                               \
            const int jc_ = (strip * NCH + lch) * 8;                                                                    \
            const float* const g_ = gyrow + (size_t)min(max(oy0 + tl_ - LAG, oy0), oy1 - 1) * p.OW + strip * (NCH * 8); \
            const f32x4 q0_ = *reinterpret_cast<const f32x4*>(g_), q1_ = *reinterpret_cast<const f32x4*>(g_ + 4);       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) { gwin[set_][i - DMIN] = q0_[i]; gwin[set_][i + 4 - DMIN] = q1_[i]; } \
            _Pragma("unroll") for (int e = 0; e < WIN; ++e)                                                             \
                if (e < -DMIN) gwin[set_][e] = g_[jc_ + DMIN + e < 0 ? 0 : DMIN + e];                                   \
                else if (e >= 8 - DMIN) gwin[set_][e] = g_[jc_ + DMIN + e >= p.OW ? 7 : DMIN + e];                      \
        }                                                                                                               \
        gwin[set_][WIN] = 0.f;                                                                                          \
        ++ltau;                                                                                                         \
    }

    // convert register set set_ (the next tick in order): source rows into their ring slots, the gy row into buffer (tick & 1)
#define WG_CONVERT(set_)                                                                                                \
    {                                                                                                                   \
        /* ---- source rows ---- */                                                                                     \
        float mxf_ = 0.f;                                                                                               \
        _Pragma("unroll") for (int e = 0; e < ST; ++e) {                                                                \
            const int iy_ = ST * (oy0 + ctau) - PADC + e;                                                               \
            if ((unsigned)iy_ >= (unsigned)p.H && !(p.reflect && iy_ >= -1 && iy_ <= p.H)) {   /* the same for every lane */ \
                _Pragma("unroll") for (int i = 0; i < 8 * ST; ++i) xv[set_][e][i] = 0.f;                                \
            }                                                                                                           \
            _Pragma("unroll") for (int i = 0; i < 8 * ST; i += 2)                                                       \
                mxf_ = fmaxf(mxf_, fmaxf(__builtin_fabsf(xv[set_][e][i]), __builtin_fabsf(xv[set_][e][i + 1])));        \
        }                                                                                                               \
        const unsigned mx_ = group_max<NCH>(__builtin_bit_cast(unsigned, mxf_));                                        \
        Ex = max(Ex, min((int)(mx_ >> 23), 254));                                                                       \
        const float sx_ = pow2f(127 + TEXP + 127 - Ex);                                                                 \
        _Pragma("unroll") for (int e = 0; e < ST; ++e) {                                                                \
            const int slot_ = (ST * ctau + e) % RING;                                                                    \
            u32x4* const Xb_ = smem + XBASE + slot_ * XR;                                                               \
            _Pragma("unroll") for (int par = 0; par < ST; ++par) {                                                      \
                u32x4 hi_, lo_;                                                                                         \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
                    const float a0_ = xv[set_][e][ST * (2 * i) + par], a1_ = xv[set_][e][ST * (2 * i + 1) + par];       \
                    f16x2 h_, l_;                                                                                       \
                    h_[0] = (_Float16)(a0_ * sx_); h_[1] = (_Float16)(a1_ * sx_);                                       \
                    l_[0] = (_Float16)__builtin_fmaf(a0_, sx_, -(float)h_[0]);                                          \
                    l_[1] = (_Float16)__builtin_fmaf(a1_, sx_, -(float)h_[1]);                                          \
                    hi_[i] = __builtin_bit_cast(unsigned, h_);                                                          \
                    lo_[i] = __builtin_bit_cast(unsigned, l_);                                                          \
                }                                                                                                       \
                if (!(p.dbg & 4)) { Xb_[((par * 2 + 0) * NCH + lch) * TP + lrow] = hi_;                                 \
                Xb_[((par * 2 + 1) * NCH + lch) * TP + lrow] = lo_; }                                                   \
            }                                                                                                           \
            if (lch == 0) reinterpret_cast<unsigned char*>(smem + EXBASE + slot_ * EW)[lrow] = (unsigned char)Ex;       \
        }                                                                                                               \
        /* ---- gy row of output row oy0 + tau - LAG ---- */                                                            \
        if (ctau >= LAG) {                                                                                              \
            u32x4* const Gb_ = smem + GBASE + (ctau & 1) * GW;                                                          \
            const int jc_ = (strip * NCH + lch) * 8;                                                                   \
            float* const w_ = gwin[set_];                                                                               \
            _Pragma("unroll") for (int e = 0; e < WIN; ++e) {                                                           \
                if (e < -DMIN) w_[e] = jc_ + DMIN + e < 0 ? 0.f : w_[e];                                                \
                else if (e >= 8 - DMIN) w_[e] = jc_ + DMIN + e >= p.OW ? 0.f : w_[e];                                   \
                else bsum += w_[e];                                                                                     \
            }                                                                                                           \
            float mgf_ = 0.f;                                                                                           \
            _Pragma("unroll") for (int e = 0; e < WIN; e += 2) mgf_ = fmaxf(mgf_, fmaxf(__builtin_fabsf(w_[e]), __builtin_fabsf(w_[e + 1]))); \
            /* reflect: the mirrored padding columns fold onto source columns 1 and W - 2 of the outermost copies (sums of two */ \
            /* elements: one more exponent step) */                                                                     \
            const bool refl_ = KS == 3 && ST == 1 && p.reflect;                                                         \
            const unsigned mg_ = group_max<NCH>(__builtin_bit_cast(unsigned, mgf_));                                    \
            Eg = max(Eg, min((int)(mg_ >> 23) + (refl_ ? 1 : 0), 254));                                                 \
            const float sg_ = pow2f(127 + TEXP + 127 - Eg);                                                             \
            /* element pairs (w[2i], w[2i + 1]): copies at an even window offset are runs of pairs, odd offsets one shift per word */ \
            unsigned ph_[(WIN + 1) / 2], pl_[(WIN + 1) / 2];                                                            \
            _Pragma("unroll") for (int i = 0; i < (WIN + 1) / 2; ++i) {                                                 \
                f16x2 h_, l_;                                                                                           \
                h_[0] = (_Float16)(w_[2 * i] * sg_); h_[1] = (_Float16)(w_[2 * i + 1] * sg_);                           \
                l_[0] = (_Float16)__builtin_fmaf(w_[2 * i], sg_, -(float)h_[0]);                                        \
                l_[1] = (_Float16)__builtin_fmaf(w_[2 * i + 1], sg_, -(float)h_[1]);                                    \
                ph_[i] = __builtin_bit_cast(unsigned, h_);                                                              \
                pl_[i] = __builtin_bit_cast(unsigned, l_);                                                              \
            }                                                                                                           \
            _Pragma("unroll") for (int d = DMIN; d <= DMAX; ++d) {                                                      \
                const int off_ = d - DMIN;                                                                              \
                u32x4 hi_, lo_;                                                                                         \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
                    if (off_ % 2 == 0) { hi_[i] = ph_[off_ / 2 + i]; lo_[i] = pl_[off_ / 2 + i]; }                      \
                    else {                                                                                              \
                        hi_[i] = (ph_[off_ / 2 + i] >> 16) | (ph_[off_ / 2 + i + 1] << 16);                             \
                        lo_[i] = (pl_[off_ / 2 + i] >> 16) | (pl_[off_ / 2 + i + 1] << 16);                             \
                    }                                                                                                   \
                }                                                                                                       \
                if (refl_) {                                                                                            \
                    /* copy d = +1 holds gy[x' + 1]: position x' = 1 (element 1 of chunk 0) also receives gy[0]; copy d = -1 */ \
                    /* holds gy[x' - 1]: position x' = W - 2 (element 6 of the last chunk) also receives gy[W - 1] */    \
                    if (d == 1 && jc_ == 0) {                                                                           \
                        const float v_ = w_[1 + d - DMIN] + w_[-DMIN];                                                  \
                        const _Float16 h_ = (_Float16)(v_ * sg_), l_ = (_Float16)__builtin_fmaf(v_, sg_, -(float)h_);   \
                        hi_[0] = (hi_[0] & 0xffffu) | ((unsigned)__builtin_bit_cast(unsigned short, h_) << 16);         \
                        lo_[0] = (lo_[0] & 0xffffu) | ((unsigned)__builtin_bit_cast(unsigned short, l_) << 16);         \
                    }                                                                                                   \
                    if (d == -1 && jc_ + 8 == p.W) {                                                                    \
                        const float v_ = w_[6 + d - DMIN] + w_[7 - DMIN];                                               \
                        const _Float16 h_ = (_Float16)(v_ * sg_), l_ = (_Float16)__builtin_fmaf(v_, sg_, -(float)h_);   \
                        hi_[3] = (hi_[3] & 0xffff0000u) | (unsigned)__builtin_bit_cast(unsigned short, h_);             \
                        lo_[3] = (lo_[3] & 0xffff0000u) | (unsigned)__builtin_bit_cast(unsigned short, l_);             \
                    }                                                                                                   \
                }                                                                                                       \
                if (!(p.dbg & 4)) { Gb_[((off_ * 2 + 0) * NCH + lch) * TP + lrow] = hi_;                                \
                Gb_[((off_ * 2 + 1) * NCH + lch) * TP + lrow] = lo_; }                                                  \
            }                                                                                                           \
            /* exponent bytes of the gy rows in accumulator-register order: byte 16 (k >> 2 & 1) + 4 (k >> 3) + (k & 3) of */ \
            /* each 32-row group */                                                                                     \
            if (lch == 0) {                                                                                             \
                const int kl_ = lrow & 31;                                                                              \
                reinterpret_cast<unsigned char*>(smem + EGBASE + (ctau & 1) * EW)[(lrow & ~31) + 16 * ((kl_ >> 2) & 1) + 4 * (kl_ >> 3) + (kl_ & 3)] = (unsigned char)Eg; \
            }                                                                                                           \
        }                                                                                                               \
        ++ctau;                                                                                                         \
    }

    // ---- MFMA role ----
    const int wk = TB == 64 ? wid >> 1 : 0, wc = TB == 64 ? wid & 1 : 0;
    f32x16 acc[NT];
#pragma unroll
    for (int tp = 0; tp < NT; ++tp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tp][e] = 0.f;
    u32x4 egR = {0u, 0u, 0u, 0u};                         // exponents the accumulator rows are scaled by (bytes, register order)
    unsigned exR[KS];                                      // ... and this lane's column, per filter row
#pragma unroll
    for (int r = 0; r < KS; ++r) exR[r] = 0;

    WG_LOAD(0)
    WG_LOAD(1)
    WG_LOAD(2)
    WG_LOAD(3)
    WG_CONVERT(0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    for (int g0 = 0; g0 < nticks; g0 += DEPTH) {
#pragma unroll
        for (int half = 0; half < DEPTH; ++half) {
            const int g = g0 + half;                       // (ticks beyond the last: loads clamped, nothing converted or multiplied)
            const int mtau = g;
            // the exponent bytes of this tick's operands first: their LDS latency hides behind the loader work below
            const int row0 = ST * (g - LAG);                                        // ring index of filter row 0's source row
            u32x4 egN = egR;
            unsigned exN[KS];
#pragma unroll
            for (int r = 0; r < KS; ++r) exN[r] = exR[r];
            if (mtau >= LAG && g < nticks) {
                egN = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(smem + EGBASE + (half & 1) * EW) + wk * 32 + lhi * 16);
#pragma unroll
                for (int r = 0; r < KS; ++r)
                    exN[r] = reinterpret_cast<const unsigned char*>(smem + EXBASE + ((row0 + r) % RING) * EW)[wc * 32 + l31];
            }
            // loads of tick g + DEPTH into the register set tick g used; conversion of tick g + 1
            WG_LOAD(half)
            if (g + 1 < nticks) WG_CONVERT((half + 1) % DEPTH)
            if (mtau >= LAG && g < nticks) {
                const u32x4* const Gb = smem + GBASE + (half & 1) * GW;             // (g & 1 == half & 1: DEPTH is even)
                {
                    bool chg = egN[0] != egR[0] || egN[1] != egR[1] || egN[2] != egR[2] || egN[3] != egR[3];
#pragma unroll
                    for (int r = 0; r < KS; ++r) chg = chg || exN[r] != exR[r];
                    if (!(p.dbg & 8) && __any(chg)) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int de = (int)((egR[e >> 2] >> (8 * (e & 3))) & 255u) - (int)((egN[e >> 2] >> (8 * (e & 3))) & 255u);
#pragma unroll
                            for (int r = 0; r < KS; ++r) {
                                const float f = pow2f(127 + de + (int)exR[r] - (int)exN[r]);
#pragma unroll
                                for (int sx = 0; sx < KS; ++sx) acc[r * KS + sx][e] *= f;
                            }
                        }
                        egR = egN;
#pragma unroll
                        for (int r = 0; r < KS; ++r) exR[r] = exN[r];
                    }
                }
                constexpr int KSTEPS = TB == 64 ? 2 : 1;      // MFMA k-steps this wave runs per tick (TB = 32: wave = k-step)
#pragma unroll
                for (int u = 0; u < ((p.dbg & 1) ? 0 : KSTEPS); ++u) {
                    const int ch = (TB == 64 ? 2 * u : 2 * wid) + lhi;
                    u32x4 af[ND][2], bf[KS][ST][2];
#pragma unroll
                    for (int d = 0; d < ND; ++d)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) af[d][pl] = Gb[((d * 2 + pl) * NCH + ch) * TP + wk * 32 + l31];
#pragma unroll
                    for (int r = 0; r < KS; ++r) {
                        const u32x4* const Xb = smem + XBASE + ((row0 + r) % RING) * XR;
#pragma unroll
                        for (int par = 0; par < ST; ++par)
#pragma unroll
                            for (int pl = 0; pl < 2; ++pl) bf[r][par][pl] = Xb[((par * 2 + pl) * NCH + ch) * TP + wc * 32 + l31];
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int r = 0; r < KS; ++r)          // (source rows outside the image were written as zeros)
#pragma unroll
                            for (int sx = 0; sx < KS; ++sx)
                                acc[r * KS + sx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    __builtin_bit_cast(f16x8, af[tap_d(sx, PADC, ST) - DMIN][q == 0 ? 1 : 0]),
                                    __builtin_bit_cast(f16x8, bf[r][tap_par(sx, PADC, ST)][q == 1 ? 1 : 0]), acc[r * KS + sx], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);           // this wave's LDS writes of tick g + 1 are done, its reads of tick g too
            if (!(p.dbg & 16)) __builtin_amdgcn_s_barrier();
        }
    }
#undef WG_LOAD
#undef WG_CONVERT

    // ---- epilogue: (TB = 32: the four waves' accumulators — all on the same row / column scales, every wave saw every tick's exponent
    // bytes — are added in wave order through LDS) unscale (exact powers of two), store the slab ----
    if (TB == 32) {
        float* const dump = reinterpret_cast<float*>(smem);            // NT * 16 * 64 floats
        for (int w = 1; w < 4; ++w) {
            if (wid == w) {
#pragma unroll
                for (int tp = 0; tp < NT; ++tp)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dump[(tp * 16 + e) * 64 + lane] = acc[tp][e];
            }
            __syncthreads();
            if (wid == 0) {
#pragma unroll
                for (int tp = 0; tp < NT; ++tp)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[tp][e] += dump[(tp * 16 + e) * 64 + lane];
            }
            __syncthreads();
        }
    }
    float* const out = p.part + (size_t)slab * ((size_t)p.K * p.C * NT);
    const int c = ct * TB + wc * 32 + l31;
    if (TB == 64 || wid == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = kt * TB + wk * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
            const float ug = pow2f((int)((egR[e >> 2] >> (8 * (e & 3))) & 255u) - TEXP);
            float* const o = out + (size_t)k * p.C + c;                  // slab layout [tap][k][c]: 128-byte runs per store
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                const float ux = pow2f((int)exR[r] - TEXP);
#pragma unroll
                for (int sx = 0; sx < KS; ++sx) o[(size_t)(r * KS + sx) * p.K * p.C] = (acc[r * KS + sx][e] * ug) * ux;
            }
        }
    }
    if (p.partb) {
        // bias partial of this slab: the NCH lanes of a row add up (fixed order), one value per row; only column tile 0 writes
        float b = bsum;
#pragma unroll
        for (int o = 1; o < NCH; o <<= 1) b += __shfl_xor(b, o, 64);
        if (ct == 0 && lch == 0) p.partb[(size_t)slab * p.K + kg] = b;
    }
}

// gw[k][c][tap] += sum over the slabs of part[slab][tap][k][c] (coalesced slab reads, one pass).  A workgroup = 64 (k, c) pairs of ONE
// tap (grid.y) x 4 waves; wave w adds slabs w, w + 4, ... in order (eight loads in flight), the four partial sums meet in LDS in wave
// order: a fixed association order.  (Round 6: one workgroup per 64 pairs ran all NT taps with one slab's loads in flight — 16
// workgroups and 64 dependent steps for the registration net's 32 x 32 layers: 17.7 us per call; the sums are bitwise what that form gave.)
// Workgroups beyond the (k, c) range (tap 0 only) add the bias partials: gb[k] += sum_slab partb[slab][k].
template <int NT>
__global__ __launch_bounds__(256) void s16g_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int KC, int nslab,
                                                                const float* __restrict__ partb, float* __restrict__ gb, int K) {
    __shared__ float red[3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nkc = (KC + 63) / 64;
    if ((int)blockIdx.x >= nkc) {                          // bias rows
        if (blockIdx.y != 0) return;
        const int k = ((int)blockIdx.x - nkc) * 64 + lane;
        float b = 0.f;
        if (k < K)
            for (int sl = w; sl < nslab; sl += 4) b += partb[(size_t)sl * K + k];
        if (w > 0) red[w - 1][lane] = b;
        __syncthreads();
        if (w == 0 && k < K) gb[k] += ((b + red[0][lane]) + red[1][lane]) + red[2][lane];
        return;
    }
    const int tp = blockIdx.y;
    const int i = blockIdx.x * 64 + lane;
    float a = 0.f;
    if (i < KC) {
        const float* const q = part + (size_t)tp * KC + i;
        const size_t ss = (size_t)NT * KC;
        for (int sl = w; sl < nslab; sl += 32) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = q[(size_t)min(sl + 4 * j, nslab - 1) * ss];      // unconditional: all eight in flight
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (sl + 4 * j < nslab) a += v[j];
        }
    }
    if (w > 0) red[w - 1][lane] = a;
    __syncthreads();
    if (w == 0 && i < KC) gw[(size_t)i * NT + tp] += ((a + red[0][lane]) + red[1][lane]) + red[2][lane];
}

}  // namespace

static int s16g_wgrad_tile(int C0, int C1, int K);

bool nemar_s16g_wgrad_eligible(int N, int C0, int C1, int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad,
                               int pad_mode) {
    const int C = C0 + C1;
    if (R != S || !((R == 3 && pad == 1) || (R == 1 && pad == 0 && stride == 1))) return false;
    if (stride != 1 && stride != 2) return false;
    if (pad_mode == 1 && !(R == 3 && stride == 1)) return false;         // reflect: the 3x3 stride-1 fold only
    if (K % 32 || C % 32 || C0 % 32) return false;
    if (W % (8 * stride) || OW % 8 || H < 4 || OW < 8 || OW != W / stride) return false;
    if ((W / stride) % (s16g_wgrad_tile(C0, C1, K) == 64 ? 32 : 64)) return false;      // a row is a whole number of steps
    if ((H + 2 * pad - R) / stride + 1 != OH || (W + 2 * pad - S) / stride + 1 != OW) return false;
    if ((long long)N * (C > K ? C : K) * H * W >= (1ll << 31)) return false;
    return true;
}

static int s16g_wgrad_tile(int C0, int C1, int K) { return (K % 64 == 0 && (C0 + C1) % 64 == 0 && C0 % 64 == 0) ? 64 : 32; }

// row blocks per image: a slab is (image, row block, column strip of one tick's width); ~256 workgroups in total (one resident round:
// every slab costs a write + a read of K C R S floats, and LAG warm-up ticks), at least 2 output rows per slab
static int s16g_wgrad_spi(int N, int C, int K, int OH, int TB, int spr) {
    const int tiles = (K / TB) * (C / TB);
    int spi = (256 + tiles * N * spr - 1) / (tiles * N * spr);
    if (spi > OH / 2) spi = OH / 2;
    if (spi < 1) spi = 1;
    const int rb = (OH + spi - 1) / spi;
    return (OH + rb - 1) / rb;
}

static int s16g_wgrad_spr(int W, int stride, int TB) { return (W / stride) / (TB == 64 ? 32 : 64); }

int nemar_s16g_wgrad_slabs(int N, int C0, int C1, int K, int OH, int W, int stride) {
    const int TB = s16g_wgrad_tile(C0, C1, K), spr = s16g_wgrad_spr(W, stride, TB);
    return N * s16g_wgrad_spi(N, C0 + C1, K, OH, TB, spr) * spr;
}

// upper bound over the channel splits C0 + C1 = C (the workspace query does not know the split)
int nemar_s16g_wgrad_slabs_max(int N, int C, int K, int OH, int W, int stride) {
    int m = 0;
    if ((W / stride) % 64 == 0) m = N * s16g_wgrad_spi(N, C, K, OH, 32, s16g_wgrad_spr(W, stride, 32)) * s16g_wgrad_spr(W, stride, 32);
    if (K % 64 == 0 && C % 64 == 0 && (W / stride) % 32 == 0) {
        const int a = N * s16g_wgrad_spi(N, C, K, OH, 64, s16g_wgrad_spr(W, stride, 64)) * s16g_wgrad_spr(W, stride, 64);
        m = a > m ? a : m;
    }
    return m;
}

// part: nemar_s16g_wgrad_slabs slabs of K C R S floats, then (gb) as many slabs of K floats
void nemar_s16g_wgrad(const float* x0, int C0, const float* x1, int C1, const float* gy, float* gw, float* gb, int N, int H, int W,
                      int K, int OH, int OW, int KS, int stride, int pad_mode, float* part, int dbg, hipStream_t st) {
    const int C = C0 + C1, TB = s16g_wgrad_tile(C0, C1, K);
    const int nslab = nemar_s16g_wgrad_slabs(N, C0, C1, K, OH, W, stride);
    WgGParams p;
    p.x0 = x0; p.x1 = x1; p.C0 = C0; p.C1 = C1; p.gy = gy;
    p.part = part;
    p.partb = gb ? part + (size_t)nslab * K * C * KS * KS : nullptr;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.OH = OH; p.OW = OW;
    p.reflect = pad_mode == 1 ? 1 : 0;
    p.pad = KS == 1 ? 0 : 1;
    p.spr = s16g_wgrad_spr(W, stride, TB);
    p.spi = s16g_wgrad_spi(N, C, K, OH, TB, p.spr);
    p.RB = (OH + p.spi - 1) / p.spi;
    p.KT = K / TB; p.CT = C / TB;
    p.dbg = dbg;
    const dim3 g(nslab * p.KT * p.CT), b(256);
    p.xcd = (g.x % 8 == 0 && (g.x / 8) % (p.KT * p.CT) == 0) ? 1 : 0;
#define WG_GO(KS_, ST_)                                                                          \
    {                                                                                            \
        if (TB == 64) hipLaunchKernelGGL((s16g_wgrad_kernel<KS_, ST_, 64>), g, b, 0, st, p);     \
        else hipLaunchKernelGGL((s16g_wgrad_kernel<KS_, ST_, 32>), g, b, 0, st, p);              \
    }
    if (KS == 3 && stride == 1) WG_GO(3, 1)
    else if (KS == 3) WG_GO(3, 2)
    else WG_GO(1, 1)
#undef WG_GO
    const int KC = K * C;
    const int rgrid = (KC + 63) / 64 + (gb ? (K + 63) / 64 : 0);
    if (KS == 3) hipLaunchKernelGGL((s16g_wgrad_reduce_kernel<9>), dim3(rgrid, 9), dim3(256), 0, st, (const float*)part, gw, KC, nslab,
                                    (const float*)p.partb, gb, K);
    else hipLaunchKernelGGL((s16g_wgrad_reduce_kernel<1>), dim3(rgrid, 1), dim3(256), 0, st, (const float*)part, gw, KC, nslab,
                            (const float*)p.partb, gb, K);
}
