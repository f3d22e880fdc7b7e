// nemar_amd — the 7x7 / stride-1 / pad-3 convolutions between FEW channels (<= 4) and MANY (a multiple of 32) on the 16-bit matrix
// pipe at fp32 accuracy: the translation net's stem  ReflectionPad2d(3) + Conv2d(3 -> 64, k7)  and head  ReflectionPad2d(3) +
// Conv2d(64 -> 3, k7) + Tanh  (reference models/networks.py:349-350, 375-377).  With 3 channels on one side the generic implicit GEMM
// (rows = output channels, reduction = taps x channels in 16-channel steps) wastes 13 of 16 reduction slots or 29 of 32 rows; these
// layers ran on VALU / exact-fp32 kernels at 23-50 TFLOP/s: 3.1 ms of a 36.6 ms step (profiles/r3_conv_layers.txt).
//
// Arithmetic: fp16 x 3 as in conv_s16g.hip — every fp32 operand v becomes v s = h + l + e, h = RN16(v s), l = RN16(v s - h),
// s a power of two, and the fp32 product is rebuilt from (l h') + (h l') + (h h') on v_mfma_f32_32x32x16_f16, accumulated in fp32.
// The split happens in the kernel, on the way into LDS.  Scales: the MANY-channel operand carries a running power-of-two scale per
// workgroup (raised, with an exact rescale of the accumulators, whenever a row strip holds a larger value: the flash-attention
// running maximum, as in conv_s16g.hip); the FEW-channel operand one scale per sample from a tiny max pass over its <= 4 planes.
//
// WEIGHT GRADIENT (both layers, one kernel):  G[m][c][dy][dx] = sum_{n,y,x} Big[n][m][y][x] * Small[n][c][y + dy][x + dx]
//   stem: Big = gy (64 channels), Small = x seen through its 3-pixel reflect / zero border;
//   head: Big = x through its border (64 channels, (H + 6) x (W + 6) positions), Small = gy through a 6-pixel ZERO border, and
//         gw[k][c][dy][dx] = G[c][k][6 - dy][6 - dx]   (substitute Y = y + dy, X = x + dx in the definition).
// As a GEMM the reduction runs over PIXELS — 8 consecutive pixels of a row are one MFMA operand word — rows = the 64 Big channels,
// columns = (c, dx) pairs, 4 x 8 = 32 per vertical tap dy.  The Small operand word of column (c, dx) is the row window starting
// dx pixels to the right: instead of unaligned LDS reads, every Small row is written into LDS as SEVEN SHIFTED COPIES (it has <= 4
// channels: 2-byte stores from the one thread that converted the element), so every operand read is an aligned 16-byte word.
// A workgroup owns (sample, 32-pixel column strip, block of rows): per Big row it stages the row strip (64 channels x 32 pixels,
// double-buffered) and one new Small row (ring of 8), then issues 2 m-tiles x 7 vertical taps x 2 k-steps x 3 products.  Partial
// results go to slabs (plain stores) summed in order: bitwise reproducible, like every other weight gradient of the library.
#include "common.h"
#include "conv_k7.h"
#include "pack_plan.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TEXP = 14;                // scaled magnitudes stay below 2^15
constexpr int MODE_ZERO = 0, MODE_REFLECT = 1;

__device__ __forceinline__ float pow2f(int biased) {          // 2^(biased - 127); 0 below the normal range
    return biased < 1 ? 0.f : __builtin_bit_cast(float, (unsigned)(biased > 254 ? 254 : biased) << 23);
}
__device__ __forceinline__ int max_exponent(unsigned maxbits) {
    int e = (int)(maxbits >> 23);
    if (e < TEXP + 2) e = TEXP + 2;
    if (e > 254) e = 254;
    return e;
}
// v s = h + l (+ e): both halves of eight values as two 16-byte words (see conv_s16g.hip split8)
__device__ __forceinline__ void split8(const float* v, float s, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f16x2 h, l;
        h[0] = (_Float16)(v[2 * j] * s);
        h[1] = (_Float16)(v[2 * j + 1] * s);
        l[0] = (_Float16)__builtin_fmaf(v[2 * j], s, -(float)h[0]);
        l[1] = (_Float16)__builtin_fmaf(v[2 * j + 1], s, -(float)h[1]);
        hi[j] = __builtin_bit_cast(unsigned, h);
        lo[j] = __builtin_bit_cast(unsigned, l);
    }
}
__device__ __forceinline__ unsigned wave_max_to_lane63(unsigned x) {
#ifdef NEMAR_HOST_EMULATION
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = max(x, (unsigned)__shfl_xor((int)x, o, 64));
    return x;
#else
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));      // row_shr:1
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));      // row_shr:2
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));      // row_shr:4
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));      // row_shr:8
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));      // row_bcast:15
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));      // row_bcast:31
    return x;
#endif
}
// largest FINITE magnitude of eight values as a bit pattern (non-finite elements do not take part: they must not flush the others)
__device__ __forceinline__ unsigned finite_max8(const float* v) {
    float mf = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) mf = fmaxf(mf, fmaxf(__builtin_fabsf(v[j]), __builtin_fabsf(v[j + 1])));
    unsigned m = __builtin_bit_cast(unsigned, mf);
    if (m >= 0x7f800000u) {
        m = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned u = __builtin_bit_cast(unsigned, v[j]) & 0x7fffffffu;
            m = max(m, u < 0x7f800000u ? u : 0u);
        }
    }
    return m;
}

// A tensor [N][C][H][W] seen through a border of `pad` texels on every side (zero or mirrored): view coordinate v -> source index
struct K7View {
    const float* p;
    int C, H, W, pad, mode;
};
__device__ __forceinline__ int view_index(int v, int pad, int n, int mode) {      // -1: a zero of the border / outside the view
    int i = v - pad;
    if (mode == MODE_REFLECT) {
        i = i < 0 ? -i : i;
        i = i >= n ? 2 * (n - 1) - i : i;
    }
    return (unsigned)i < (unsigned)n ? i : -1;
}

// max |t| over the finite elements of every SAMPLE of a small tensor (<= 4 planes), stage 1: SMAX_CHUNKS workgroups per sample, one
// partial word each (plain stores — no zero fill, no atomics); the consumer takes the maximum of a sample's partials itself
constexpr int SMAX_CHUNKS = 32;
__global__ __launch_bounds__(256) void k7_sample_max_kernel(const float* __restrict__ t, long long per, unsigned* __restrict__ out) {
    __shared__ unsigned red[4];
    const long long chunk = (per + SMAX_CHUNKS - 1) / SMAX_CHUNKS, lo = chunk * blockIdx.x, hi = min(per, lo + chunk);
    const float* s = t + (size_t)blockIdx.y * per;
    unsigned m = 0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const unsigned u = __builtin_bit_cast(unsigned, s[i]) & 0x7fffffffu;
        m = max(m, u < 0x7f800000u ? u : 0u);
    }
    m = wave_max_to_lane63(m);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.y * SMAX_CHUNKS + blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}

struct K7WgParams {
    K7View big, small;
    const unsigned* smax;          // [N][SMAX_CHUNKS] partial maxima of |small| per sample (bit patterns)
    float* part;                   // slabs
    long long slab_stride;
    int sm, sc, flip;              // slab element of (m, c, dy, dx): m sm + c sc + (flip ? 6 - dy : dy) 7 + (flip ? 6 - dx : dx)
    int N, Hv, Wv;                 // positions of the Big view the reduction runs over
    int strips, rblocks, RB, mblks;
    int bias_off;                  // >= 0: slab element where the per-channel sums of Big go (the bias gradient of the stem: Big = gy)
    int strip0, nstrips;           // this launch: strips [strip0, strip0 + nstrips) of `strips`
    int xoff;                      // strip s covers view columns [32 s - xoff, 32 s - xoff + 32): aligned to SOURCE columns of Big
};

// 32-pixel column strips, 4 waves.  Wave (mt, q): Big channels [32 mt, 32 mt + 32) of the workgroup's 64, vertical taps dy = q, q + 2, ...
// The Big operand never touches LDS: lane (channel l31, k half lhi) loads ITS OWN operand words — 8 consecutive pixels of its channel's
// row per k-step — straight from global memory (rows fetched four row-steps ahead through a register ring), splits them in registers
// under the WAVE's own running scale (no other wave reads them: no exchange, no barrier), and feeds the MFMAs.  The two waves that
// share a row tile (q = 0, 1) load the same rows (the second hit is in L1 / L2).  Only the Small operand is shared: its rows live
// in a 16-row LDS ring, staged FOUR ROWS per barrier.
// FAST: every column of the strip is an in-range, 16-byte-aligned source column (interior strips: strips are aligned to SOURCE columns
// of the Big tensor, view column = source column + pad); !FAST: per-element column table (edge strips, ragged widths).  CS: channels
// of the Small tensor.  Both compile-time: the loops below must be straight-line code for hipcc to keep the loads in flight (a select
// or branch right behind a load makes it wait on the spot — results are only touched a batch later).
template <bool FAST, bool EDGE, int CS>
__global__ __launch_bounds__(256, 2) void k7_wgrad_kernel(K7WgParams p) {
    // FAST strips carry an EDGE word (k word 4) for a mirrored border of 3 texels: positions 0..2 = the view columns left of the first
    // strip, 4..6 = right of the last one (their Big values are mirrors of columns the strip has loaded anyway), so that a reflect
    // border needs no strips of its own; the Small window is 3 columns wider on both sides for it
    constexpr int KW = 4, KWX = EDGE ? 5 : 4, SPX = 8 * KW, XL = EDGE ? 3 : 0, NSC = SPX + 6 + 2 * XL, RING = 14, BATCH = 4;
    constexpr int NSS = (BATCH * CS * NSC + 255) / 256;      // Small loader slots per thread and batch (4 rows x CS channels x NSC columns)
    __shared__ __attribute__((aligned(16))) u32x4 Sm[RING][2][KWX][32];       // [ring row][plane][k word][c * 8 + dx]
    __shared__ unsigned smax_s;
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    int b = blockIdx.x;
    const int mblk = b % p.mblks;
    b /= p.mblks;
    const int strip = p.strip0 + b % p.nstrips;
    b /= p.nstrips;
    const int rb = b % p.rblocks, n = b / p.rblocks;
    const int y0 = rb * p.RB, rows = min(p.RB, p.Hv - y0), xs = strip * SPX - p.xoff;
    const bool eL = EDGE && strip == 0, eR = EDGE && strip == p.strips - 1;      // (workgroup-uniform)
    {
        u32x4* z = &Sm[0][0][0][0];
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (int i = tid; i < RING * 2 * KWX * 32; i += 256) z[i] = zero;    // (c, dx) columns nobody writes stay zero
    }
    if (wid == 0) {                                          // this sample's max |small|: the maximum of its partial words
        unsigned m = lane < SMAX_CHUNKS ? p.smax[n * SMAX_CHUNKS + lane] : 0u;
        m = wave_max_to_lane63(m);
        if (lane == 63) smax_s = m;
    }
    // ---- Small loader: slot = (row of the batch, channel, column of the strip's window) ----
    int s_rci[NSS];                                          // row << 16 | channel << 8 | column, -1: no element
    unsigned s_off[NSS];                                     // element offset of (sample, channel, column) in the Small tensor (< 2^30: eligibility)
#pragma unroll
    for (int j = 0; j < NSS; ++j) {
        const int idx = tid + 256 * j, rr = idx / (CS * NSC), rem = idx - rr * (CS * NSC), c = rem / NSC, i = rem - c * NSC;
        const bool on = idx < BATCH * CS * NSC;
        const int col = on ? view_index(xs - XL + i, p.small.pad, p.small.W, p.small.mode) : -1;
        s_rci[j] = (on && col >= 0) ? ((rr << 16) | (c << 8) | i) : -1;
        s_off[j] = (unsigned)(n * CS + (on ? c : 0)) * (unsigned)(p.small.H * p.small.W) + (unsigned)(col >= 0 ? col : 0);
    }
    float sS = 0.f;
    float sv[NSS];
    auto small_load = [&](int R0) {                          // rows R0 .. R0 + 3: raw values (zeros are selected in small_store)
#pragma unroll
        for (int j = 0; j < NSS; ++j) {
            const int rr = s_rci[j] < 0 ? 0 : (s_rci[j] >> 16);
            const int ry = view_index(R0 + rr, p.small.pad, p.small.H, p.small.mode);
            sv[j] = p.small.p[s_off[j] + (unsigned)((ry < 0 ? 0 : ry) * p.small.W)];
        }
    };
    _Float16* const sm16 = (_Float16*)&Sm[0][0][0][0];
    auto small_store = [&](int R0) {                         // the seven shifted copies of every element, both planes
#pragma unroll
        for (int j = 0; j < NSS; ++j) {
            if (s_rci[j] >= 0) {
                const int rr = s_rci[j] >> 16, c = (s_rci[j] >> 8) & 0xff, i = s_rci[j] & 0xff;
                const float v = view_index(R0 + rr, p.small.pad, p.small.H, p.small.mode) >= 0 ? sv[j] : 0.f;
                const _Float16 h = (_Float16)(v * sS);
                const _Float16 l = (_Float16)__builtin_fmaf(v, sS, -(float)h);
                const int slot = (R0 + rr + 2 * RING) % RING;
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    const int xl = i - XL - dx;
                    if (xl >= 0 && xl < SPX) {
                        const int idx = ((((slot * 2) * KWX + (xl >> 3)) * 32) + c * 8 + dx) * 8 + (xl & 7);
                        sm16[idx] = h;
                        sm16[idx + KWX * 32 * 8] = l;
                    }
                    if (EDGE) {                              // edge word: view columns xs - 3 + j (left), xs + 32 + j (right), j = 0..2
                        const int jl = i - dx, jr = i - (SPX + XL) - dx;
                        if ((eL && jl >= 0 && jl < 3) || (eR && jr >= 0 && jr < 3)) {
                            const int pos = (eL && jl >= 0 && jl < 3) ? jl : 4 + jr;
                            const int idx = ((((slot * 2) * KWX + 4) * 32) + c * 8 + dx) * 8 + pos;
                            sm16[idx] = h;
                            sm16[idx + KWX * 32 * 8] = l;
                        }
                    }
                }
            }
        }
    };
    // ---- Big loader: lane = (channel, k half); per row 2 k-steps x 8 pixels ----
    const int mc = mblk * 64 + (wid & 1) * 32 + l31;
    const bool b_ch = mc < p.big.C;
    const float* const b_base = p.big.p + ((size_t)n * p.big.C + (b_ch ? mc : 0)) * p.big.H * p.big.W;
    const int src0 = xs - p.big.pad;                         // source column of the strip's first pixel
    int bcol[FAST ? 1 : 16];                                 // !FAST: source column per element, -1 = zero
    if (!FAST) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int X = xs + 8 * (2 * (e >> 3) + lhi) + (e & 7);
            bcol[FAST ? 0 : e] = (X >= 0 && X < p.Wv) ? view_index(X, p.big.pad, p.big.W, p.big.mode) : -1;
        }
    }
    constexpr int PF = 4;
    float bq[PF][16];
    auto big_load = [&](float* bv, int Y) {                  // raw values; big_mask() selects the zeros when the row is consumed
        const int ry = Y < p.Hv ? view_index(Y, p.big.pad, p.big.H, p.big.mode) : -1;
        const float* r = b_base + (size_t)(ry < 0 ? 0 : ry) * p.big.W;
        if (FAST) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float4 a = *(const float4*)(r + src0 + 8 * (2 * s + lhi)), c = *(const float4*)(r + src0 + 8 * (2 * s + lhi) + 4);
                float* o = bv + 8 * s;
                o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = c.x; o[5] = c.y; o[6] = c.z; o[7] = c.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) bv[e] = r[bcol[FAST ? 0 : e] < 0 ? 0 : bcol[FAST ? 0 : e]];
        }
    };
    auto big_mask = [&](float* bv, int Y) {
        const bool ok = b_ch && Y < p.Hv && view_index(Y, p.big.pad, p.big.H, p.big.mode) >= 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] = (ok && (FAST || bcol[FAST ? 0 : e] >= 0)) ? bv[e] : 0.f;
    };

    const int mt = wid & 1, q = wid >> 1;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // prologue: Small rows y0 .. y0 + 9 into the ring (what the first batch reads), rows y0 + 10 .. 13 in flight; the first four Big
    // rows in flight.  Ring invariant at the top of batch k: rows 4k .. 4k + 9 present, rows 4k + 10 .. 13 in registers — they go to
    // the slots of rows 4k - 4 .. 4k - 1 (14 slots: the ten live rows + the four being written).
#pragma unroll
    for (int i = 0; i < PF; ++i) big_load(bq[i], y0 + i);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                            // the zero fill is complete, smax_s is written
    sS = uniform_f(pow2f(127 + TEXP + 127 - max_exponent(smax_s)));
    small_load(y0 - 2);                                      // rows y0 - 2 .. y0 + 1: the first two are overwritten right below
    small_store(y0 - 2);
    small_load(y0 + 2);
    small_store(y0 + 2);
    small_load(y0 + 6);
    small_store(y0 + 6);                                     // ... rows y0 + 6 .. 9 (rows y0 - 2, - 1 sat in the slots of y0 + 12, 13)
    small_load(y0 + 10);
    int E = 0;                                               // this WAVE's running exponent of the Big operand
    float bsum = 0.f;                                        // bias gradient (stem: Big = gy): this lane's channel, its 16 pixels of every row
    // (whole batches: rows beyond the block / the view are zeros — big_mask — so the loop body has no exit in the middle)
    for (int r0 = 0; r0 < rows; r0 += BATCH) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                        // ring rows < y0 + r0 + 10 are visible; every wave has left the previous batch
        small_store(y0 + r0 + 10);                           // into the slots of rows y0 + r0 - 4 .. - 1: dead
        small_load(y0 + r0 + 14);
#pragma unroll
        for (int ri = 0; ri < BATCH; ++ri) {
            float* const bv = bq[ri];
            const int Y = y0 + r0 + ri;
            big_mask(bv, r0 + ri < rows ? Y : p.Hv);
            if (!EDGE && p.bias_off >= 0) {          // (the bias sums belong to the stem: Big = gy without a border, never an EDGE launch)
                float t = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) t += bv[e];
                bsum += t;
            }
            // the wave's running scale: raised (accumulators rescaled by the exact power of two) when this row holds a larger value
            {
                unsigned m = wave_max_to_lane63(max(finite_max8(bv), finite_max8(bv + 8)));
#ifdef NEMAR_HOST_EMULATION
                m = (unsigned)__shfl((int)m, 63, 64);
#else
                m = (unsigned)__builtin_amdgcn_readlane((int)m, 63);
#endif
                const int e = max_exponent(m);
                if (e > E) {
                    const float f = E ? pow2f(127 + E - e) : 0.f;   // (nothing accumulated yet: E == 0)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) acc[i][kk] *= f;
                    E = e;
                }
            }
            u32x4 ah[2], al[2];
            {
                const float sc = pow2f(127 + TEXP + 127 - E);
#pragma unroll
                for (int s = 0; s < 2; ++s) split8(bv + 8 * s, sc, ah[s], al[s]);
            }
            u32x4 eh = {0u, 0u, 0u, 0u}, el = eh;            // edge word: [x(3), x(2), x(1), 0 | x(W-2), x(W-3), x(W-4), 0] of this channel's row
            if (EDGE && (eL || eR)) {
                // left mirrors: elements 3, 2, 1 of k word 0 (this lane's own when lhi == 0); right mirrors: elements 6, 5, 4 of k word 3,
                // held by the lhi == 1 lane of the same channel
                float ev[8];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float r = __shfl_xor(bv[8 + 6 - j], 32, 64);
                    ev[j] = (eL && !lhi) ? bv[3 - j] : 0.f;
                    ev[4 + j] = (eR && !lhi) ? r : 0.f;
                }
                ev[3] = 0.f;
                ev[7] = 0.f;
                split8(ev, pow2f(127 + TEXP + 127 - E), eh, el);
            }
            big_load(bv, Y + PF);                            // this slot's next occupant: in flight during the next four row-steps
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int dy = q + 2 * i;
                    if (dy < 7) {
                        const int slot = (Y + dy) % RING;
                        const u32x4 bh = Sm[slot][0][2 * s + lhi][l31], bl = Sm[slot][1][2 * s + lhi][l31];
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[s]), __builtin_bit_cast(f16x8, bh), acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s]), __builtin_bit_cast(f16x8, bl), acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s]), __builtin_bit_cast(f16x8, bh), acc[i], 0, 0, 0);
                    }
                }
            }
            if (EDGE && (eL || eR)) {                        // the mirrored border columns: one more k-step, operand words held by the lhi = 0 half
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int dy = q + 2 * i;
                    if (dy < 7) {
                        const int slot = (Y + dy) % RING;
                        const u32x4 zero = {0u, 0u, 0u, 0u};
                        const u32x4 bh = lhi ? zero : Sm[slot][0][EDGE ? 4 : 0][l31], bl = lhi ? zero : Sm[slot][1][EDGE ? 4 : 0][l31];
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, el), __builtin_bit_cast(f16x8, bh), acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, eh), __builtin_bit_cast(f16x8, bl), acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, eh), __builtin_bit_cast(f16x8, bh), acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- epilogue: the two power-of-two scales out (exact), this workgroup's slab ----
    const float u1 = pow2f(E - TEXP), u2 = pow2f(max_exponent(smax_s) - TEXP);
    float* const slab = p.part + (size_t)((n * p.rblocks + rb) * p.strips + strip) * p.slab_stride;
    if (!EDGE && p.bias_off >= 0 && q == 0) {                // (the q == 1 wave loaded the same rows)
        bsum += __shfl_xor(bsum, 32, 64);
        if (!lhi && mc < p.big.C) slab[p.bias_off + mc] = bsum;
    }
    const int c = l31 >> 3, dx = l31 & 7;
    if (c < CS && dx < 7) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int dy = q + 2 * i;
            if (dy >= 7) continue;
            const int tap = (p.flip ? 6 - dy : dy) * 7 + (p.flip ? 6 - dx : dx);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int m = mblk * 64 + mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * lhi;
                if (m < p.big.C) slab[(size_t)m * p.sm + (size_t)c * p.sc + tap] = (acc[i][k] * u1) * u2;
            }
        }
    }
}

constexpr int WG_KW = 4;

// ---------------------------------------------------------------------------------------------------------------------------------
// FEW -> MANY:  out[n][m][y][x] = bias[m] + sum_{c < Cs, dy, dx} Wt[m][c][dy][dx] * Small[n][c][y + dy][x + dx]
//   stem forward: Small = x through its 3-pixel border, Wt = w;   head data gradient: Small = gy through a ZERO border (3 pixels
//   for a zero-padded layer; 6 for a reflect one, whose gradient is taken on the padded (H + 6) x (W + 6) domain and folded back
//   by the caller), Wt[c][k][dy][dx] = w[k][c][6 - dy][6 - dx].
// GEMM view: rows = the many channels, columns = pixels, reduction = the (c, dy) pairs (7 Cs <= 28, in 8-element words: 2 k-steps
// of 16) for each of the 7 horizontal taps dx.  The operand word of a pixel holds src[c][y + dy][x'] for eight (c, dy) pairs: a
// ROW-EXPANDED copy of the <= 4-channel source, built in LDS in two phases — (A) every halo element is loaded once, converted once
// and parked as h | l << 16; (B) the thread of (halo column, tile row) gathers its 7 Cs elements and writes whole 16-byte operand
// words, conflict-free — so that a horizontal tap is a shift by whole words: every MFMA operand read is aligned.  The weights of a
// wave's 32 rows (7 taps x 2 k-steps x 2 planes = 28 words per lane) stay in REGISTERS for the whole kernel: workgroups are
// persistent over pixel tiles (4 rows x 64 columns), the tap loop reads only the source operand from LDS.  Scale of the source: per
// TILE (its whole receptive field is in the tile's halo: no running rescale), from a max over the values just loaded.
constexpr int FM_RT = 4, FM_TW = 64, FM_HC = FM_TW + 6, FM_HCP = 72, FM_G = 4;

struct K7FmParams {
    K7View small;
    const u32x4* wp;
    const unsigned* wmax;
    const float* bias;
    float* dst;
    int M, N, Hv, Wv, act;
    float slope;
    int tiles_x, tiles_y, mblks;
};


// pack (one workgroup: the tensors have <= 4 x 49 x M elements): max |w|, then the scaled hi / lo operand words.  Also a job type of the
// weight-pack plans (pack_plan.h).
struct K7PackArgs {
    const float* w; long long wsm, wsc; int flip, M, Cs; u32x4* out; unsigned* maxword;
    int gx, gy;
};
__device__ __forceinline__ void k7_fm_pack_body(const K7PackArgs& a, int, int, int) {
    const float* __restrict__ w = a.w;
    const long long wsm = a.wsm, wsc = a.wsc;
    const int flip = a.flip, M = a.M, Cs = a.Cs;
    u32x4* __restrict__ out = a.out;
    unsigned* __restrict__ maxword = a.maxword;
    __shared__ unsigned red[16];
    __shared__ unsigned mx;
    const int total = M * Cs * 49;
    unsigned m = 0;
    for (int i = threadIdx.x; i < total; i += 1024) {
        const int mm = i / (Cs * 49), r = i - mm * (Cs * 49), c = r / 49, t = r - c * 49;
        const unsigned u = __builtin_bit_cast(unsigned, w[mm * wsm + c * wsc + t]) & 0x7fffffffu;
        m = max(m, u < 0x7f800000u ? u : 0u);
    }
    m = wave_max_to_lane63(m);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned r = 0;
        for (int i = 0; i < 16; ++i) r = max(r, red[i]);
        mx = r;
        maxword[0] = r;
    }
    __syncthreads();
    const float scale = pow2f(127 + TEXP + 127 - max_exponent(mx));
    const int mblks = (M + 63) / 64, words = mblks * 2 * 7 * 2 * 2 * 64;
    for (int i = threadIdx.x; i < words; i += 1024) {
        const int lane = i & 63;
        int t = i >> 6;
        const int pl = t & 1; t >>= 1;
        const int s = t & 1; t >>= 1;
        const int dx = t % 7; t /= 7;
        const int mt = t & 1, mblk = t >> 1;
        const int mm = mblk * 64 + mt * 32 + (lane & 31), g = 2 * s + (lane >> 5);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = 8 * g + j, c = e / 7, dy = e - c * 7;
            v[j] = (mm < M && c < Cs) ? w[mm * wsm + c * wsc + (flip ? (6 - dy) * 7 + (6 - dx) : dy * 7 + dx)] : 0.f;
        }
        u32x4 hi, lo;
        split8(v, scale, hi, lo);
        out[i] = pl ? lo : hi;
    }
}
__global__ __launch_bounds__(1024) void k7_fm_pack_kernel(K7PackArgs a) { k7_fm_pack_body(a, 0, 0, 1); }
NEMAR_PACK_MULTI(k7_fm_pack_multi_kernel, K7PackArgs, k7_fm_pack_body, 1024)
void k7_fm_pack_multi(const void* jobs, int njobs, int gx, int gy, hipStream_t st) {
    hipLaunchKernelGGL(k7_fm_pack_multi_kernel, dim3(gx, gy, njobs), dim3(1024), 0, st, (const K7PackArgs*)jobs);
}
struct RegK7Pack {
    RegK7Pack() { nemar_pack_register(PACK_FAM_K7, sizeof(K7PackArgs), k7_fm_pack_multi); }
} g_reg_k7_pack;

// FOLD (the data gradient of a REFLECT-padded layer, W % 64 == 0, H % 4 == 0, H >= 8): the tile is a tile of the IMAGE; the gradient of
// the padded input at every padded position that mirrors onto one of its pixels is accumulated into the SAME accumulator — more taps
// over other rows / columns of the row-expanded source, no padded-domain tensor, no fold pass.  Padded row of image row y: y + 3, plus
// 3 - y (1 <= y <= 3) / 2 (H - 1) - y + 3 (H - 4 <= y <= H - 2): the first / last tile stage 7 rows (padded rows 0..6 / H-1..H+5) instead
// of 4; columns likewise through per-lane operand addresses (lanes without a mirror read a column that stays zero).
template <int Cs, bool FOLD>
__global__ __launch_bounds__(256, 2) void k7_fm_kernel(K7FmParams p) {
    constexpr int NRB = FOLD ? 7 : FM_RT, NR = NRB + 6;      // operand rows / halo rows a tile may stage
    constexpr int HC = FOLD ? 76 : FM_HC, HCP = FOLD ? 84 : FM_HCP, GW = (7 * Cs + 7) / 8;      // valid / allocated columns (FOLD: 76..83 stay zero)
    __shared__ __attribute__((aligned(16))) u32x4 Bs[2][GW][NRB][HCP];
    __shared__ unsigned Raw[Cs][NR][HCP];
    __shared__ unsigned red[4];
    __shared__ float bias_s[64];
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int mblk = blockIdx.x % p.mblks, mt = wid & 1, half = wid >> 1;
    const bool full = mblk * 64 + mt * 32 + 32 <= p.M;
    if (tid < 64) bias_s[tid] = (p.bias && mblk * 64 + tid < p.M) ? p.bias[mblk * 64 + tid < p.M ? mblk * 64 + tid : 0] : 0.f;

    u32x4 wa[7][2][2];
#pragma unroll
    for (int dx = 0; dx < 7; ++dx)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wa[dx][s][pl] = p.wp[((((size_t)(mblk * 2 + mt) * 7 + dx) * 2 + s) * 2 + pl) * 64 + lane];
    const float u2 = pow2f(max_exponent(p.wmax[0]) - TEXP);
    const int ntiles = p.N * p.tiles_y * p.tiles_x, tstep = (int)gridDim.x / p.mblks;
    const size_t splane = (size_t)p.small.H * p.small.W, dplane = (size_t)p.Hv * p.Wv;

    constexpr int NSL = (Cs * NR * HCP + 255) / 256, nsl = NSL;
    if (FOLD) {                                              // the always-zero columns
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (int i = tid; i < 2 * GW * NRB * (HCP - HC); i += 256) {
            const int col = HC + i % (HCP - HC), rest = i / (HCP - HC);
            Bs[rest / (GW * NRB)][(rest / NRB) % GW][rest % NRB][col] = zero;
        }
    }
    // loader slot i of this thread = element (c, halo row r, halo column xi) of the halo; recomputed where needed (constant divisors: a
    // few multiplies) instead of held in NSL registers — with the 112 weight registers the kernel sits at the 256-register limit
#define K7_SC(i_) ((tid + 256 * (i_)) / (NR * HCP))
#define K7_SR(i_) (((tid + 256 * (i_)) % (NR * HCP)) / HCP)
#define K7_SX(i_) ((tid + 256 * (i_)) % HCP)
    // the halo of a tile, one element per slot — UNCONDITIONAL loads from clamped addresses (a conditional load, or a select right
    // behind a load, makes hipcc wait for every load separately), zeros selected when the values are consumed.  Issued one tile AHEAD.
    float v[NSL];
    // view row of operand row 0 of tile row ty: the tile's first row; FOLD: padded rows y0 + 3 .. (the first tile starts at padded row 0)
    auto ybase_of = [&](int ty) { return FOLD ? (ty == 0 ? 0 : ty * FM_RT + 3) : ty * FM_RT; };
    auto issue_loads = [&](int t) {
        const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n = t / (p.tiles_x * p.tiles_y);
        const float* const sbase = p.small.p + (size_t)n * Cs * splane;
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            if (i < nsl) {
                const int c = K7_SC(i) < Cs ? K7_SC(i) : Cs - 1;
                const int ry = view_index(ybase_of(ty) + K7_SR(i), p.small.pad, p.small.H, p.small.mode);
                const int cx = view_index(tx * FM_TW + K7_SX(i), p.small.pad, p.small.W, p.small.mode);
                v[i] = sbase[(size_t)c * splane + (size_t)(ry < 0 ? 0 : ry) * p.small.W + (cx < 0 ? 0 : cx)];
            }
        }
    };
    const int t_first = (int)blockIdx.x / p.mblks;
    if (t_first < ntiles) issue_loads(t_first);
    for (int t = t_first; t < ntiles; t += tstep) {
        const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n = t / (p.tiles_x * p.tiles_y);
        const int y0 = ty * FM_RT, x0 = tx * FM_TW, Yb = ybase_of(ty);
        const bool top = FOLD && ty == 0, bot = FOLD && ty == p.tiles_y - 1;      // tiles with mirrored rows: 7 operand rows
        const int nrb = (top || bot) ? NRB : FM_RT;
        unsigned mloc = 0;
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            if (i < nsl) {
                const int ry = view_index(Yb + K7_SR(i), p.small.pad, p.small.H, p.small.mode);
                const int cx = view_index(x0 + K7_SX(i), p.small.pad, p.small.W, p.small.mode);
                v[i] = (K7_SC(i) < Cs && K7_SX(i) < HC && ry >= 0 && cx >= 0) ? v[i] : 0.f;
                const unsigned u = __builtin_bit_cast(unsigned, v[i]) & 0x7fffffffu;
                mloc = max(mloc, u < 0x7f800000u ? u : 0u);
            }
        }
        mloc = wave_max_to_lane63(mloc);
        if (lane == 63) red[wid] = mloc;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        const int E = max_exponent(max(max(red[0], red[1]), max(red[2], red[3])));
        const float scale = pow2f(127 + TEXP + 127 - E);
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            if (i < nsl && K7_SC(i) < Cs) {
                const _Float16 h = (_Float16)(v[i] * scale);
                const _Float16 l = (_Float16)__builtin_fmaf(v[i], scale, -(float)h);
                f16x2 hl;
                hl[0] = h;
                hl[1] = l;
                Raw[K7_SC(i)][K7_SR(i)][K7_SX(i)] = __builtin_bit_cast(unsigned, hl);
            }
        }
        if (t + tstep < ntiles) issue_loads(t + tstep);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();

#pragma unroll
        for (int pass = 0; pass < (NRB * HC + 255) / 256; ++pass) {
            const int q = tid + 256 * pass;
            if (q < nrb * HC) {
                const int y = q / HC, xi = q - y * HC;
#pragma unroll
                for (int g = 0; g < GW; ++g) {
                    {
                        unsigned reg[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int e = 8 * g + j, c = e / 7, dy = e - c * 7;
                            const unsigned w = Raw[c < Cs ? c : 0][y + dy][xi];
                            reg[j] = c < Cs ? w : 0u;
                        }
                        u32x4 hi, lo;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            hi[j] = (reg[2 * j] & 0xffffu) | (reg[2 * j + 1] << 16);
                            lo[j] = (reg[2 * j] >> 16) | (reg[2 * j + 1] & 0xffff0000u);
                        }
                        Bs[0][g][y][xi] = hi;
                        Bs[1][g][y][xi] = lo;
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();

        const float u1 = pow2f(E - TEXP);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int yl = 2 * half + pr;
            f32x16 acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            // operand rows / columns of this wave's pixels: the pixel's own padded position, and (FOLD) its mirrors
            const int rm = FOLD ? yl + 3 - (Yb - y0) : yl;                              // main row (Yb - y0 = 3, or 0 in the first tile)
            const int rr = top ? (yl >= 1 ? 3 - yl : -1) : (bot ? (yl <= 2 ? 6 - yl : -1) : -1);      // mirrored row or none (wave-uniform)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int xl = 32 * i + l31, cm = FOLD ? xl + 3 : xl;                   // main column
                int cx = HC;                                                            // mirrored column (a zero column: none)
                bool anyc = false;                                                      // (wave-uniform: this pixel tile has border columns)
                if (FOLD) {
                    if (x0 == 0 && i == 0) { anyc = true; if (xl >= 1 && xl <= 3) cx = 3 - xl; }
                    if (x0 + FM_TW == p.Wv && i == 1) { anyc = true; if (xl >= 60 && xl <= 62) cx = 2 * 63 - xl + 3; }
                }
#pragma unroll
                for (int var = 0; var < (FOLD ? 4 : 1); ++var) {
                    const int row = (var & 1) ? rr : rm;
                    if ((var & 1) && rr < 0) continue;
                    if ((var & 2) && !anyc) continue;
                    const int col = (var & 2) ? cx : cm;
#pragma unroll
                    for (int dx = 0; dx < 7; ++dx)
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            if (2 * s >= GW) continue;
                            const u32x4 zero = {0u, 0u, 0u, 0u};
                            const bool on = 2 * s + lhi < GW;                           // (Cs <= 2: the second k word does not exist)
                            const u32x4 bh = on ? Bs[0][on ? 2 * s + lhi : 0][row][col + dx] : zero, bl = on ? Bs[1][on ? 2 * s + lhi : 0][row][col + dx] : zero;
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[dx][s][1]), __builtin_bit_cast(f16x8, bh), acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[dx][s][0]), __builtin_bit_cast(f16x8, bl), acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[dx][s][0]), __builtin_bit_cast(f16x8, bh), acc[i], 0, 0, 0);
                        }
                }
            }

            const float u12 = u1 * u2;
            const bool one_mul = u12 != 0.f && u12 < 3.0e38f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int y = y0 + yl, x = x0 + 32 * i + l31;
                if (y >= p.Hv || x >= p.Wv) continue;
                const int m0 = mblk * 64 + mt * 32 + 4 * lhi;
                float* const d0 = p.dst + ((size_t)n * p.M + m0) * dplane + (size_t)y * p.Wv + x;
                // the activation is chosen once per wave: three compact store loops (a per-element switch costs more than the stores)
#define K7_STORES(EXPR_)                                                                                                 \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                          \
                    const int mo = (r & 3) + 8 * (r >> 2);                                                                \
                    if (full || m0 + mo < p.M) {                                                                          \
                        const float o_ = (one_mul ? acc[i][r] * u12 : (acc[i][r] * u1) * u2) + bias_s[mt * 32 + 4 * lhi + mo]; \
                        d0[(size_t)mo * dplane] = (EXPR_);                                                                \
                    }                                                                                                     \
                }
                if (p.act == 1) { K7_STORES(fmaxf(o_, 0.f)) }
                else if (p.act == 2) { K7_STORES(o_ > 0.f ? o_ : o_ * p.slope) }
                else { K7_STORES(o_) }
#undef K7_STORES
            }
        }
    }
#undef K7_SC
#undef K7_SR
#undef K7_SX
}


// Strip geometry.  W % 32 == 0 (every shape of the translation net): W / 32 strips aligned to the SOURCE columns of the Big tensor (view
// column = source column + pad), all on the FAST instantiation — a mirrored border rides along as the edge word of the first / last strip,
// a zero border contributes nothing.  Otherwise: 32-column strips over the whole view on the general instantiation.
struct K7WgGeom { int Hv, Wv, strips, rblocks, RB, mblks; bool aligned; };
K7WgGeom k7_wg_geom(int N, int C, int H, int W, int K) {
    K7WgGeom g;
    const bool stem = C <= 4;
    g.Hv = stem ? H : H + 6;
    g.Wv = stem ? W : W + 6;
    g.aligned = W % 32 == 0;
    g.strips = g.aligned ? W / 32 : nemar_cdiv(g.Wv, 8 * WG_KW);
    g.mblks = nemar_cdiv(stem ? K : C, 64);
    // row blocks: ~512 workgroups (two per CU, one round), whole batches of four rows, at least 8 rows each
    const long long cols = (long long)N * g.strips * g.mblks;
    int rblocks = (int)((512 + cols - 1) / cols);
    if (rblocks < 1) rblocks = 1;
    g.RB = (nemar_cdiv(g.Hv, rblocks) + 3) & ~3;
    if (g.RB < 8) g.RB = 8;
    g.rblocks = nemar_cdiv(g.Hv, g.RB);
    return g;
}

}  // namespace

bool nemar_k7_wgrad_eligible(int N, int C, int H, int W, int K, int R, int S, int stride, int pad) {
    if (R != 7 || S != 7 || stride != 1 || pad != 3 || N <= 0 || H < 4 || W < 4) return false;
    const bool stem = C <= 4 && K % 32 == 0, head = K <= 4 && C % 32 == 0;
    return (stem || head) && (long long)N * (H + 12) * (W + 12) < (1ll << 30);
}

int nemar_k7_wgrad_slabs(int N, int C, int H, int W, int K) {
    const K7WgGeom g = k7_wg_geom(N, C, H, W, K);
    return N * g.rblocks * g.strips;
}

// slab = K C 49 weight-gradient partials (+ K bias partials for the stem, whose Big operand IS gy)
size_t nemar_k7_wgrad_floats(int N, int C, int H, int W, int K) {
    return (size_t)nemar_k7_wgrad_slabs(N, C, H, W, K) * ((size_t)K * C * 49 + (C <= 4 ? K : 0)) + (size_t)N * SMAX_CHUNKS;
}

// -> true: gb (when given) has been accumulated too (stem); false: the caller runs its own bias reduction (head: K <= 4 planes of gy)
bool nemar_k7_wgrad(const float* x, const float* gy, float* gw, float* gb, int N, int C, int H, int W, int K, int pad_mode, float* part,
                    hipStream_t st) {
    const bool stem = C <= 4;
    const K7WgGeom g = k7_wg_geom(N, C, H, W, K);
    K7WgParams p;
    const long long J = (long long)K * C * 49, stride = J + (stem ? K : 0);
    const int slabs = N * g.rblocks * g.strips;
    unsigned* smax = (unsigned*)(part + (size_t)slabs * stride);
    if (stem) {
        p.big = K7View{gy, K, H, W, 0, MODE_ZERO};
        p.small = K7View{x, C, H, W, 3, pad_mode ? MODE_REFLECT : MODE_ZERO};
        p.sm = C * 49; p.sc = 49; p.flip = 0;
    } else {
        p.big = K7View{x, C, H, W, 3, pad_mode ? MODE_REFLECT : MODE_ZERO};
        p.small = K7View{gy, K, H, W, 6, MODE_ZERO};
        p.sm = 49; p.sc = C * 49; p.flip = 1;
    }
    p.Hv = g.Hv; p.Wv = g.Wv;
    hipLaunchKernelGGL(k7_sample_max_kernel, dim3(SMAX_CHUNKS, N), dim3(256), 0, st, p.small.p, (long long)p.small.C * H * W, smax);
    p.smax = smax;
    p.part = part; p.slab_stride = stride;
    p.bias_off = (stem && gb) ? (int)J : -1;
    p.N = N;
    p.xoff = g.aligned ? -p.big.pad : 0;
    p.strips = g.strips; p.RB = g.RB; p.rblocks = g.rblocks; p.mblks = g.mblks;
    p.strip0 = 0; p.nstrips = p.strips;
    const bool edge = g.aligned && p.big.pad == 3 && p.big.mode == MODE_REFLECT;
    const dim3 grid(p.mblks * p.strips * p.rblocks * N), block(256);
#define K7_WG(F_, E_, C_) hipLaunchKernelGGL((k7_wgrad_kernel<F_, E_, C_>), grid, block, 0, st, p)
#define K7_WGC(F_, E_) { if (cs == 1) K7_WG(F_, E_, 1); else if (cs == 2) K7_WG(F_, E_, 2); else if (cs == 3) K7_WG(F_, E_, 3); else K7_WG(F_, E_, 4); }
    const int cs = p.small.C;
    if (!g.aligned) K7_WGC(false, false)
    else if (edge) K7_WGC(true, true)
    else K7_WGC(true, false)
#undef K7_WGC
#undef K7_WG
    nemar_sum_partials_pair(part, stride, slabs, gw, J, p.bias_off >= 0 ? part + J : nullptr, stride, slabs, gb, K, true, st);
    return p.bias_off >= 0 || !gb;
}

// ---- few -> many (stem forward, head data gradient) ----
bool nemar_k7_fm_eligible(int Cs, int M, int R, int S, int stride, int pad) {
    return R == 7 && S == 7 && stride == 1 && pad == 3 && Cs >= 1 && Cs <= 4 && M >= 32 && M % 32 == 0;
}

size_t nemar_k7_fm_pack_floats(int M) { return (size_t)nemar_cdiv(M, 64) * 2 * 7 * 2 * 2 * 64 * 4 + 4; }

void nemar_k7_fm_pack(const float* w, long long wsm, long long wsc, int flip, int M, int Cs, void* packed, hipStream_t st) {
    unsigned* maxword = (unsigned*)((float*)packed + nemar_k7_fm_pack_floats(M) - 4);
    K7PackArgs a{w, wsm, wsc, flip, M, Cs, (u32x4*)packed, maxword, 1, 1};
    if (nemar_pack_recording()) nemar_pack_record_job(PACK_FAM_K7, &a, 1, 1);
    hipLaunchKernelGGL(k7_fm_pack_kernel, dim3(1), dim3(1024), 0, st, a);
}

void nemar_k7_fm_conv(const float* src, int Cs, int Hs, int Ws, int pad, int reflect, const void* packed, const float* bias, float* dst,
                      int M, int N, int Hv, int Wv, int act, float slope, int fold, hipStream_t st) {
    K7FmParams p;
    p.small = K7View{src, Cs, Hs, Ws, pad, reflect ? MODE_REFLECT : MODE_ZERO};
    p.wp = (const u32x4*)packed;
    p.wmax = (const unsigned*)((const float*)packed + nemar_k7_fm_pack_floats(M) - 4);
    p.bias = bias; p.dst = dst; p.M = M; p.N = N; p.Hv = Hv; p.Wv = Wv; p.act = act; p.slope = slope;
    p.tiles_x = nemar_cdiv(Wv, FM_TW); p.tiles_y = nemar_cdiv(Hv, FM_RT); p.mblks = nemar_cdiv(M, 64);
    const long long ntiles = (long long)N * p.tiles_x * p.tiles_y;
    const int wgs = (int)(ntiles < 512 ? ntiles : 512) * p.mblks;
#define K7_FM(C_, F_) hipLaunchKernelGGL((k7_fm_kernel<C_, F_>), dim3(wgs), dim3(256), 0, st, p)
    if (fold) { if (Cs == 1) K7_FM(1, true); else if (Cs == 2) K7_FM(2, true); else if (Cs == 3) K7_FM(3, true); else K7_FM(4, true); }
    else { if (Cs == 1) K7_FM(1, false); else if (Cs == 2) K7_FM(2, false); else if (Cs == 3) K7_FM(3, false); else K7_FM(4, false); }
#undef K7_FM
}
