// nemar_amd — weight gradient of the wide 3x3 / stride-1 / pad-1 layers on the 16-bit matrix pipe (fp16 x 3 products at fp32
// accuracy: the operand split, its scaling and its error analysis are those of conv_split16.hip).
//
//     dW[k][c][r][s] = sum over n, y, x of  gy[n][k][y][x] * xpad[n][c][y + r][x + s]          (reference: autograd of
//     nn.Conv2d inside ResnetBlock, models/networks.py:418-439 — 36 of these per step, 13 of the 59 ms after the forward and data
//     gradient moved to conv_split16.hip)
//
// The reduction runs over PIXELS, so an MFMA operand (8 consecutive reduction elements per lane) is 8 consecutive pixels of one
// channel, and the tap's horizontal shift s would misalign every second operand read.  The split pass therefore writes
//   G_s[n][k][y][x'] = gy[n][k][y][x' - s]   for s = 0, 1, 2 (zero where x' - s falls outside the row),   x' = 0 .. 8 CPR - 1
//   X  [n][c][yp][x'] = xpad[n][c][yp][x']                    (padding materialised, zero beyond column W + 1)
// with CPR = ceil((W + 2) / 8) 8-pixel chunks per row, so that   dW[k][c][r][s] = sum_{n,y,x'} G_s[k][y][x'] X[c][y + r][x']
// reads BOTH operands at the same aligned chunk: flat chunk f = y CPR + q of G_s against flat chunk f + r CPR of X.  Both are
// stored tile-ordered — [64-channel block][flat chunk][64 channels][8 pixels] fp16, high and low plane — so a stage of the
// reduction is a run of contiguous 1 KiB copies and an LDS fragment read is conflict-free (32 lanes = 32 consecutive channels).
//
// wgrad_split16_kernel: workgroup = 64 k x 64 c x ALL NINE taps over RB image rows of one image (a slab of the pixel reduction;
// slabs are summed in order by nemar_sum_partials: bitwise reproducible), four waves of 32 k x 32 c (nine accumulators each), one
// per SIMD.  Same machinery as igemm_split16_kernel: no loader waves (each wave issues a quarter of every stage's copies in the
// shadow of its MFMAs and waits for ITS copies before the step's barrier), 4-slot stage ring with the stage for step T + 4 issued
// during step T, every fragment of step T + 1 read during step T into a second register set, the body cut into pinned slots.
#include "common.h"
#include "conv_split16.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pow2_scale(unsigned maxbits) {       // as in conv_split16.hip: 2^(11 - floor(log2 max))
    const int e = (int)((maxbits >> 23) & 255u);
    if (e == 0 || e == 255) return 1.f;
    const int se = 127 + 11 - (e - 127);
    if (se < 1 || se > 254) return 1.f;
    return __builtin_bit_cast(float, (unsigned)se << 23);
}
__device__ __forceinline__ unsigned short f16_rn(float v) {
    const _Float16 h = (_Float16)v;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ void split2_f16(float v, unsigned short& h, unsigned short& l) {
    h = f16_rn(v);
    const float r = v - (float)__builtin_bit_cast(_Float16, h);
    l = f16_rn(r);
}
__device__ __forceinline__ u32x4 pack8(const unsigned short* b) {
    u32x4 o;
    o[0] = (unsigned)b[0] | ((unsigned)b[1] << 16);
    o[1] = (unsigned)b[2] | ((unsigned)b[3] << 16);
    o[2] = (unsigned)b[4] | ((unsigned)b[5] << 16);
    o[3] = (unsigned)b[6] | ((unsigned)b[7] << 16);
    return o;
}
__device__ __forceinline__ int mirror(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__device__ __forceinline__ void glds16(const u32x4* g, u32x4* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// G_s planes: word (((s * 2 + pl) * N + n) * KBLK + kblk) * F + f) * 64 + kk, F = H * CPR.  One workgroup = one image row y of one
// 64-channel block: the 64 rows are read coalesced (256 B runs) into LDS, then thread (kk, chunk) takes the 10 texels x' - 2 .. x' + 7
// of its chunk from there and writes the three shifted versions, two planes each — 64 consecutive channels = one 1 KiB store.
// (A first version read straight from global memory, one channel row per lane: 7x over-fetch, 60 us; PMC in profiles/.)
// TW = row-tile width class (>= W): 64 x (TW + 1) floats of LDS, so that narrow maps keep many workgroups per CU (the pass is
// latency-bound per workgroup: load a row block, barrier, write)
constexpr int SPLIT_MAXW = 256;
// gy is [N, K, H, W] (H x W = ITS extents: one less than the layer's input for the 4x4 layers); the planes have Hg >= H rows (zero
// below row H: the row count is rounded up to a whole number of row blocks) and KS shifted versions.
template <int TW, int KS>
__global__ __launch_bounds__(256) void split_wgrad_g_kernel(const float* __restrict__ gy, u32x4* __restrict__ out, int N, int K, int H,
                                                            int W, int Hg, int CPR, long long total, const unsigned* maxbits,
                                                            int mstride) {
    __shared__ float tile[64][TW + 1];
    const int KBLK = K >> 6, F = Hg * CPR;
    const int y = blockIdx.x % Hg, kblk = (blockIdx.x / Hg) % KBLK, n = blockIdx.x / (Hg * KBLK);
    const float scale = pow2_scale(maxbits[n * mstride]);        // per sample
    const bool rowok = y < H;
    const float* src = gy + (((size_t)n * K + kblk * 64) * H + (rowok ? y : 0)) * W;
    for (int i = threadIdx.x; i < 64 * W; i += 256) {
        const int kk = i / W, x = i - kk * W;
        tile[kk][x] = rowok ? src[(size_t)kk * H * W + x] * scale : 0.f;
    }
    __syncthreads();
    const int kk = threadIdx.x & 63;
    for (int q = threadIdx.x >> 6; q < CPR; q += 4) {
        float v[8 + KS - 1];
#pragma unroll
        for (int e = 0; e < 8 + KS - 1; ++e) {
            const int x = q * 8 - (KS - 1) + e;
            v[e] = (x >= 0 && x < W) ? tile[kk][x] : 0.f;
        }
        const size_t t = (((size_t)n * KBLK + kblk) * F + (size_t)y * CPR + q) * 64 + kk;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            unsigned short h[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) split2_f16(v[e + KS - 1 - s], h[e], l[e]);
            out[(size_t)(s * 2) * total + t] = pack8(h);
            out[(size_t)(s * 2 + 1) * total + t] = pack8(l);
        }
    }
}

// X planes: word ((pl * N + n) * CBLK + cblk) * FX + f) * 64 + cc, FX = (H + 2) * CPR; padding materialised.  Same structure: one
// workgroup = one padded row yp of one 64-channel block.
template <int TW>
__global__ __launch_bounds__(256) void split_wgrad_x_kernel(const float* __restrict__ x, u32x4* __restrict__ out, int N, int C, int H,
                                                            int W, int Hp, int CPR, int reflect, long long total,
                                                            const unsigned* maxbits, int mstride) {
    __shared__ float tile[64][TW + 1];
    const int CBLK = C >> 6, FX = Hp * CPR;            // Hp >= H + 2 plane rows (rows beyond the padded image: zero)
    const int yp = blockIdx.x % Hp, cblk = (blockIdx.x / Hp) % CBLK, n = blockIdx.x / (Hp * CBLK);
    const float scale = pow2_scale(maxbits[n * mstride]);        // per sample
    int y = yp - 1;
    bool rowok = yp < H + 2;
    if (reflect) y = mirror(min(y, H), H);
    else rowok = rowok && (unsigned)y < (unsigned)H;
    const float* src = x + (((size_t)n * C + cblk * 64) * H + (rowok ? y : 0)) * W;
    for (int i = threadIdx.x; i < 64 * W; i += 256) {
        const int cc = i / W, xs = i - cc * W;
        tile[cc][xs] = rowok ? src[(size_t)cc * H * W + xs] * scale : 0.f;
    }
    __syncthreads();
    const int cc = threadIdx.x & 63;
    for (int q = threadIdx.x >> 6; q < CPR; q += 4) {
        unsigned short h[8], l[8];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {                // all eight LDS reads first (unconditional, clamped), then the masks
            const int xs = q * 8 + e - 1;
            v[e] = tile[cc][min(max(reflect ? mirror(xs, W) : xs, 0), W - 1)];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int xp = q * 8 + e;                // padded column
            const bool ok = xp < W + 2 && (reflect || (unsigned)(xp - 1) < (unsigned)W);
            split2_f16(ok ? v[e] : 0.f, h[e], l[e]);
        }
        const size_t t = (((size_t)n * CBLK + cblk) * FX + (size_t)yp * CPR + q) * 64 + cc;
        out[t] = pack8(h);
        out[total + t] = pack8(l);
    }
}

struct WgParams {
    const u32x4* G;            // [3 s][2 planes] blocks of gplane16 words
    const u32x4* X;            // [2 planes] blocks of xplane16 words
    float* part;               // slabs [splits][K][C][9]
    int N, H, W, C, K;         // H, W: the layer's INPUT extents
    int CPR, F, FX, RB, spi;   // chunks per row, flat chunks per image of G / X, rows per split, splits per image
    int KBLK, CBLK;
    long long gplane16, xplane16;
    const unsigned* gmax;      // per-sample max words (word n * stride)
    const unsigned* xmax;
    int gstride, xstride;
    int xcd;
};

template <int KS>        // 3x3 layers, or the discriminator's 4x4 / pad 1 layers (16 taps: 16 accumulators per wave)
__global__ __launch_bounds__(256) void wgrad_split16_kernel(WgParams p) {
    constexpr int RING = 4, NCP = 2 * KS;              // copies per wave per stage: 4 KS of G + 4 KS of X over four waves
    constexpr int STAGE16 = 8 * KS * 64;               // 4 KS G columns + 4 KS X columns: [(s | r) * 2 + plane][chunk 0 | 1][64 channels]
    constexpr int NMF = 3 * KS * KS, NSL = 2 * 2 * KS + NCP;       // MFMAs and slots per step
    __shared__ __attribute__((aligned(16))) u32x4 smem[RING * STAGE16];
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int tiles = p.KBLK * p.CBLK;
    int t = blockIdx.x;
    if (p.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);      // the tiles of one pixel slab stay on one XCD (same sources)
    const int split = t / tiles, tile = t - split * tiles;
    const int kblk = tile / p.CBLK, cblk = tile - kblk * p.CBLK;
    const int n = split / p.spi, y0 = (split - n * p.spi) * p.RB;
    const int nsteps = p.RB * p.CPR / 2, f0 = y0 * p.CPR;

    // ---- this wave's copies of a stage: copy i = NCP wid + q of 8 KS (i < 4 KS: G, [s][plane][chunk]; else X, [r][plane][chunk]) ----
    const u32x4* csrc[NCP];
#pragma unroll
    for (int q = 0; q < NCP; ++q) {
        const int i = NCP * wid + q;
        if (i < 4 * KS) {
            const int s = i >> 2, pl = (i >> 1) & 1, ch = i & 1;
            csrc[q] = p.G + (size_t)(s * 2 + pl) * p.gplane16 + (((size_t)n * p.KBLK + kblk) * p.F + f0 + ch) * 64 + lane;
        } else {
            const int j = i - 4 * KS, r = j >> 2, pl = (j >> 1) & 1, ch = j & 1;
            csrc[q] = p.X + (size_t)pl * p.xplane16 + (((size_t)n * p.CBLK + cblk) * p.FX + f0 + r * p.CPR + ch) * 64 + lane;
        }
    }
#define WG_COPIES(stage_)                                                                                               \
    {                                                                                                                   \
        const int st_ = min((stage_), nsteps - 1);                   /* tail: harmless re-copies of the last stage */   \
        u32x4* const d_ = smem + ((stage_) & (RING - 1)) * STAGE16 + wid * (NCP * 64);                                  \
        _Pragma("unroll") for (int q = 0; q < NCP; ++q) glds16(csrc[q] + (size_t)st_ * 128, d_ + q * 64);               \
    }
#define WG_VMCNT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14));

    const int l31 = lane & 31, lhi = lane >> 5;
    const int wk = wid >> 1, wc = wid & 1;
    f32x16 acc[KS][KS];                                // [s][r]
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int r = 0; r < KS; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][r][e] = 0.f;
    u32x4 af[2][KS][2], bf[2][KS][2];                  // [register set][s | r][plane]
    const int a_off = lhi * 64 + wk * 32 + l31, b_off = 4 * KS * 64 + lhi * 64 + wc * 32 + l31;
#define WG_READ(set_, slot_)                                                                                            \
    {                                                                                                                   \
        const u32x4* const S_ = smem + (slot_) * STAGE16;                                                               \
        _Pragma("unroll") for (int i = 0; i < 2 * KS; ++i) af[set_][i >> 1][i & 1] = S_[a_off + i * 128];               \
        _Pragma("unroll") for (int i = 0; i < 2 * KS; ++i) bf[set_][i >> 1][i & 1] = S_[b_off + i * 128];               \
    }
    // partial products, smallest first: (l h') (h l') (h h')
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};

    WG_COPIES(0)
    WG_COPIES(1)
    WG_COPIES(2)
    WG_COPIES(3)
    WG_VMCNT(2 * NCP)                                  // stages 0 and 1 have landed (2 and 3 may be in flight)
    __builtin_amdgcn_s_barrier();
    WG_READ(0, 0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                      // every wave holds the fragments of step 0: slot 0 may be refilled
    for (int T0 = 0; T0 < nsteps; T0 += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int T = T0 + half, cur = half, nxt = half ^ 1;
            // NSL slots of [<= 1 memory instruction][its share of the NMF MFMAs], pinned (see conv_split16.hip)
#define WG_MFMAS(beg_, end_)                              /* m = KS KS q + KS s + r */                                      \
            _Pragma("unroll") for (int m_ = (beg_); m_ < (end_) && m_ < NMF; ++m_) {                                    \
                const int q_ = m_ / (KS * KS), s_ = (m_ % (KS * KS)) / KS, r_ = m_ % KS;                                \
                acc[s_][r_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[cur][s_][PA[q_]]),    \
                                                                     __builtin_bit_cast(f16x8, bf[cur][r_][PB[q_]]),    \
                                                                     acc[s_][r_], 0, 0, 0);                             \
            }                                                                                                           \
            __builtin_amdgcn_sched_barrier(0);
#define WG_SLOT(i_) WG_MFMAS((i_) * NMF / NSL, ((i_) + 1) * NMF / NSL)
            const u32x4* const S_ = smem + ((T + 1) & (RING - 1)) * STAGE16;
#pragma unroll
            for (int i = 0; i < 2 * KS; ++i) {
                af[nxt][i >> 1][i & 1] = S_[a_off + i * 128];
                WG_SLOT(i)
            }
#pragma unroll
            for (int i = 0; i < 2 * KS; ++i) {
                bf[nxt][i >> 1][i & 1] = S_[b_off + i * 128];
                WG_SLOT(2 * KS + i)
            }
            {
                const int st_ = min(T + 4, nsteps - 1);
                u32x4* const d_ = smem + (T & (RING - 1)) * STAGE16 + wid * (NCP * 64);
#pragma unroll
                for (int q = 0; q < NCP; ++q) {
                    glds16(csrc[q] + (size_t)st_ * 128, d_ + q * 64);
                    WG_SLOT(4 * KS + q)
                }
            }
#undef WG_SLOT
#undef WG_MFMAS
            WG_VMCNT(2 * NCP)                          // this wave's copies of stage T + 2 have landed (T + 3, T + 4 in flight)
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): every fragment of step T + 1 is in registers
            __builtin_amdgcn_s_barrier();              // stage T + 2 complete for all waves; slot of stage T + 1 is free
        }
    }
    wait_vmem();                                       // (the tail's re-copies)
#undef WG_READ
#undef WG_VMCNT
#undef WG_COPIES

    // slab [split][k][c][tap = KS r + s]; D register e of lane l = row (e & 3) + 8 (e >> 2) + 4 (l >> 5), column l & 31
    const float unscale = 1.f / (pow2_scale(p.gmax[n * p.gstride]) * pow2_scale(p.xmax[n * p.xstride]));     // a slab = rows of ONE image
    float* const slab = p.part + (size_t)split * p.K * p.C * (KS * KS);
    const int c = cblk * 64 + wc * 32 + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = kblk * 64 + wk * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        float* const o = slab + ((size_t)k * p.C + c) * (KS * KS);
#pragma unroll
        for (int r = 0; r < KS; ++r)
#pragma unroll
            for (int s = 0; s < KS; ++s) o[KS * r + s] = acc[s][r][e] * unscale;
    }
}

int rows_per_split(int N, int H, int CPR, int tiles) {
    // ~256 workgroups: splits per image = ceil(256 / (tiles N)), rounded to a divisor of H whose row block is a whole number of
    // double steps (RB CPR % 4 == 0)
    int want = (256 + tiles * N - 1) / (tiles * N);
    if (want < 1) want = 1;
    int best = 0;
    for (int spi = 1; spi <= H; ++spi) {
        if (H % spi) continue;
        const int rb = H / spi;
        if ((rb * CPR) % 4) continue;
        if (best == 0 || spi <= want) best = spi;
        if (spi >= want) break;
    }
    return best ? H / best : 0;
}

}  // namespace

// Hg = rows of the gy planes: gy's H + 2 - KS + 1... i.e. (H + 3 - KS) rounded up to a multiple of 4 (whole double steps for any CPR)
static int g_rows(int H, int KS) { return (H + 3 - KS + 3) / 4 * 4; }

bool nemar_split16_wgrad_eligible(int N, int C, int H, int W, int K, int R, int S, int stride, int pad) {
    if (R != S || (R != 3 && R != 4) || stride != 1 || pad != 1) return false;
    if (C % 64 || K % 64 || C < 128 || K < 128 || W % 8 || H < 4 || W < 8 || W > SPLIT_MAXW || N > 256) return false;
    const int CPR = (W + 2 + 7) / 8;
    if (rows_per_split(N, g_rows(H, R), CPR, (K / 64) * (C / 64)) == 0) return false;
    if ((long long)N * (C > K ? C : K) * (H + 6) * CPR * 64 >= (1ll << 31)) return false;
    return true;
}

size_t nemar_split16_wgrad_scratch_bytes(int N, int C, int H, int W, int K, int KS) {
    const int CPR = (W + 2 + 7) / 8, Hg = g_rows(H, KS);
    return ((size_t)2 * KS * N * K * Hg * CPR + (size_t)2 * N * C * (Hg + KS - 1) * CPR) * 16 + 16384;
}

int nemar_split16_wgrad_splits(int N, int C, int H, int W, int K, int KS) {
    const int CPR = (W + 2 + 7) / 8, Hg = g_rows(H, KS);
    const int rb = rows_per_split(N, Hg, CPR, (K / 64) * (C / 64));
    return rb ? N * (Hg / rb) : 0;
}

// gw [K][C][KS][KS] += dW;  `part` holds nemar_split16_wgrad_splits slabs of K C KS KS floats.  x [N, C, H, W], gy [N, K, H + 3 - KS, W + 3 - KS]
void nemar_split16_wgrad(const float* x, const float* gy, float* gw, int N, int C, int H, int W, int K, int KS, int reflect,
                         void* scratch, float* part, int xcd_map, hipStream_t st) {
    const int CPR = (W + 2 + 7) / 8, KBLK = K / 64, CBLK = C / 64;
    const int OHg = H + 3 - KS, OWg = W + 3 - KS, Hg = g_rows(H, KS), Hx = Hg + KS - 1;
    const long long gtotal = (long long)N * K * Hg * CPR, xtotal = (long long)N * C * Hx * CPR;      // words per plane block
    u32x4* const G = (u32x4*)scratch;
    u32x4* const X = G + 2 * KS * gtotal;
    unsigned* const mw = (unsigned*)((char*)scratch + nemar_split16_wgrad_scratch_bytes(N, C, H, W, K, KS) - 2048);    // 2 x 256 words
    int gstride = 0, xstride = 0;
    const unsigned* const gmax = nemar_split16_source_max(gy, N, (long long)K * OHg * OWg, mw, &gstride, st);
    const unsigned* const xmax = nemar_split16_source_max(x, N, (long long)C * H * W, mw + 256, &xstride, st);
#define WG_SPLIT(TW_)                                                                                                              \
    if (KS == 3)                                                                                                                   \
        hipLaunchKernelGGL((split_wgrad_g_kernel<TW_, 3>), dim3(N * KBLK * Hg), dim3(256), 0, st, gy, G, N, K, OHg, OWg, Hg, CPR, gtotal, \
                           gmax, gstride);                                                                                         \
    else                                                                                                                           \
        hipLaunchKernelGGL((split_wgrad_g_kernel<TW_, 4>), dim3(N * KBLK * Hg), dim3(256), 0, st, gy, G, N, K, OHg, OWg, Hg, CPR, gtotal, \
                           gmax, gstride);                                                                                         \
    hipLaunchKernelGGL((split_wgrad_x_kernel<TW_>), dim3(N * CBLK * Hx), dim3(256), 0, st, x, X, N, C, H, W, Hx, CPR, reflect, xtotal, \
                       xmax, xstride);
    if (W <= 64) { WG_SPLIT(64) } else if (W <= 128) { WG_SPLIT(128) } else { WG_SPLIT(256) }
#undef WG_SPLIT
    WgParams p;
    p.G = G; p.X = X; p.part = part;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K;
    p.CPR = CPR; p.F = Hg * CPR; p.FX = Hx * CPR;
    p.RB = rows_per_split(N, Hg, CPR, KBLK * CBLK);
    p.spi = Hg / p.RB;
    p.KBLK = KBLK; p.CBLK = CBLK;
    p.gplane16 = gtotal; p.xplane16 = xtotal;
    p.gmax = gmax; p.xmax = xmax; p.gstride = gstride; p.xstride = xstride;
    const int splits = N * p.spi, grid = splits * KBLK * CBLK;
    p.xcd = (xcd_map && grid % 8 == 0 && (grid / 8) % (KBLK * CBLK) == 0) ? 1 : 0;
    if (KS == 3) hipLaunchKernelGGL((wgrad_split16_kernel<3>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_split16_kernel<4>), dim3(grid), dim3(256), 0, st, p);
    nemar_sum_partials(part, (long long)K * C * KS * KS, splits, gw, (long long)K * C * KS * KS, true, st);
}
