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
// reads BOTH operands at the same aligned chunk: flat chunk f = y CPR + q of G_s against flat chunk f + r CPR of X.
// Round 4: only G_0 goes through HBM (one copy, 74 instead of 221 MB per batch-16 layer).  Every row of G_0 ends in >= KS - 1 zero
// pixels, so the shifted operand of chunk f is the funnel shift of chunks f - 1 and f of G_0: the kernel keeps the last dword(s) of
// the previous chunk (v_permlane32_swap_b32 moves them between the two 32-lane halves = the two chunks of a stage) and builds
// G_1 .. G_(KS-1) with v_alignbit_b32 in the shadow of the MFMAs (first generation, three HBM copies: nemar_tune(34, 0)).  Both are
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
template <int TW, int KS>          // KS = number of shifted copies written (1: G_0 only)
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

// One pass over gy for BOTH gradients of a 3x3 / pad-1 layer (round 4: gy was read twice, by split_planes_kernel and by
// split_wgrad_g_kernel).  Workgroup = one row of the data gradient's padded planes (rows 0 .. H + 3) of one 64-channel block: the
// image row (or, for the reflect fold rows H + 2 / H + 3, the sum of the two image rows that fold together) goes to LDS once, then
//   * the data gradient's words [n][k / 8][row][slot 0 .. W + 3][8 channels] of that row — the same values, bit for bit, as
//     split_planes_kernel's (scaling by a power of two commutes with the fold sums), and
//   * for image rows, the weight gradient's G_0 words of that row (the three border rows also write the <= 3 zero rows below the image).
template <int TW>
__global__ __launch_bounds__(256) void split_dual_kernel(const float* __restrict__ gy, u32x4* __restrict__ dpl, u32x4* __restrict__ gpl,
                                                         int N, int K, int H, int W, int Hg, int CPR, int reflect, long long dtotal,
                                                         long long gtotal, const unsigned* maxbits, int mstride) {
    __shared__ float tile[64][TW + 1];
    const int Hp = H + 4, Ws = W + 4, KBLK = K >> 6, CG = K >> 3, F = Hg * CPR;
    const int prow = blockIdx.x % Hp, kblk = (blockIdx.x / Hp) % KBLK, n = blockIdx.x / (Hp * KBLK);
    const float scale = pow2_scale(maxbits[n * mstride]);        // per sample
    const bool img = prow >= 1 && prow <= H;
    int ya = 0, yb = -1;
    bool any = img;
    if (img) ya = prow - 1;
    else if (reflect && prow == H + 2) { ya = 0; yb = 2; any = true; }
    else if (reflect && prow == H + 3) { ya = H - 3; yb = H - 1; any = true; }
    const float* src = gy + ((size_t)n * K + kblk * 64) * H * W;
    for (int i = threadIdx.x; i < 64 * W; i += 256) {
        const int kk = i / W, x = i - kk * W;
        const float a = src[((size_t)kk * H + ya) * W + x];
        const float b = src[((size_t)kk * H + max(yb, 0)) * W + x];
        tile[kk][x] = any ? (yb >= 0 ? a + b : a) * scale : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * Ws; i += 256) {            // data-gradient planes: 8 channel groups x (W + 4) slots of row `prow`
        const int cgl = i / Ws, slot = i - cgl * Ws;
        int xa = -1, xb = -1;
        if (slot >= 1 && slot <= W) xa = slot - 1;
        else if (reflect && slot == W + 2) { xa = 0; xb = 2; }
        else if (reflect && slot == W + 3) { xa = W - 3; xb = W - 1; }
        unsigned short h[8], l[8];
        float va[8], vb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                            // (unconditional, clamped LDS reads; masks afterwards)
            va[j] = tile[cgl * 8 + j][max(xa, 0)];
            vb[j] = tile[cgl * 8 + j][max(xb, 0)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) split2_f16(xa >= 0 ? (xb >= 0 ? va[j] + vb[j] : va[j]) : 0.f, h[j], l[j]);
        const size_t t = (((size_t)n * CG + kblk * 8 + cgl) * Hp + prow) * Ws + slot;
        dpl[t] = pack8(h);
        dpl[dtotal + t] = pack8(l);
    }
    const int grow = img ? prow - 1 : (prow == 0 ? H : prow);              // border rows 0, H + 1, H + 2 -> the zero rows H, H + 1, H + 2
    if (grow < Hg && (img || prow <= H + 2)) {
        const int kk = threadIdx.x & 63;
        for (int q = threadIdx.x >> 6; q < CPR; q += 4) {
            unsigned short h[8], l[8];
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[kk][min(q * 8 + e, W - 1)];
#pragma unroll
            for (int e = 0; e < 8; ++e) split2_f16((img && q * 8 + e < W) ? v[e] : 0.f, h[e], l[e]);
            const size_t t = (((size_t)n * KBLK + kblk) * F + (size_t)grow * CPR + q) * 64 + kk;
            gpl[t] = pack8(h);
            gpl[gtotal + t] = pack8(l);
        }
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

// 3x3 layers, or the discriminator's 4x4 / pad 1 layers (16 taps: 16 accumulators per wave).  ONEG: G_0 only comes from HBM.
// XREG: no LDS-DMA — every piece travels global -> registers -> ds_write_b128: loaded at step T for stage T + 4 into one of two register
// sets, written to the ring during step T + 2, in the slot that reloads the register (so a stage is complete at the same barrier as with LDS-DMA).  The same code in
// every wave: no role branches for the compiler to guard with vmcnt(0).  (LDS-DMA written X pieces are what is misread
// beside a foreign LDS-active workgroup; nemar_tune(38) selects the form, common.h has the LDS claim the DMA form needs.)
template <int KS, bool ONEG, bool XREG = false>
__global__ __launch_bounds__(256) void wgrad_split16_kernel(WgParams p) {
    static_assert(!XREG || (KS == 3 && ONEG), "XREG: instantiated and measured for the 3x3 one-copy form only");

    constexpr int GC = ONEG ? 1 : KS;                  // copies of G staged per chunk
    constexpr int RING = 4, NCOL = 4 * GC + 4 * KS, NCP = NCOL / 4;     // copies per wave per stage: NCOL columns over four waves
    constexpr int STAGE16 = NCOL * 64;                 // 4 GC G columns + 4 KS X columns: [(s | r) * 2 + plane][chunk 0 | 1][64 channels]
    constexpr int NMF = 3 * KS * KS, NSL = 2 * GC + 2 * KS + NCP;       // MFMAs and slots per step
    constexpr int NPV = KS / 2;                        // dwords of the previous chunk a shift by <= KS - 1 pixels reaches into
#ifdef NEMAR_HOST_EMULATION
    __shared__ __attribute__((aligned(16))) u32x4 smem[RING * STAGE16];      // (the emulator has no dynamic LDS)
#else
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];             // wg_lds_bytes<KS, ONEG>(), or the whole CU's LDS (common.h)
#endif
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
        if (i < 4 * GC) {
            const int s = i >> 2, pl = (i >> 1) & 1, ch = i & 1;
            csrc[q] = p.G + (size_t)(s * 2 + pl) * p.gplane16 + (((size_t)n * p.KBLK + kblk) * p.F + f0 + ch) * 64 + lane;
        } else {
            const int j = i - 4 * GC, r = j >> 2, pl = (j >> 1) & 1, ch = j & 1;
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
    u32x4 af[2][GC][2], bf[2][KS][2];                  // [register set][s | r][plane]
    u32x4 sh[2][ONEG ? KS : 1][2];                     // ONEG: [register set][s][plane] = G_s built from G_0 (sh[.][0] unused)
    const int a_off = lhi * 64 + wk * 32 + l31, b_off = 4 * GC * 64 + lhi * 64 + wc * 32 + l31;
#define WG_READ(set_, slot_)                                                                                            \
    {                                                                                                                   \
        const u32x4* const S_ = smem + (slot_) * STAGE16;                                                               \
        _Pragma("unroll") for (int i = 0; i < 2 * GC; ++i) af[set_][i >> 1][i & 1] = S_[a_off + i * 128];               \
        _Pragma("unroll") for (int i = 0; i < 2 * KS; ++i) bf[set_][i >> 1][i & 1] = S_[b_off + i * 128];               \
    }
    // G_s of the stage in register set `set_` from its G_0 words and the last NPV dwords of the chunk before: lanes 0..31 hold chunk
    // 0 of the stage (previous chunk = chunk 1 of the stage before: lanes 32..63 of `pv_`), lanes 32..63 chunk 1 (previous = this
    // stage's chunk 0).  in(j) = dword j of the chunk, in(-1), in(-2) = the previous chunk's last dwords; a shift by s pixels takes
    // dword i from in(i - s / 2) (even s) or from the 16-bit funnel of in(i - s / 2), in(i - s / 2 - 1) (odd s).
#define WG_SHIFT(set_, pv_)                                                                                             \
    if (ONEG) {                                                                                                         \
        _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                                              \
            const u32x4 a_ = af[set_][0][pl];                                                                           \
            unsigned in_[4 + NPV];                                 /* in_[NPV + j] = in(j) */                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) in_[NPV + j] = a_[j];                                         \
            _Pragma("unroll") for (int j = 0; j < NPV; ++j) {                                                           \
                const auto sw_ = __builtin_amdgcn_permlane32_swap(pv_[pl][j], a_[4 - NPV + j], false, false);           \
                in_[j] = lhi ? sw_[0] : sw_[1];                                                                         \
            }                                                                                                           \
            _Pragma("unroll") for (int s = 1; s < KS; ++s)                                                              \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
                    const int j_ = NPV + i - (s >> 1);                                                                  \
                    sh[set_][s][pl][i] = (s & 1) ? __builtin_amdgcn_alignbit(in_[j_], in_[j_ - 1], 16) : in_[j_];       \
                }                                                                                                       \
        }                                                                                                               \
    }
    // partial products, smallest first: (l h') (h l') (h h')
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#define WG_A(set_, s_, pl_) ((ONEG && (s_) > 0) ? sh[set_][(ONEG ? (s_) : 0)][pl_] : af[set_][ONEG ? 0 : (s_)][pl_])

    u32x4 xr[2][XREG ? NCP : 1];                       // XREG: stage T + 4 (set T & 1) on its way from global memory to the ring
    if (XREG) {
        // stages 0, 1 straight into the ring; stages 2, 3 wait in the two register sets for steps 0 and 1
#pragma unroll
        for (int stg = 0; stg < 2; ++stg) {
            const int st_ = min(stg, nsteps - 1);
            u32x4* const d_ = smem + stg * STAGE16 + wid * (NCP * 64);
#pragma unroll
            for (int q = 0; q < NCP; ++q) d_[q * 64 + lane] = csrc[q][(size_t)st_ * 128];
        }
#pragma unroll
        for (int stg = 2; stg < 4; ++stg)
#pragma unroll
            for (int q = 0; q < (XREG ? NCP : 1); ++q) xr[stg & 1][q] = csrc[q][(size_t)min(stg, nsteps - 1) * 128];
    } else {
        WG_COPIES(0)
        WG_COPIES(1)
        WG_COPIES(2)
        WG_COPIES(3)
        WG_VMCNT(2 * NCP)                              // stages 0 and 1 have landed (2 and 3 may be in flight)
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                // (XREG: the ds_writes of stages 0, 1)
    __builtin_amdgcn_s_barrier();
    WG_READ(0, 0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    {
        unsigned zero_[2][NPV > 0 ? NPV : 1];          // the chunk before a slab's first one is the zero tail of the row above
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < NPV; ++j) zero_[pl][j] = 0u;
        WG_SHIFT(0, zero_)
    }
    __builtin_amdgcn_s_barrier();                      // every wave holds the fragments of step 0: slot 0 may be refilled
    for (int T0 = 0; T0 < nsteps; T0 += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int T = T0 + half, cur = half, nxt = half ^ 1;
            // NSL slots of [<= 1 memory instruction][its share of the NMF MFMAs], pinned (see conv_split16.hip)
#define WG_MFMAS(beg_, end_)                              /* m = KS KS q + KS s + r */                                      \
            _Pragma("unroll") for (int m_ = (beg_); m_ < (end_) && m_ < NMF; ++m_) {                                    \
                const int q_ = m_ / (KS * KS), s_ = (m_ % (KS * KS)) / KS, r_ = m_ % KS;                                \
                acc[s_][r_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, WG_A(cur, s_, PA[q_])),  \
                                                                     __builtin_bit_cast(f16x8, bf[cur][r_][PB[q_]]),    \
                                                                     acc[s_][r_], 0, 0, 0);                             \
            }                                                                                                           \
            __builtin_amdgcn_sched_barrier(0);
#define WG_SLOT(i_) WG_MFMAS((i_) * NMF / NSL, ((i_) + 1) * NMF / NSL)
            const u32x4* const S_ = smem + ((T + 1) & (RING - 1)) * STAGE16;
#pragma unroll
            for (int i = 0; i < 2 * GC; ++i) {
                af[nxt][i >> 1][i & 1] = S_[a_off + i * 128];
                WG_SLOT(i)
            }
#pragma unroll
            for (int i = 0; i < 2 * KS; ++i) {
                bf[nxt][i >> 1][i & 1] = S_[b_off + i * 128];
                WG_SLOT(2 * GC + i)
            }
            if (ONEG) {
                // the 2 G_0 words of step T + 1 have landed once at most the 2 KS X reads are outstanding (LDS returns in order):
                // build its shifted operands now, between this step's MFMAs
                __builtin_amdgcn_s_waitcnt(0xC07F | ((2 * KS) << 8));
                unsigned pv_[2][NPV > 0 ? NPV : 1];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int j = 0; j < NPV; ++j) pv_[pl][j] = af[cur][0][pl][4 - NPV + j];
                WG_SHIFT(nxt, pv_)
            }
            {
                const int st_ = min(T + 4, nsteps - 1);
                u32x4* const d_ = smem + (T & (RING - 1)) * STAGE16 + wid * (NCP * 64);
                // XREG: in the slot that reloads a register of set `cur`, its old content — stage T + 2, loaded at step T - 2 — goes into its
                // ring slot (free since step T - 2) first.  Late in the step: the wait for that load sits behind most of the step's MFMAs.
                u32x4* const d2_ = smem + ((T + 2) & (RING - 1)) * STAGE16 + wid * (NCP * 64);
#pragma unroll
                for (int q = 0; q < NCP; ++q) {
                    if (XREG) {
                        d2_[q * 64 + lane] = xr[cur][XREG ? q : 0];
                        xr[cur][XREG ? q : 0] = csrc[q][(size_t)st_ * 128];
                    } else {
                        glds16(csrc[q] + (size_t)st_ * 128, d_ + q * 64);
                    }
                    WG_SLOT(2 * GC + 2 * KS + q)
                }
            }
#undef WG_SLOT
#undef WG_MFMAS
            if (!XREG) WG_VMCNT(2 * NCP)               // this wave's copies of stage T + 2 have landed (T + 3, T + 4 in flight)
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): every fragment of step T + 1 is in registers (XREG: and stage T + 2 is written)
            __builtin_amdgcn_s_barrier();              // stage T + 2 complete for all waves; slot of stage T + 1 is free
        }
    }
    wait_vmem();                                       // (the tail's re-copies)
#undef WG_READ
#undef WG_SHIFT
#undef WG_A
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

template <int KS, bool ONEG>
constexpr size_t wg_lds_bytes() { return (size_t)4 * (4 * (ONEG ? 1 : KS) + 4 * KS) * 64 * 16; }      // RING x STAGE16 words of 16 bytes

template <int KS, bool ONEG, bool XREG = false>
void wg_launch(int grid, const WgParams& p, hipStream_t st) {
    const void* const k = reinterpret_cast<const void*>(&wgrad_split16_kernel<KS, ONEG, XREG>);
    // (only a form that stages by LDS-DMA needs the whole-CU claim: common.h)
    hipLaunchKernelGGL((wgrad_split16_kernel<KS, ONEG, XREG>), dim3(grid), dim3(256), nemar_lds_bytes(k, wg_lds_bytes<KS, ONEG>(), !XREG && (g_lds_claim & 1) != 0), st, p);
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
NEMAR_SWITCH(int, g_wg_xreg, 1);      // nemar_tune(38): the 3x3 kernel stages through registers (1, default: 373 vs 364 us per batch-16 call, and no
                                      // whole-CU LDS claim needed: 0.25 ms per step less than LDS-DMA + claim) / by LDS-DMA (0)
static NEMAR_SWITCH(int, g_one_g, 1);        // nemar_tune(34): 1 = one copy of the gy planes, shifted operands built in registers; 0 = KS copies in HBM
#ifdef NEMAR_AB
void nemar_split16_wgrad_tune(int v) { g_one_g = v ? 1 : 0; }
#endif

size_t nemar_split16_wgrad_g_bytes(int N, int H, int W, int K, int KS) {
    return g_one_g ? (size_t)2 * N * K * g_rows(H, KS) * ((W + 2 + 7) / 8) * 16 : 0;
}

void nemar_split16_dual_split(const float* gy, void* dplanes, void* gplanes, int N, int K, int H, int W, int mode, const unsigned* maxbits,
                              int mstride, hipStream_t st) {
    const int CPR = (W + 2 + 7) / 8, Hg = g_rows(H, 3);
    const long long dtotal = (long long)N * (K / 8) * (H + 4) * (W + 4), gtotal = (long long)N * K * Hg * CPR;
    const int reflect = mode == SPLIT16_DGRAD_REFLECT ? 1 : 0;
    const dim3 grid(N * (K / 64) * (H + 4));
#define WG_DUAL(TW_)                                                                                                               \
    hipLaunchKernelGGL((split_dual_kernel<TW_>), grid, dim3(256), 0, st, gy, (u32x4*)dplanes, (u32x4*)gplanes, N, K, H, W, Hg, CPR, reflect, \
                       dtotal, gtotal, maxbits, mstride);
    if (W <= 64) { WG_DUAL(64) } else if (W <= 128) { WG_DUAL(128) } else { WG_DUAL(256) }
#undef WG_DUAL
}

static thread_local const float* t_bias_partials = nullptr;
static thread_local float* t_bias_dst = nullptr;
void nemar_split16_wgrad_set_bias(const float* bias_partials, float* gb) { t_bias_partials = bias_partials; t_bias_dst = gb; }

void nemar_split16_wgrad(const float* x, const float* gy, float* gw, int N, int C, int H, int W, int K, int KS, int reflect,
                         void* scratch, float* part, int xcd_map, const void* g_planes, const void* x_planes, hipStream_t st) {
    const int CPR = (W + 2 + 7) / 8, KBLK = K / 64, CBLK = C / 64;
    const int OHg = H + 3 - KS, OWg = W + 3 - KS, Hg = g_rows(H, KS), Hx = Hg + KS - 1;
    const long long gtotal = (long long)N * K * Hg * CPR, xtotal = (long long)N * C * Hx * CPR;      // words per plane block
    const bool oneg = g_one_g != 0;
    const bool have_g = oneg && g_planes != nullptr;
    const u32x4* const G = have_g ? (const u32x4*)g_planes : (const u32x4*)scratch;
    const bool have_x = x_planes != nullptr && KS == 3;       // a forward producer wrote the X planes (norm_planes.hip)
    u32x4* const X = have_x ? (u32x4*)const_cast<void*>(x_planes) : (u32x4*)scratch + 2 * (oneg ? 1 : KS) * gtotal;
    unsigned* const mw = (unsigned*)((char*)scratch + nemar_split16_wgrad_scratch_bytes(N, C, H, W, K, KS) - 2048);    // 2 x 256 words
    int gstride = 0, xstride = 0;
    const unsigned* const gmax = nemar_split16_source_max(gy, N, (long long)K * OHg * OWg, mw, &gstride, st);
    const unsigned* const xmax = nemar_split16_source_max(x, N, (long long)C * H * W, mw + 256, &xstride, st);
#define WG_SPLIT(TW_)                                                                                                              \
    if (have_g) {                                                                                                                  \
    } else if (oneg)                                                                                                               \
        hipLaunchKernelGGL((split_wgrad_g_kernel<TW_, 1>), dim3(N * KBLK * Hg), dim3(256), 0, st, gy, (u32x4*)scratch, N, K, OHg, OWg, Hg, CPR, gtotal, \
                           gmax, gstride);                                                                                         \
    NEMAR_AB_ONLY(                                                                                                                 \
    else if (KS == 3)                                                                                                              \
        hipLaunchKernelGGL((split_wgrad_g_kernel<TW_, 3>), dim3(N * KBLK * Hg), dim3(256), 0, st, gy, (u32x4*)scratch, N, K, OHg, OWg, Hg, CPR, gtotal, \
                           gmax, gstride);                                                                                         \
    else                                                                                                                           \
        hipLaunchKernelGGL((split_wgrad_g_kernel<TW_, 4>), dim3(N * KBLK * Hg), dim3(256), 0, st, gy, (u32x4*)scratch, N, K, OHg, OWg, Hg, CPR, gtotal, \
                           gmax, gstride);)                                                                                        \
    if (!have_x)                                                                                                                   \
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
    NEMAR_AB_ONLY(if (KS == 3 && oneg && !g_wg_xreg) wg_launch<3, true, false>(grid, p, st); else)
    if (KS == 3 && oneg) wg_launch<3, true, true>(grid, p, st);
    else if (oneg) wg_launch<4, true>(grid, p, st);
#ifdef NEMAR_AB      // nemar_tune(34, 0): KS shifted copies of the gy planes in HBM
    else if (KS == 3) wg_launch<3, false>(grid, p, st);
    else wg_launch<4, false>(grid, p, st);
#endif
    // (+ the bias gradient from the backward producer's per-plane sums, where the caller handed them over: one launch for both reductions)
    nemar_sum_partials_pair(part, (long long)K * C * KS * KS, splits, gw, (long long)K * C * KS * KS, t_bias_partials, K, N, t_bias_dst, K, true, st);
    t_bias_partials = nullptr; t_bias_dst = nullptr;
}
