// nemar_amd — GENERAL convolutions on the 16-bit matrix pipe at fp32 accuracy: every forward / data-gradient / transposed
// convolution of the three networks that is not one of the translation net's wide residual-block layers (those have their own
// kernel, conv_split16.hip) — the stride-2 3x3 layers and ConvTranspose2d of ResnetGenerator (reference models/networks.py:
// 355-374), the discriminator's 4x4 stride-2 layers (networks.py:576-593), the registration net's 32 / 64-channel 3x3 layers incl.
// the decoder's concatenated inputs (models/stn/unet_stn.py:28-102, layers.py:73-106).  On the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) these ran at 40-110 TF: 13 of the 45 ms of a round-2 step.
//
// Arithmetic: fp16 x 3 as in conv_split16.hip — every fp32 operand v becomes  v s = h + l + e  with h = RN16(v s),
// l = RN16(v s - h), |e| <= 2^-22 |v s| while l stays a normal fp16 number, and the fp32 product is rebuilt from the three MFMA
// products h h' + h l' + l h' accumulated in fp32.  What is different here:
//   * THE SPLIT HAPPENS IN THE KERNEL.  The source is read as plain NCHW fp32 (4 bytes per element — exactly what two fp16
//     planes would cost), converted in registers and written to LDS as channel-blocked hi / lo planes ([pixel][8 channels] =
//     one 16-byte MFMA operand per word).  No split pass, no max pass, no scratch arena, no process-global state: the operator
//     reads its inputs once and writes its output once.
//   * ONLINE BLOCK SCALING.  fp16's exponent range is narrow, so the source is scaled by a power of two — chosen PER WORKGROUP
//     TILE AND PER 16-CHANNEL CHUNK from the largest finite magnitude the tile's own halo holds (a register max over what was
//     just loaded + one LDS exchange), monotone over the chunks: when a later chunk needs a smaller scale the fp32
//     accumulators are multiplied by the exact power-of-two ratio first (the flash-attention running-max idea).  An output's
//     error is therefore relative to ITS OWN tile's receptive field — samples, image regions or channels of very different
//     magnitude inside one tensor do not share a scale.  Bound (include/nemar_hip.h): elements within 2^-11 of the running
//     tile maximum keep the full 22 bits of h + l; below that the absolute error is 2^-36 of that maximum.  Non-finite
//     inputs propagate as NaN / Inf to exactly the outputs whose receptive field holds them (they are excluded from the max).
//     Weights: one power-of-two scale per tensor, from a max pass at pack time (once per optimizer step).
//   * Halo tiles.  A workgroup owns MB = 32 MT output channels x NP = 128 NT output pixels (RT rows x TW columns of one image).
//     Per 16-channel chunk the source halo of the tile ((RT - 1) stride + tap extent rows) is loaded ONCE, split once, and every
//     tap reads it at a shifted LDS address: conversion work is per source element, not per (element, tap).  Stride-2 sources
//     are stored with even and odd columns de-interleaved so that the 32 pixels of an MFMA operand tile stay consecutive words.
//   * The chunk's packed weights ([tap][plane][k group][MB] 16-byte words, written by s16g_pack_kernel) come in by direct
//     global -> LDS copies issued before the conversion of the chunk and waited for after it; the NEXT chunk's source loads are
//     issued only then, so nothing inside the tap loop ever waits for memory.
//   * Tap classes: a stride-2 data gradient / ConvTranspose2d is up to four stride-1 correlations (one per output parity) with
//     their own taps — grid.z selects the class; all classes share the source.
// Per chunk: [exchange the chunk max] barrier [rescale? convert + write the halo | weight copies land] barrier [all taps: LDS
// fragment reads + MFMAs, no memory waits; the NEXT chunk's source loads are in flight] barrier.
#include "common.h"
#include "conv_s16g.h"
#include "pack_plan.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BORDER_REFLECT = 1;
constexpr int ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3;      // (tanh: not served here, see nemar_s16g_plan)
constexpr int BWORDS = 3568;          // LDS words of the halo region: 4 (2 planes x 2 k groups) x HR x HCP <= BWORDS
constexpr int NS4LIM = 2;             // 4-pixel halo groups per loader thread (128 threads per k group): 16-byte source loads
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte load at a 4-byte-aligned address
constexpr int TEXP = 14;              // scaled magnitudes stay below 2^15

// v s = h + l (+ e): both halves of eight values as two 16-byte words.  v * s is exact (s a power of two), so fmaf(v, s, -h) is the
// exact residual — written as an fma so that hipcc selects v_fma_mix (fp16 operand read in place) instead of convert + multiply + subtract
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ float pow2f(int biased) {          // 2^(biased - 127); 0 below the normal range
    return biased < 1 ? 0.f : __builtin_bit_cast(float, (unsigned)(biased > 254 ? 254 : biased) << 23);
}
// biased exponent of the largest finite magnitude (bit pattern of a non-negative float), kept inside the range where both the
// scale 2^(TEXP + 127 - E) and its inverse are normal numbers
__device__ __forceinline__ int max_exponent(unsigned maxbits) {
    int e = (int)(maxbits >> 23);
    if (e < TEXP + 2) e = TEXP + 2;
    if (e > 254) e = 254;
    return e;
}
__device__ __forceinline__ float weight_scale(unsigned maxbits) { return pow2f(127 + TEXP + 127 - max_exponent(maxbits)); }

__device__ __forceinline__ int mirror_clamp(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
__device__ __forceinline__ void glds16(const u32x4* g, u32x4* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// max over the wavefront of a non-negative word, valid in lane 63 (DPP row shifts + row broadcasts: six VALU instructions; the
// __shfl_xor form goes through ds_bpermute — six LDS round trips per chunk)
__device__ __forceinline__ unsigned wave_max_to_lane63(unsigned x) {
#ifdef NEMAR_HOST_EMULATION
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = max(x, (unsigned)__shfl_xor((int)x, o, 64));
    return x;
#else
    // lanes a shift does not reach read `old` = 0: the neutral element of an unsigned max
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));      // row_shr:1
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));      // row_shr:2
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));      // row_shr:4
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));      // row_shr:8 -> lane 15 of every row
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1, 3
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));      // row_bcast:31 into rows 2, 3
    return x;
#endif
}

struct S16gParams {
    const float* src0; const float* src1; int C0, C1, Hs, Ws;
    const u32x4* wp; const unsigned* wmax;
    const float* bias; float* dst0; float* dst1; int M, M0, N;
    int OHf, OWf, osy, osx;
    int border, act; float slope;
    int ncls, xcd;
    int pair;                                // class-fused kernel: the two column parities of an output pixel pair are stored as one 8-byte word
    long long* tl;                           // NEMAR_TIMELINE builds: cycle stamps of one workgroup (tools/timeline_s16g.py)
    int dbg;                                 // ablation bits (nemar_tune key 2, tools/ only): 1 no tap loop, 2 no source loads, 4 no conversion
    int TW, RT, wshift, tiles_x, tiles_y, mblks, nchunks;
    int HR, HC, HCP, HCH, GPR, OFS, hp16, dymin, dxmin;     // GPR: 4-pixel groups per halo row; HCP = 4 GPR words per row; OFS: LDS column of
                                                            // image column ox0 SX (-dxmin rounded up to 4 when taps reach left of it, else 0)
    int aw16;                                // LDS words of the weight region
    FastDiv fd_gpr;
    long long cls_words;                     // packed words per class
    int ntaps[S16G_MAX_CLS], OH[S16G_MAX_CLS], OW[S16G_MAX_CLS], ooy[S16G_MAX_CLS], oox[S16G_MAX_CLS];
    int tapoff[S16G_MAX_TAPS];               // halo word offset of tap t of class c at [c * S16G_CLS_TAPS + t] (one class: all 64)
};

// max |w| of a weight tensor as a bit pattern, stage 1: ABSMAX_WGS workgroups, one partial each (plain stores — no zero-fill, no
// atomics); the pack kernel and the convolution take the maximum of the partials themselves
constexpr int ABSMAX_WGS = 64;       // (16 workgroups took 20-32 us on the 300-500 K element tensors: latency-bound)
__global__ __launch_bounds__(256) void s16g_absmax_kernel(const float* __restrict__ x, long long n, unsigned* out) {
    __shared__ unsigned red[4];
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += 256ll * ABSMAX_WGS) {
        const unsigned u = __builtin_bit_cast(unsigned, x[i]) & 0x7fffffffu;
        m = max(m, u < 0x7f800000u ? u : 0u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}
__device__ __forceinline__ unsigned absmax_of_partials(const unsigned* part) {
    const u32x4* const q = reinterpret_cast<const u32x4*>(part);
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < ABSMAX_WGS / 4; ++i) {
        const u32x4 v = q[i];
        m = max(max(m, v[0]), max(max(v[1], v[2]), v[3]));
    }
    return m;
}

// packed weights of class c: word (((chunk * mblks + mblk) * ntaps[c] + tap) * 2 + plane) * 2 + kg) * MB + m behind c * cls_words; element j
// of the word = reduction channel chunk * 16 + kg * 8 + j.  One launch for all classes (grid.y = class).
__global__ __launch_bounds__(256) void s16g_pack_kernel(const float* __restrict__ w, u32x4* __restrict__ out, int M, int Cred, int MB,
                                                        int mblks, int nchunks, long long wsm, long long wsc, long long cls_words,
                                                        const unsigned* maxbits, S16gProblem q) {
    __shared__ int s_wofs[S16G_MAX_TAPS];
    const int cls = blockIdx.y, ntaps = q.ntaps[cls];
    for (int i = threadIdx.x; i < S16G_MAX_TAPS; i += blockDim.x) s_wofs[i] = i < ntaps ? q.wofs[cls][i] : 0;
    __syncthreads();
    const float scale = weight_scale(absmax_of_partials(maxbits));
    out += (size_t)cls * cls_words;
    const long long total = (long long)nchunks * mblks * ntaps * 2 * MB;          // one thread = both planes of a word
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(t % MB);
        long long r = t / MB;
        const int kg = (int)(r & 1);
        r >>= 1;
        const int tap = (int)(r % ntaps);
        r /= ntaps;
        const int mblk = (int)(r % mblks), chunk = (int)(r / mblks);
        const int mg = mblk * MB + m;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cr = chunk * 16 + kg * 8 + j;
            v[j] = (mg < M && cr < Cred) ? w[(long long)mg * wsm + (long long)cr * wsc + s_wofs[tap]] : 0.f;
        }
        u32x4 hi, lo;
        split8(v, scale, hi, lo);
        u32x4* const o = out + ((((long long)chunk * mblks + mblk) * ntaps + tap) * 4 + kg) * MB + m;
        o[0] = hi;
        o[2 * MB] = lo;
    }
}

// the same pack as a job of a weight-pack plan (pack_plan.h): arguments from device memory, grid (blocks, class, job)
struct S16gPackArgs {
    const float* w; u32x4* out;
    int M, Cred, MB, mblks, nchunks;
    long long wsm, wsc, cls_words;
    const unsigned* maxbits;
    int gx, gy;
    int ntaps[S16G_MAX_CLS];
    int wofs[S16G_MAX_CLS][S16G_MAX_TAPS];
};
__device__ __forceinline__ void s16g_pack_body(const S16gPackArgs& a, int bx, int cls, int gx) {
    const int ntaps = a.ntaps[cls];
    const float scale = weight_scale(absmax_of_partials(a.maxbits));
    u32x4* const out = a.out + (size_t)cls * a.cls_words;
    const long long total = (long long)a.nchunks * a.mblks * ntaps * 2 * a.MB;
    for (long long t = (long long)bx * 256 + threadIdx.x; t < total; t += (long long)gx * 256) {
        const int m = (int)(t % a.MB);
        long long r = t / a.MB;
        const int kg = (int)(r & 1);
        r >>= 1;
        const int tap = (int)(r % ntaps);
        r /= ntaps;
        const int mblk = (int)(r % a.mblks), chunk = (int)(r / a.mblks);
        const int mg = mblk * a.MB + m;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cr = chunk * 16 + kg * 8 + j;
            v[j] = (mg < a.M && cr < a.Cred) ? a.w[(long long)mg * a.wsm + (long long)cr * a.wsc + a.wofs[cls][tap]] : 0.f;
        }
        u32x4 hi, lo;
        split8(v, scale, hi, lo);
        u32x4* const o = out + ((((long long)chunk * a.mblks + mblk) * ntaps + tap) * 4 + kg) * a.MB + m;
        o[0] = hi;
        o[2 * a.MB] = lo;
    }
}
NEMAR_PACK_MULTI(s16g_pack_multi_kernel, S16gPackArgs, s16g_pack_body, 256)
void s16g_pack_multi(const void* jobs, int njobs, int gx, int gy, hipStream_t st) {
    hipLaunchKernelGGL(s16g_pack_multi_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, (const S16gPackArgs*)jobs);
}
struct RegS16gPack {
    RegS16gPack() { nemar_pack_register(PACK_FAM_S16G, sizeof(S16gPackArgs), s16g_pack_multi); }
} g_reg_s16g_pack;

// MT x 32 output channels, NT x 32 pixels per wave (four waves side by side in the pixel direction); SX = source stride.
// LDS (dynamic: exactly what the layer needs, so that narrow layers keep several workgroups per CU): [weights of one chunk,
// p.aw16 words][halo planes, 4 p.hp16 words]
// MBL (round 6): channel blocks per workgroup.  The grid used to hold one workgroup per (tile, channel block): a layer with 128 / 256
// output channels loaded, reduced and converted every halo chunk two / four times, and those phases — not the MFMAs — are most of a
// chunk (timeline: max + barrier 3000, conversion 3300, loads + barrier 1500 cycles beside 3000 cycles of taps).  With MBL > 1 a
// workgroup keeps MBL accumulator sets and runs the taps of MBL channel blocks on ONE converted halo; their weights alternate between two
// LDS regions (the next block's copies are issued before the current block's taps).
// CF (round 6, measurement build only — it LOST): CLASS-FUSED form of a four-class problem (the output-parity classes of a stride-2 data gradient / ConvTranspose2d: 1 + 2 + 2 + 4
// taps of a 3x3 filter, 4 x 4 of a 4x4 one).  One workgroup per (tile, class) loaded, reduced and converted the SAME halo four times for
// nine taps' worth of MFMAs — those phases, not the taps, are most of a chunk (above).  With CF a workgroup keeps one accumulator set per
// class and runs all classes' taps on ONE converted halo; the chunk's weights of the four classes sit side by side in the one weight
// region (as many words as a plain 3x3 / 4x4 layer's), and the epilogue stores the two column parities of a pixel pair as one 8-byte word
// (a class's own stores are 4 bytes every 8).  Measured at batch 16 / 24 (profiles/r6_s16g_class_fused.txt): the translation net's 64 <- 128 stride-2
// data gradient 309 -> 345 us, 128 <- 256: 261 -> 281 us, the discriminator's 64 <- 128 4x4: 118 -> 153 us.  Four accumulator sets of a
// 64-row tile take 332 registers: ONE workgroup per CU, and what hides a workgroup's load / exchange / convert phases is the OTHER
// workgroups of its CU, not fewer conversions.  (The 32-row form, 204 registers, is even: 163 vs 165 us.)
template <int MT, int NT, int SX, int NS4MAX, int MBL = 1, int CF = 0>          // NS4MAX: 4-pixel halo groups per loader thread (1: tiles of <= 128 groups)
__global__ __launch_bounds__(256) void s16g_kernel(S16gParams p) {
    static_assert(!CF || (MBL == 1 && NT == 1 && SX == 1), "class-fused form: four accumulator sets of one 128-pixel tile, stride-1 source");
    constexpr int MB = 32 * MT, NPW = 32 * NT, NSET = CF ? 4 : MBL;
#ifdef NEMAR_HOST_EMULATION
    __shared__ __attribute__((aligned(16))) u32x4 smem[49 * 4 * 32 * 4 + BWORDS];      // (the emulator has no dynamic LDS)
#else
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
#endif
    __shared__ unsigned red[4];
    u32x4* const As = smem;
    u32x4* const Bs = smem + (MBL > 1 ? 2 : 1) * p.aw16;          // MBL > 1: two weight regions

    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    // 1-D grid; consecutive workgroup ids sit on consecutive XCDs: give every XCD a contiguous run of (tile, class, channel block)
    // triples, so that the channel blocks / classes of a tile (same halo) and neighbouring tiles (shared halo rows) share one L2
    int t = blockIdx.x;
    if (p.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);
    const int mblk = (t % (p.mblks / MBL)) * MBL;              // first channel block of this workgroup
    t /= p.mblks / MBL;
    const int cls = CF ? 0 : t % p.ncls;               // (CF: class 0 = parity (0, 0) has the largest extents — the tile test below)
    if (!CF) t /= p.ncls;
    const int txi = t % p.tiles_x;
    t /= p.tiles_x;
    const int tyi = t % p.tiles_y, n = t / p.tiles_y;
    const int oy0 = tyi * p.RT, ox0 = txi * p.TW;
    const int ntaps = CF ? p.ntaps[0] + p.ntaps[1] + p.ntaps[2] + p.ntaps[3] : p.ntaps[cls];      // CF: all classes' taps, class after class
    const int OH = p.OH[cls], OW = p.OW[cls];
    if (oy0 >= OH || ox0 >= OW) return;                    // (classes of an odd-sized plane differ by one row / column of tiles)
    const int C = p.C0 + p.C1;
    const size_t HWs = (size_t)p.Hs * p.Ws;

    // ---- loader role: waves 0, 1 fill k group 0 (channels 0..7 of the chunk), waves 2, 3 k group 1; thread = FOUR consecutive IMAGE
    // columns 4a .. 4a + 3 of a halo row, fetched with one aligned 16-byte load per channel (round 4: the vector-memory pipe issues
    // about one wave-wide load per 30 clocks and CU whatever its width — profiles/r4_load_width.txt — and 56 four-byte loads per thread
    // and chunk sat at that cap).  Groups are aligned to the IMAGE (Ws % 4 == 0), so no load straddles a border: a group is inside or
    // outside as a whole.  Padding positions of the halo are never loaded: under a zero border they keep the zeros the region is
    // filled with once; under a reflect border the owners of the first / last group of a row also write their words to the <= 3
    // mirrored positions beside them (rows are mirrored by address).  LDS column of image column ix = ix - ox0 SX + OFS.
    const int kgl = wid >> 1;
    const int ngr = p.HR * p.GPR;
    const int ns = __builtin_amdgcn_readfirstlane((ngr + 127) >> 7);          // group slots in use (of NS4MAX)
    int soff[NS4MAX], lpos[NS4MAX];    // soff: source offset of the group inside a channel image; lpos < 0: nothing to load (padding / no group)
    int mirr[NS4MAX];                  // reflect border: bits 0..1 = columns mirrored to the left of group 0, bits 2..3 = to the right of the last group
    bool smirr[NS4MAX];                // some lane of the wave has mirror columns to write (wave-uniform)
    bool sdead[NS4MAX];                // some lane of the wave has no group in this slot (wave-uniform)
#pragma unroll
    for (int i = 0; i < NS4MAX; ++i) {
        const int gp = (tid & 127) + 128 * i;
        soff[i] = 0;
        lpos[i] = -1;
        mirr[i] = 0;
        if (gp < ngr) {
            const int hr = (int)fd_div((unsigned)gp, p.fd_gpr), g = gp - hr * p.GPR;
            int iy = oy0 * SX + p.dymin + hr;
            const int ixg = ox0 * SX - p.OFS + 4 * g;
            bool ok = ixg >= 0 && ixg < p.Ws;
            if (p.border == BORDER_REFLECT) iy = mirror_clamp(iy, p.Hs);
            else ok = ok && (unsigned)iy < (unsigned)p.Hs;
            if (ok) {
                soff[i] = iy * p.Ws + ixg;
                lpos[i] = kgl * p.hp16 + hr * p.HCP + (SX == 2 ? 2 * g : 4 * g);
                if (p.border == BORDER_REFLECT) {
                    // halo columns left of the image: -1 .. ox0 SX + dxmin (tiles in the first column); right: Ws .. last halo column
                    const int nl = ixg == 0 ? min(3, -(ox0 * SX + p.dxmin)) : 0;
                    const int nr = ixg == p.Ws - 4 ? min(3, ox0 * SX - p.OFS + 4 * p.GPR - p.Ws) : 0;
                    mirr[i] = max(nl, 0) | (max(nr, 0) << 2);
                }
            }
        }
        smirr[i] = __any(mirr[i] != 0) != 0;
        sdead[i] = __any(lpos[i] < 0) != 0;
    }
    // zero fill of the halo planes (padding positions are never written afterwards; the barrier of the first chunk orders it)
    for (int w = tid; w < 4 * p.hp16; w += 256) Bs[w] = u32x4{0u, 0u, 0u, 0u};
    f32x4u v[NS4MAX][8];
    // source values of chunk `ch_`: 8 channels (wave-uniform base pointers) x this thread's groups
#define S16G_LOAD(ch_)                                                                                                  \
    {                                                                                                                   \
        const float* cb_[8];                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                 \
            const int c_ = (ch_) * 16 + kgl * 8 + j;                                                                    \
            const float* b_ = c_ < p.C0 ? p.src0 + ((size_t)n * p.C0 + c_) * HWs                                        \
                                        : p.src1 + ((size_t)n * p.C1 + (c_ - p.C0)) * HWs;                              \
            cb_[j] = c_ < C ? b_ : p.src0;      /* beyond the last channel: any valid address, zeroed below */          \
        }                                                                                                               \
        /* slots beyond the tile's halo (ns of NS4MAX: wave-uniform) issue nothing */                                   \
        _Pragma("unroll") for (int i = 0; i < NS4MAX; ++i) {                                                            \
            if (i < ns) {                                                                                               \
                if (p.dbg & 2) { _Pragma("unroll") for (int j = 0; j < 8; ++j) v[i][j] = f32x4u{1.f, 1.f, 1.f, 1.f}; }  \
                else { _Pragma("unroll") for (int j = 0; j < 8; ++j) v[i][j] = *reinterpret_cast<const f32x4u*>(cb_[j] + soff[i]); } /* unconditional per lane */ \
            }                                                                                                           \
        }                                                                                                               \
    }
    // lanes without a group loaded from offset 0 (their values take part in nothing but must not reach the maximum); channels
    // beyond C become zero (wave-uniform test: layers with C % 16 == 0 skip it)
#define S16G_MASK(ch_)                                                                                                  \
    {                                                                                                                   \
        const int jlim_ = C - (ch_) * 16 - kgl * 8;                                                                     \
        _Pragma("unroll") for (int i = 0; i < NS4MAX; ++i) {                                                            \
            if (i < ns && (sdead[i] || jlim_ < 8)) {                                                                    \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                         \
                    if (lpos[i] < 0 || j >= jlim_) v[i][j] = f32x4u{0.f, 0.f, 0.f, 0.f};                                \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
    }
    // this wave's share of the chunk's packed weights: 1 KiB copies wid, wid + 4, ...
    const int acopies = ntaps * 2 * MT;                        // ntaps * 4 * MB / 64
    const u32x4* const wcls = p.wp + (size_t)cls * p.cls_words + (size_t)mblk * ntaps * 4 * MB;
    const size_t wchunk = (size_t)p.mblks * ntaps * 4 * MB;
#define S16G_WEIGHTS(ch_, mb_)                                        /* block mb_ of the workgroup -> region mb_ & 1 */ \
    { if (CF) {                                           /* the four classes' runs of this chunk, side by side */      \
        int pre_ = 0;                                                                                                   \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                                              \
            const int nt_ = p.ntaps[c_];                                                                                \
            const u32x4* const a_ = p.wp + (size_t)c_ * p.cls_words + ((size_t)(ch_) * p.mblks + mblk) * (nt_ * 4 * MB) + lane; \
            u32x4* const d_ = As + pre_ * 4 * MB;                                                                       \
            for (int q = wid; q < nt_ * 2 * MT; q += 4) glds16(a_ + 64 * q, d_ + 64 * q);                               \
            pre_ += nt_;                                                                                                \
        }                                                                                                               \
    } else {                                                                                                            \
        const u32x4* const a_ = wcls + (size_t)(ch_) * wchunk + (size_t)(mb_) * (ntaps * 4 * MB) + lane;                \
        u32x4* const d_ = As + ((mb_) & 1) * p.aw16;                                                                    \
        if (!(p.dbg & 8)) for (int q = wid; q < acopies; q += 4) glds16(a_ + 64 * q, d_ + 64 * q);                      \
    } }

    // ---- MFMA role ----
    int bbase[NT], oyx[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int px = wid * NPW + 32 * nt + l31;
        const int ty = px >> p.wshift, tx = px & (p.TW - 1);
        bbase[nt] = lhi * p.hp16 + ty * SX * p.HCP + tx;
        oyx[nt] = ((oy0 + ty) << 16) | (ox0 + tx);
    }
    const int abase = lhi * MB + l31;
    f32x16 acc[NSET][MT][NT];
#pragma unroll
    for (int mb = 0; mb < NSET; ++mb)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][mt][nt][r] = 0.f;

    int E = 0;                                             // running biased exponent of the tile's source maximum
    // halo word offsets of this class's taps, lane t = tap t: the tap loop takes them with v_readlane.  (Round 6: as p.tapoff[...] the
    // offset was an s_load_dword PER TAP — scalar memory shares the lgkm counter with LDS, so the s_waitcnt lgkmcnt(0) behind it also
    // drained the fragment reads in flight: ~250 exposed cycles per tap beside 192 - 384 cycles of MFMAs.)
#ifndef NEMAR_HOST_EMULATION
    const int tapv = p.tapoff[min(cls * S16G_CLS_TAPS + lane, S16G_MAX_TAPS - 1)];       // (CF: cls == 0, the host wrote the classes' taps back to back)
#define S16G_TAPOFF(tap_) __builtin_amdgcn_readlane(tapv, (tap_))
#else
#define S16G_TAPOFF(tap_) p.tapoff[cls * S16G_CLS_TAPS + (tap_)]
#endif
#ifdef NEMAR_TIMELINE
#define S16G_STAMP(i_) if (p.tl != nullptr && blockIdx.x == 8 && lane == 0 && chunk < 8) p.tl[(wid * 8 + chunk) * 8 + (i_)] = clock64();
#else
#define S16G_STAMP(i_)
#endif
    S16G_LOAD(0)
    for (int chunk = 0; chunk < ((p.dbg & 64) ? 0 : p.nchunks); ++chunk) {
        // -- the chunk's maximum over the four waves (its loads were issued a whole chunk ago) --
        S16G_STAMP(0)
        S16G_MASK(chunk)
        if (!(p.dbg & 16)) {
            // |v| as floats: one v_max3_f32 per two elements (fmaxf ignores NaN; an infinity sends this thread through the slow
            // path that leaves non-finite values out of the maximum — they must not flush the finite ones)
            float mf = 0.f;
#pragma unroll
            for (int i = 0; i < NS4MAX; ++i)
                if (i < ns) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        mf = fmaxf(fmaxf(mf, fmaxf(__builtin_fabsf(v[i][j][0]), __builtin_fabsf(v[i][j][1]))),
                                   fmaxf(__builtin_fabsf(v[i][j][2]), __builtin_fabsf(v[i][j][3])));
                }
            unsigned m = __builtin_bit_cast(unsigned, mf);
            if (m >= 0x7f800000u) {
                m = 0;
#pragma unroll
                for (int i = 0; i < NS4MAX; ++i)
                    if (i < ns) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned u = __builtin_bit_cast(unsigned, (float)v[i][j][e]) & 0x7fffffffu;
                                m = max(m, u < 0x7f800000u ? u : 0u);
                            }
                    }
            }
            m = wave_max_to_lane63(m);
            if (lane == 63) red[wid] = m;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): the max word is written
        if (!(p.dbg & 32)) __builtin_amdgcn_s_barrier();   // (also: every wave has left the previous chunk's tap loop)
        S16G_STAMP(1)
        S16G_WEIGHTS(chunk, 0)                             // land during the conversion below
        {
            const unsigned m = max(max(red[0], red[1]), max(red[2], red[3]));
            const int e = max_exponent(m);
            if (e > E) {
                if (chunk > 0) {
                    const float f = pow2f(127 + E - e);    // exact (power of two); accumulators far below the new scale flush
#pragma unroll
                    for (int mb = 0; mb < NSET; ++mb)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[mb][mt][nt][r] *= f;
                }
                E = e;
            }
        }
        {
            const float scale = pow2f(127 + TEXP + 127 - E);
#pragma unroll
            for (int i = 0; i < NS4MAX; ++i) {
                if (lpos[i] < 0 || (p.dbg & 4)) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {              // pixel e of the group: its eight channels as one hi and one lo word
                    float t8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t8[j] = v[i][j][e];
                    u32x4 hi, lo;
                    split8(t8, scale, hi, lo);
                    const int pos = lpos[i] + (SX == 2 ? (e & 1) * p.HCH + (e >> 1) : e);      // stride 2: de-interleaved by column parity
                    Bs[pos] = hi;
                    Bs[2 * p.hp16 + pos] = lo;
                    if (smirr[i]) {
                        // reflect: column -c mirrors column c (group 0, element c -> LDS column OFS - c), column Ws - 1 + c mirrors
                        // column Ws - 1 - c (last group, element 3 - c -> LDS column of the group + 3 + c)
                        const int nl = mirr[i] & 3, nr = mirr[i] >> 2;
                        if (e >= 1 && e <= nl) {
                            const int col = -e;                                                 // relative to the group's first column
                            const int q = lpos[i] + (SX == 2 ? (col & 1) * p.HCH + (col >> 1) : col);
                            Bs[q] = hi;
                            Bs[2 * p.hp16 + q] = lo;
                        }
                        if (e <= 2 && 3 - e <= nr) {
                            const int col = 3 + (3 - e);
                            const int q = lpos[i] + (SX == 2 ? (col & 1) * p.HCH + (col >> 1) : col);
                            Bs[q] = hi;
                            Bs[2 * p.hp16 + q] = lo;
                        }
                    }
                }
            }
        }
        // the weight copies are the only vector-memory operations in flight here: wait for them, THEN issue the next chunk's source
        // loads (nothing waits for those until the top of the next iteration: they have the whole tap loop to arrive)
        S16G_STAMP(2)
        wait_vmem();
        S16G_STAMP(3)
        if (MBL > 1) S16G_WEIGHTS(chunk, 1)                // the second block's weights: in flight during the first block's taps
        if (chunk + 1 < p.nchunks) S16G_LOAD(chunk + 1)
        __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): this wave's halo words are written
        __builtin_amdgcn_s_barrier();                      // halo planes + weights of this chunk are in LDS for every wave
        S16G_STAMP(4)
        // fragments of tap t + 1 are read while the MFMAs of tap t issue (two register sets)
        u32x4 fa[2][MT][2], fb[2][NT][2];
#define S16G_READ(set_, tap_, mb_)                                                                                      \
        {                                                                                                               \
            const int to_ = S16G_TAPOFF(tap_);                                                                          \
            const u32x4* const A_ = As + ((mb_) & 1) * p.aw16;                                                          \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                           \
                _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) fb[set_][nt][pl] = Bs[pl * 2 * p.hp16 + bbase[nt] + to_]; \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
                _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) fa[set_][mt][pl] = A_[((tap_) * 2 + pl) * 2 * MB + abase + mt * 32]; \
        }
        // partial products, smallest first: (l h') (h l') (h h')
#define S16G_MMA(set_, mb_)                                                                                             \
        _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                                   \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                       \
                    acc[mb_][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set_][mt][q == 0 ? 1 : 0]), \
                                                                              __builtin_bit_cast(f16x8, fb[set_][nt][q == 1 ? 1 : 0]), \
                                                                              acc[mb_][mt][nt], 0, 0, 0);
        if constexpr (CF != 0) {
            // class after class on the one converted halo: set c = class c's accumulators, taps g0 .. g0 + ntaps[c] - 1 of the flat table
            int g0 = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int nt_c = p.ntaps[c];
                S16G_READ(0, g0, 0)
                for (int tap = 0; tap + 1 < nt_c && !(p.dbg & 1); tap += 2) {
                    S16G_READ(1, g0 + tap + 1, 0)
                    S16G_MMA(0, c)
                    if (tap + 2 < nt_c) S16G_READ(0, g0 + tap + 2, 0)
                    S16G_MMA(1, c)
                }
                if (nt_c & 1) { S16G_MMA(0, c) }
                g0 += nt_c;
            }
        } else {
#pragma unroll
        for (int mb = 0; mb < MBL; ++mb) {
            if (mb > 0) {
                // block mb's weights (issued before block mb - 1's taps) have landed for every wave; region (mb + 1) & 1 is free again
                wait_vmem();
                __builtin_amdgcn_s_barrier();
                if (mb + 1 < MBL) S16G_WEIGHTS(chunk, mb + 1)
            }
            S16G_READ(0, 0, mb)
            for (int tap = 0; tap + 1 < ntaps && !(p.dbg & 1); tap += 2) {
                S16G_READ(1, tap + 1, mb)
                S16G_MMA(0, mb)
                if (tap + 2 < ntaps) S16G_READ(0, tap + 2, mb)
                S16G_MMA(1, mb)
            }
            if (ntaps & 1) { S16G_MMA(0, mb) }
        }
        }
        S16G_STAMP(5)
#undef S16G_READ
#undef S16G_MMA
        // (the barrier at the top of the next chunk keeps the LDS regions until every wave is done with them)
    }
#undef S16G_LOAD
#undef S16G_MASK
#undef S16G_WEIGHTS
#undef S16G_TAPOFF

    // ---- epilogue: take the two power-of-two scales out (exact), bias, activation ----
    if (p.dbg & 128) return;
    const float u1 = pow2f(E - TEXP), u2 = 1.f / weight_scale(absmax_of_partials(p.wmax));
    const size_t plane = (size_t)p.OHf * p.OWf;
    const int M1 = p.M - p.M0;
    const float u12 = u1 * u2;                               // (both powers of two; their product is a normal number for any finite result)
    const bool one_mul = u12 != 0.f && u12 < 3.0e38f;
    if constexpr (CF != 0) {
        // (the plan offers this form only for full channel blocks into one destination each and classes (ph, 0), (ph, 1) of equal extents:)
        // element pairs (2 ox, 2 ox + 1) of destination row 2 oy + ph as one aligned 8-byte store (p.pair: the host checked the alignment;
        // otherwise two 4-byte stores)
        {
            float* const dplain = mblk * MB >= p.M0 ? p.dst1 + ((size_t)n * M1 + (mblk * MB - p.M0)) * plane : p.dst0 + ((size_t)n * p.M0 + mblk * MB) * plane;
            const int oy = oyx[0] >> 16, ox = oyx[0] & 0xffff;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                if (oy >= p.OH[2 * ph] || ox >= p.OW[2 * ph]) continue;
                float* const d0 = dplain + (size_t)(oy * 2 + ph) * p.OWf + (size_t)(ox * 2) + (size_t)(4 * lhi) * plane;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float bv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) bv[r] = p.bias ? p.bias[mblk * MB + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float o0 = (one_mul ? acc[2 * ph][mt][0][r] * u12 : (acc[2 * ph][mt][0][r] * u1) * u2) + bv[r];
                        float o1 = (one_mul ? acc[2 * ph + 1][mt][0][r] * u12 : (acc[2 * ph + 1][mt][0][r] * u1) * u2) + bv[r];
                        if (p.act == ACT_RELU) { o0 = fmaxf(o0, 0.f); o1 = fmaxf(o1, 0.f); }
                        else if (p.act == ACT_LRELU) { o0 = o0 > 0.f ? o0 : o0 * p.slope; o1 = o1 > 0.f ? o1 : o1 * p.slope; }
                        float* const d = d0 + (size_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * plane;
                        if (p.pair) *reinterpret_cast<float2*>(d) = make_float2(o0, o1);
                        else { d[0] = o0; d[1] = o1; }
                    }
                    __builtin_amdgcn_sched_barrier(0);     // (one group of 32 accumulators at a time out of the accumulator file)
                }
            }
        }
    } else {
#pragma unroll
    for (int mb = 0; mb < NSET; ++mb) {                    // (MBL > 1: the channel blocks of this workgroup, one after the other; CF: its classes)
    const int mbk = CF ? mblk : mblk + mb;
    const int ec = CF ? mb : cls;
    const int OHe = p.OH[ec], OWe = p.OW[ec];
        float bv[MT][16];                  // the bias of this lane's rows: all loads in flight at once (one wait, not one per row)
    #pragma unroll
        for (int mt = 0; mt < MT; ++mt)
    #pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbk * MB + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                bv[mt][r] = p.bias ? p.bias[m < p.M ? m : 0] : 0.f;
            }
        // full channel block into one destination (every layer of the three nets; ragged blocks below): plain stores, the activation
        // chosen once per wave — three compact store loops (an inlined tanhf per element made this epilogue 30 000 instructions and
        // instruction-fetch bound: 24 of 160 us; tanh layers have <= 4 output channels and never come here)
        const bool plain = (mbk + 1) * MB <= p.M && ((mbk + 1) * MB <= p.M0 || mbk * MB >= p.M0);
        float* const dplain = mbk * MB >= p.M0 ? p.dst1 + ((size_t)n * M1 + (mbk * MB - p.M0)) * plane : p.dst0 + ((size_t)n * p.M0 + mbk * MB) * plane;
    #pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int oy = oyx[nt] >> 16, ox = oyx[nt] & 0xffff;
            if (oy >= OHe || ox >= OWe) continue;
            const size_t opix = (size_t)(oy * p.osy + p.ooy[ec]) * p.OWf + (size_t)(ox * p.osx + p.oox[ec]);
            if (plain) {
                float* const d0 = dplain + opix + (size_t)(4 * lhi) * plane;
    #define S16G_STORES(EXPR_)                                                                                              \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                        \
                        const float o_ = (one_mul ? acc[mb][mt][nt][r] * u12 : (acc[mb][mt][nt][r] * u1) * u2) + bv[mt][r];         \
                        d0[(size_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * plane] = (EXPR_);                                   \
                    }
                if (p.act == ACT_RELU) { S16G_STORES(fmaxf(o_, 0.f)) }
                else if (p.act == ACT_LRELU) { S16G_STORES(o_ > 0.f ? o_ : o_ * p.slope) }
                else { S16G_STORES(o_) }
    #undef S16G_STORES
                continue;
            }
    #pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbk * MB + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (m >= p.M) continue;
                    float o = (acc[mb][mt][nt][r] * u1) * u2 + bv[mt][r];
                    o = p.act == ACT_RELU ? fmaxf(o, 0.f) : (p.act == ACT_LRELU ? (o > 0.f ? o : o * p.slope) : o);
                    float* const d = m < p.M0 ? p.dst0 + ((size_t)n * p.M0 + m) * plane : p.dst1 + ((size_t)n * M1 + (m - p.M0)) * plane;
                    d[opix] = o;
                }
            }
        }
    }
    }
}

int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

// ---- measurement hook: HIP events around every launch while enabled (bench.py roofline; shares the read-out with conv_split16) ----
constexpr int MAX_TIMED = 2048;
hipEvent_t g_tev[MAX_TIMED][2];
int g_tev_made = 0, g_tev_used = 0;
double g_tev_flop = 0.0;
bool g_timing = false;

}  // namespace

void nemar_s16g_timer(int on) {
    g_timing = on != 0;
    if (on) { g_tev_used = 0; g_tev_flop = 0.0; }
}

int nemar_s16g_timer_read(double* total_ms, double* total_flop) {
    double t = 0.0;
    for (int i = 0; i < g_tev_used; ++i) {
        float ms = 0.f;
        (void)hipEventSynchronize(g_tev[i][1]);
        (void)hipEventElapsedTime(&ms, g_tev[i][0], g_tev[i][1]);
        t += ms;
    }
    *total_ms = t;
    *total_flop = g_tev_flop;
    const int n = g_tev_used;
    g_tev_used = 0;
    g_tev_flop = 0.0;
    return n;
}

static NEMAR_SWITCH(int, g_s16g_maxmt, 2);      // widest channel tile (x 32): nemar_s16g_tune(0, v).  64 channels: the step is 1.3 % faster than with 128-channel
                                   // tiles (36.4-36.7 vs 37.0-37.2 ms, A/B on one box) — their fragment sets leave no room for latency hiding
static NEMAR_SWITCH(int, g_s16g_lds_pref, 0);   // prefer pixel tiles that leave room for two workgroups per CU: nemar_s16g_tune(1, v)
static NEMAR_SWITCH(int, g_s16g_mbl_wgs, 256);  // ... as long as the grid keeps this many workgroups: nemar_s16g_tune(3, v) (tests: 0)
static NEMAR_SWITCH(int, g_s16g_cf, 0);         // fuse the four parity classes of a stride-2 data gradient / ConvTranspose2d into one workgroup per tile: nemar_s16g_tune(4, v)
                                                // (0 off = default: MEASURED SLOWER, below; 1 where the grid keeps >= min(192, key 3) workgroups, 2 = 1 + refuse four-class
                                                // problems that do not fuse: tests).  Measurement build only.
static NEMAR_SWITCH(int, g_s16g_maxmbl, 4);     // most channel blocks per workgroup (1: one workgroup per channel block, the rounds 3-5 form): nemar_s16g_tune(2, v)
#ifdef NEMAR_AB
void nemar_s16g_tune(int key, int value) {
    if (key == 0) g_s16g_maxmt = value == 1 || value == 4 ? value : 2;
    if (key == 1) g_s16g_lds_pref = value;
    if (key == 2) g_s16g_maxmbl = value >= 4 ? 4 : 1;
    if (key == 3) g_s16g_mbl_wgs = value;
    if (key == 4) g_s16g_cf = value;
}
#endif

S16gPlan nemar_s16g_plan(const S16gProblem& q) {
    S16gPlan pl;
    pl.ok = 0;
    pl.CF = 0;
    const int C = q.C0 + q.C1;
    if (q.ncls < 1 || q.ncls > S16G_MAX_CLS || (q.sstride != 1 && q.sstride != 2) || q.M < 5 || C < 1) return pl;       // (<= 4 rows: the narrow VALU kernels)
    if (q.ncls > 1 && q.sstride != 1) return pl;
    if (q.act == ACT_TANH) return pl;
    int maxtaps = 0, dymin = 1 << 20, dymax = -(1 << 20), dxmin = 1 << 20, dxmax = -(1 << 20), OH = 0, OW = 0;
    for (int c = 0; c < q.ncls; ++c) {
        if (q.ntaps[c] < 1 || q.ntaps[c] > (q.ncls > 1 ? S16G_CLS_TAPS : S16G_MAX_TAPS)) return pl;
        if (q.ntaps[c] > maxtaps) maxtaps = q.ntaps[c];
        for (int t = 0; t < q.ntaps[c]; ++t) {
            if (q.dy[c][t] < dymin) dymin = q.dy[c][t];
            if (q.dy[c][t] > dymax) dymax = q.dy[c][t];
            if (q.dx[c][t] < dxmin) dxmin = q.dx[c][t];
            if (q.dx[c][t] > dxmax) dxmax = q.dx[c][t];
        }
        if (q.OH[c] > OH) OH = q.OH[c];
        if (q.OW[c] > OW) OW = q.OW[c];
    }
    if (OH < 1 || OW < 24 || OH >= 32768 || OW >= 32768) return pl;
    if (q.Ws % 4 != 0 || q.Ws < 8) return pl;                                       // 16-byte source loads at image-aligned 4-column groups
    if (q.border == BORDER_REFLECT && (dxmin < -3 || dxmax > 3)) return pl;         // (mirrored columns come from the first / last group)
    // channel tile: as wide as the layer, but the LDS must hold a chunk's weights for every tap (ATAPS x MB x 64 B) next to the halo
    pl.MT = q.M <= 32 ? 1 : (q.M <= 64 ? 2 : 4);
    if (pl.MT > g_s16g_maxmt) pl.MT = g_s16g_maxmt;
    if (maxtaps > 9 && pl.MT == 4) pl.MT = 2;
    if (maxtaps > 16 && pl.MT == 2) pl.MT = 1;
    pl.ATAPS = maxtaps;
    if (maxtaps > 49) return pl;
    const int MB = 32 * pl.MT;
    pl.mblks = (q.M + MB - 1) / MB;
    pl.nchunks = (C + 15) / 16;
    // pixel tile: 128 NT pixels as RT rows x TW columns; the smallest halo that fits wins
    const int sx = q.sstride, ey = dymax - dymin, ex = dxmax - dxmin;
    // class-fused form (s16g_kernel<..., CF = 1>): the four parity classes of a stride-2 data gradient on one converted halo — one 128-pixel
    // tile, four accumulator sets, all classes' taps of a chunk in the one weight region
    int alltaps = 0;
    for (int c = 0; c < q.ncls; ++c) alltaps += q.ntaps[c];
    bool cf = g_s16g_cf && q.ncls == 4 && sx == 1 && pl.MT <= 2 && alltaps <= 16 && q.osy == 2 && q.osx == 2 && q.border != BORDER_REFLECT &&
              q.M % MB == 0 && (q.M0 >= q.M || q.M0 % MB == 0);            // full channel blocks, each into one destination
    for (int c = 0; cf && c < 4; ++c) cf = q.ooy[c] == (c >> 1) && q.oox[c] == (c & 1) && q.OH[c] <= q.OH[0] && q.OW[c] <= q.OW[0];
    cf = cf && q.OW[0] == q.OW[1] && q.OW[2] == q.OW[3] && q.OH[0] == q.OH[1] && q.OH[2] == q.OH[3];      // (even destination extents)
    long long best = -1;
  for (int attempt = cf ? 0 : 1; attempt < 2 && best < 0; ++attempt) {
    cf = attempt == 0;
    const int wtaps = cf ? alltaps : maxtaps, wcls = cf ? 1 : q.ncls;
    // (128 channels x 256 pixels per workgroup — MT 4, NT 2 — needs 280 VGPRs: 144 of them spilled to scratch; that tile is not offered)
#ifdef NEMAR_S16G_NT1          /* variant build for an A/B run (tools/build_variant.py): 128-pixel tiles only */
    for (int NT = 1; NT >= 1; --NT) {
#else
    for (int NT = (sx == 2 || pl.MT == 4 || cf ? 1 : 2); NT >= 1; --NT) {
#endif
        const int NP = 128 * NT;
        for (int TW = 32; TW <= NP; TW *= 2) {
            if (TW > 32 && TW / 2 >= OW) break;              // wider than the rows: pure waste
            const int RT = NP / TW;
            const int HR = (RT - 1) * sx + ey + 1, HC = (TW - 1) * sx + ex + 1;
            // rows are loaded as groups of 4 image columns starting OFS (a multiple of 4) columns left of the tile when taps reach there: the row pitch
            // is a whole number of groups (stride 2: even | odd columns, half each)
            const int OFS = dxmin < 0 ? (-dxmin + 3) / 4 * 4 : 0;
            const int GPR = (OFS + (TW - 1) * sx + dxmax + 1 + 3) / 4, HCP = 4 * GPR, HCH = HCP / 2;
            if (4 * HR * HCP > BWORDS || HR * GPR > 128 * (cf ? 1 : NS4LIM)) continue;
            if (wtaps * 4 * MB + 4 * HR * HCP > 10200) continue;              // weights of a chunk + halo planes within the 160 KiB of LDS
            const int tx = (OW + TW - 1) / TW, ty = (OH + RT - 1) / RT;
            // cost: halo elements loaded + converted per launch (short rows coalesce badly: 16 elements of overhead per row),
            // plus the masked part of the tiles
            long long cost = (long long)tx * ty * (HR * HC + 16 * HR + NP / 2);
            if (g_s16g_lds_pref && (wtaps * 4 * MB + 4 * HR * HCP) * 16 > 80 * 1024 - 256) cost = cost * 3 / 2;      // one workgroup per CU only
            const long long wgs = (long long)tx * ty * q.N * pl.mblks * wcls;
            if (NT == 2 && wgs < 256) continue;              // few tiles: prefer the smaller tile
            if (cf && wgs < (g_s16g_mbl_wgs < 192 ? g_s16g_mbl_wgs : 192)) continue;      // (fused classes: a quarter of the workgroups — small problems keep one per class)
            if (best < 0 || cost < best) {
                best = cost;
                pl.NT = NT; pl.TW = TW; pl.RT = RT; pl.tiles_x = tx; pl.tiles_y = ty;
                pl.HR = HR; pl.HC = HC; pl.HCP = HCP; pl.HCH = HCH;
            }
        }
        if (best >= 0) break;
    }
    pl.CF = (cf && best >= 0) ? 1 : 0;
  }
    if (best < 0) return pl;
    if (g_s16g_cf == 2 && q.ncls == 4 && !pl.CF) return pl;      // (tests: a four-class problem that does not fuse is refused — the caller's route check fails)
    if ((long long)pl.tiles_x * pl.tiles_y * q.N * pl.mblks * q.ncls >= (1ll << 31)) return pl;
    if ((long long)q.Hs * q.Ws >= (1ll << 30)) return pl;
    pl.dymin = dymin;
    pl.dxmin = dxmin;
    // classes are packed with their own tap counts; reserve the largest for each
    pl.pack_words_per_class = (size_t)pl.nchunks * pl.mblks * maxtaps * 4 * MB;
    pl.ok = 1;
    return pl;
}

size_t nemar_s16g_pack_bytes(const S16gProblem& q, const S16gPlan& pl) {
    return pl.pack_words_per_class * 16 * (size_t)q.ncls + 4 * ABSMAX_WGS;
}

namespace {
// layout of the packed buffer: [class 0 words][class 1 words]...[max partial words, 4 ABSMAX_WGS bytes]
unsigned* pack_max_word(const S16gProblem& q, const S16gPlan& pl, void* packed) {
    return (unsigned*)((char*)packed + pl.pack_words_per_class * 16 * (size_t)q.ncls);
}
}  // namespace

void nemar_s16g_pack(const S16gProblem& q, const S16gPlan& pl, const float* w, long long wsm, long long wsc, void* packed,
                     hipStream_t st) {
    const int C = q.C0 + q.C1, MB = 32 * pl.MT;
    unsigned* const mw = pack_max_word(q, pl, packed);
    // the max runs over the whole weight tensor the rows / channels / taps are drawn from: extent = last addressed element + 1
    long long maxofs = 0;
    int maxtaps = 0;
    for (int c = 0; c < q.ncls; ++c) {
        maxtaps = q.ntaps[c] > maxtaps ? q.ntaps[c] : maxtaps;
        for (int t = 0; t < q.ntaps[c]; ++t)
            if (q.wofs[c][t] > maxofs) maxofs = q.wofs[c][t];
    }
    const long long n = (long long)(q.M - 1) * wsm + (long long)(C - 1) * wsc + maxofs + 1;
    hipLaunchKernelGGL(s16g_absmax_kernel, dim3(ABSMAX_WGS), dim3(256), 0, st, w, n, mw);
    const long long total = (long long)pl.nchunks * pl.mblks * maxtaps * 2 * MB;
    if (nemar_pack_recording()) {
        static_assert(ABSMAX_WGS == NEMAR_PACK_MAX_PARTS, "plan max jobs write the same partial words");
        NemarPackMaxArgs ma{w, n, mw, ABSMAX_WGS, 1};
        nemar_pack_record_job(PACK_FAM_MAX, &ma, ma.gx, 1);
        S16gPackArgs a;
        a.w = w; a.out = (u32x4*)packed; a.M = q.M; a.Cred = C; a.MB = MB; a.mblks = pl.mblks; a.nchunks = pl.nchunks;
        a.wsm = wsm; a.wsc = wsc; a.cls_words = (long long)pl.pack_words_per_class; a.maxbits = mw;
        a.gx = nemar_stream_grid(total, 256); a.gy = q.ncls;
        for (int c = 0; c < S16G_MAX_CLS; ++c) {
            a.ntaps[c] = c < q.ncls ? q.ntaps[c] : 0;
            for (int t = 0; t < S16G_MAX_TAPS; ++t) a.wofs[c][t] = (c < q.ncls && t < q.ntaps[c]) ? q.wofs[c][t] : 0;
        }
        nemar_pack_record_job(PACK_FAM_S16G, &a, a.gx, a.gy);
    }
    S16gProblem qq = q;                      // (pointers unused by the kernel)
    hipLaunchKernelGGL(s16g_pack_kernel, dim3(nemar_stream_grid(total, 256), q.ncls), dim3(256), 0, st, w, (u32x4*)packed, q.M, C, MB,
                       pl.mblks, pl.nchunks, wsm, wsc, (long long)pl.pack_words_per_class, (const unsigned*)mw, qq);
}

void nemar_s16g_conv(const S16gProblem& q, const S16gPlan& pl, const void* packed, hipStream_t st) {
    S16gParams p;
    p.src0 = q.src0; p.src1 = q.src1; p.C0 = q.C0; p.C1 = q.C1; p.Hs = q.Hs; p.Ws = q.Ws;
    p.wp = (const u32x4*)packed;
    p.wmax = pack_max_word(q, pl, const_cast<void*>(packed));
    p.bias = q.bias; p.dst0 = q.dst0; p.dst1 = q.dst1; p.M = q.M; p.M0 = q.M0; p.N = q.N;
    p.OHf = q.OHf; p.OWf = q.OWf; p.osy = q.osy; p.osx = q.osx;
    p.border = q.border; p.act = q.act; p.slope = q.slope; p.dbg = q.dbg; p.tl = q.tl;
    p.TW = pl.TW; p.RT = pl.RT; p.wshift = ilog2(pl.TW); p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.mblks = pl.mblks;
    p.nchunks = pl.nchunks;
    p.HR = pl.HR; p.HC = pl.HC; p.HCP = pl.HCP; p.HCH = pl.HCH; p.hp16 = pl.HR * pl.HCP; p.dymin = pl.dymin; p.dxmin = pl.dxmin;
    p.GPR = pl.HCP / 4;
    p.OFS = pl.dxmin < 0 ? (-pl.dxmin + 3) / 4 * 4 : 0;
    p.fd_gpr = make_fastdiv((unsigned)p.GPR);
    p.cls_words = (long long)pl.pack_words_per_class;
    double flop = 0.0;
    for (int c = 0; c < S16G_MAX_CLS; ++c) {
        const bool on = c < q.ncls;
        p.ntaps[c] = on ? q.ntaps[c] : 0;
        p.OH[c] = on ? q.OH[c] : 0;
        p.OW[c] = on ? q.OW[c] : 0;
        p.ooy[c] = on ? q.ooy[c] : 0;
        p.oox[c] = on ? q.oox[c] : 0;
        if (on) flop += 2.0 * q.N * q.OH[c] * q.OW[c] * (double)q.M * (q.C0 + q.C1) * q.ntaps[c];
    }
    for (int i = 0; i < S16G_MAX_TAPS; ++i) p.tapoff[i] = 0;
    int maxtaps = 0, alltaps = 0;
    for (int c = 0; c < q.ncls; ++c) {
        for (int t = 0; t < q.ntaps[c]; ++t) {
            const int ddy = q.dy[c][t] - pl.dymin, ddx = q.dx[c][t] + p.OFS;      // LDS column = image column - ox0 SX + OFS
            // (class-fused kernel: one flat table, the classes' taps back to back — the order of its weight region)
            p.tapoff[pl.CF ? alltaps + t : c * S16G_CLS_TAPS + t] = ddy * pl.HCP + (q.sstride == 2 ? (ddx & 1) * pl.HCH + (ddx >> 1) : ddx);
        }
        maxtaps = q.ntaps[c] > maxtaps ? q.ntaps[c] : maxtaps;
        alltaps += q.ntaps[c];
    }
    p.aw16 = (pl.CF ? alltaps : maxtaps) * 4 * 32 * pl.MT;
    // class-fused epilogue: 8-byte stores of the column-parity pairs (aligned destinations, equal widths of the two column classes)
    p.pair = 0;
    if (pl.CF) {
        const size_t planef = (size_t)q.OHf * q.OWf;
        p.pair = (q.OWf % 2 == 0 && planef % 2 == 0 && ((reinterpret_cast<uintptr_t>(q.dst0) | reinterpret_cast<uintptr_t>(q.dst1)) & 7) == 0) ? 1 : 0;
    }
    // channel blocks per workgroup: FOUR 64-channel blocks on 128-pixel tiles (MBL x MT x NT = 8 accumulator tiles), where two weight regions fit
    // next to the halo and the grid keeps >= 256 workgroups.  Measured stand-alone at batch 16 (profiles/r6_s16g_mbl_microbench.txt): the
    // translation net's 128 -> 256 stride-2 layer 199 -> 170 us.  TWO blocks per workgroup were measured too and lost (64 -> 128 stride 2:
    // 228 -> 243 us; stride-1 data gradients with 128 / 256 rows: 336 -> 437, 448 -> 530 us): half the conversions saved do not pay for
    // the second workgroup the CU loses to 384 registers per wave and the doubled weight region — those shapes keep one block per workgroup.
    int mbl = 1;
    if (!pl.CF) {
        const long long wgs = (long long)pl.tiles_x * pl.tiles_y * q.N * (pl.mblks / 4) * q.ncls;
        if (g_s16g_maxmbl >= 4 && pl.MT == 2 && pl.NT == 1 && pl.mblks % 4 == 0 && wgs >= g_s16g_mbl_wgs &&
            ((size_t)2 * p.aw16 + 4 * (size_t)p.hp16) * 16 <= (size_t)160 * 1024 - 2048)
            mbl = 4;
    }
    const dim3 g(pl.tiles_x * pl.tiles_y * q.N * (pl.mblks / mbl) * (pl.CF ? 1 : q.ncls)), b(256);
    p.ncls = q.ncls;
    p.xcd = g.x % 8 == 0 ? 1 : 0;
    const bool tm = g_timing && g_tev_used < MAX_TIMED;
    if (tm) {
        while (g_tev_made <= g_tev_used) {
            (void)hipEventCreate(&g_tev[g_tev_made][0]);
            (void)hipEventCreate(&g_tev[g_tev_made][1]);
            ++g_tev_made;
        }
        (void)hipEventRecord(g_tev[g_tev_used][0], st);
    }
    const size_t lds = ((size_t)(mbl > 1 ? 2 : 1) * p.aw16 + 4 * (size_t)p.hp16) * 16;
#ifdef NEMAR_HOST_EMULATION
#define S16G_GO2(MT_, NT_, SX_, NS_, MBL_) { hipLaunchKernelGGL((s16g_kernel<MT_, NT_, SX_, NS_, MBL_>), g, b, lds, st, p); }
#define S16G_GOCF(MT_) { hipLaunchKernelGGL((s16g_kernel<MT_, 1, 1, 1, 1, 1>), g, b, lds, st, p); }
#else
#define S16G_GOCF(MT_)                                                                                                  \
    {                                                                                                                   \
        const size_t lds_ = nemar_lds_bytes(reinterpret_cast<const void*>(&s16g_kernel<MT_, 1, 1, 1, 1, 1>), lds, (g_lds_claim & 4) != 0);  \
        hipLaunchKernelGGL((s16g_kernel<MT_, 1, 1, 1, 1, 1>), g, b, lds_, st, p);                                       \
    }
#endif
#ifdef NEMAR_HOST_EMULATION
#else
    // more than 64 KiB of dynamic LDS needs the attribute (nemar_lds_bytes sets it once per instantiation; whole-CU claim: common.h)
#define S16G_GO2(MT_, NT_, SX_, NS_, MBL_)                                                                              \
    {                                                                                                                   \
        const size_t lds_ = nemar_lds_bytes(reinterpret_cast<const void*>(&s16g_kernel<MT_, NT_, SX_, NS_, MBL_>), lds, (g_lds_claim & 4) != 0);       \
        hipLaunchKernelGGL((s16g_kernel<MT_, NT_, SX_, NS_, MBL_>), g, b, lds_, st, p);                                 \
    }
#endif
    // (MBL = 4 is instantiated for the 64-channel x 128-pixel tile only)
#define S16G_GO1(MT_, NT_, SX_, NS_)                                                                                    \
    {                                                                                                                   \
        if (MT_ == 2 && NT_ == 1 && mbl == 4) S16G_GO2(2, 1, SX_, NS_, 4)                                               \
        else S16G_GO2(MT_, NT_, SX_, NS_, 1)                                                                            \
    }
#define S16G_GO(MT_, NT_, SX_) { if (pl.HR * p.GPR <= 128) S16G_GO1(MT_, NT_, SX_, 1) else S16G_GO1(MT_, NT_, SX_, 2) }
#define S16G_BY_TILE(MT_)                                           \
    if (sx == 1 && pl.NT == 2) S16G_GO(MT_, 2, 1)                   \
    else if (sx == 1) S16G_GO(MT_, 1, 1)                            \
    else S16G_GO(MT_, 1, 2)
    const int sx = q.sstride;
#ifdef NEMAR_AB
    if (pl.CF) { if (pl.MT == 2) S16G_GOCF(2) else S16G_GOCF(1) } else
    if (pl.MT == 4) { S16G_BY_TILE(4) } else       // (128-channel tiles: nemar_tune(27, 4) only — the plan caps MT at g_s16g_maxmt)
#endif
    if (pl.MT == 2) { S16G_BY_TILE(2) }
    else { S16G_BY_TILE(1) }
#undef S16G_BY_TILE
#undef S16G_GO
#undef S16G_GO1
#undef S16G_GO2
#undef S16G_GOCF
    if (tm) {
        (void)hipEventRecord(g_tev[g_tev_used++][1], st);
        g_tev_flop += flop;
    }
}
