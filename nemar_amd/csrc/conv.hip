// K1 + K2 + K4 + K8: the convolution family on the gfx950 matrix cores, exact fp32.
//
// Replaces nn.Conv2d / nn.ConvTranspose2d (+ the ReflectionPad2d in front of them and the torch.cat that
// feeds them) at reference models/networks.py:349-377,418-439,576-597 and models/stn/layers.py:85,
// models/stn/unet_stn.py:80,97, models/stn/affine_stn.py:69-72,79 — forward, data gradient and weight
// gradient.  nn.Linear of the affine head is the 1x1 case on a 1x1 image.
//
// One tap-table implicit GEMM serves forward AND data-gradient (and therefore ConvTranspose2d, which is
// the data-gradient of a strided conv):
//     out[n, m, oy, ox] = act(bias[m] + sum_{t < ntaps} sum_{ch < Cs} A[(t*Cs+ch), m] *
//                              src[n, ch, oy*sy + dy[t], ox*sx + dx[t]])
//   - A is the weight tensor re-laid-out ("packed", see pack_weights_kernel) so that tile loads are dense 16 B/lane
//     runs and one ds_read_b128 feeds four MFMA steps;
//   - src is the logical channel concat of two tensors (never materialised);
//   - out-of-range taps read 0 (zero padding / strided data-gradient) or the mirrored texel (reflect);
//   - a stride-2 data-gradient is four launches, one per output-pixel parity class, each with the subset of
//     taps that lands on integer source positions;
//   - a stride-1 reflect data-gradient is the zero-padded data-gradient on the unpadded domain plus a small launch over
//     the border ring whose results are added to the texels the padding mirrored (nemar_conv2d_bwd_data).
// Kernels: igemm_ws2_kernel (wave-specialised, layers big enough for 128x128 tiles), igemm_kernel (generic, every wave
// stages and multiplies; small / odd layers and the ring), igemm_ws_kernel (first-generation wave-specialised, kept for
// A/B measurements).
// GEMM view: M = output channels, N = output pixels, K = taps x source channels.  MFMA tile
// v_mfma_f32_32x32x2_f32 with A = weights (rows = channels) and B = gathered pixels (cols = pixels), so the
// accumulator's col = lane&31 runs along pixels and NCHW stores are coalesced.  f32-in MFMA is an fmaf chain
// (bit-exact fp32, 157 TF peak): no reduced precision anywhere.
//
// The weight gradient is a second implicit GEMM, dW[k][c,r,s] = sum_pixels gy[k,p] * src[c, p (+) tap], with
// the (huge) pixel reduction split across workgroups and fp32 atomics into the caller's gradient buffer:
// conv_wgrad.hip (wave-specialised) for everything whose gy planes are 16-byte chunkable, wgrad_kernel below for the rest.
#include "common.h"
#include "conv_split16.h"
#include "conv_s16g.h"
#include "conv_k7.h"
#include "pack_plan.h"

extern int g_split16_ring3;                      // conv_split16.hip
void nemar_norm_planes_debug(int bits);          // norm_planes.hip: ablation bits of the fused producer (measurement only)

// conv_narrow.hip: VALU + LDS-halo kernels for layers with <= 4 output channels
bool nemar_narrow_eligible(int K, int C1, int R, int S, int stride, int N, int OH, int OW);
int nemar_narrow_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int K, int R,
                     int pad, int border, int act, float slope, float* part, size_t part_floats, hipStream_t st);
int nemar_narrow_wgrad_splits(int N, int C, int OH, int OW);
int nemar_narrow_wgrad(const float* x, const float* gy, float* gw, int N, int C, int H, int W, int K, int R, int pad,
                       int border, float* part, hipStream_t st);

// conv_wgrad.hip: wave-specialised weight gradient for wide layers
bool nemar_wgrad2_eligible(int K, int OH, int OW, const float* gy);
void nemar_wgrad2_plan(int K, int J, int P, int target_blocks, int* splits_out, int* pix_per_split_out);
void nemar_wgrad2_launch(const float* x0, int C0, const float* x1, int C1, const float* gy, float* gw, float* gb, int N,
                         int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad, int pad_mode,
                         int target_blocks, bool vec_ok, int dbg, float* part, hipStream_t st);
// reduce.hip: dst (+)= sum of `splits` slabs in split order (the deterministic second stage of every split reduction)
void nemar_sum_partials(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate,
                        hipStream_t st);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;        // reduction depth per LDS stage (fwd/dgrad)
constexpr int MAX_TAPS = 64;  // 7x7 = 49
constexpr int BORDER_ZERO = 0, BORDER_REFLECT = 1;
constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3;

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ int reflect(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

struct TapTable {
    int n;
    int dyx[MAX_TAPS];   // (dy << 16) | (dx & 0xffff): one dword per tap so that a wave-uniform lookup is a scalar load
    short dy[MAX_TAPS], dx[MAX_TAPS];
    int wofs[MAX_TAPS];  // offset of the tap inside one [R][S] filter (pack kernel only)
};

// ---- weight packing -----------------------------------------------------------------------------------------------
// W[m*wsm + ch*wsc + wofs[t]] with reduction index k = t*Cs + ch goes to
//     A[((k/8)*2 + (k&1)) * Mpad*4 + m*4 + ((k%8)>>1)]            (zero padded to KredPad x Mpad)
// i.e. blocks of 8 reduction rows, split by the MFMA k-slot (k&1), channel-major, with the four MFMA steps of the block
// adjacent: lane (m, kslot) of v_mfma_f32_32x32x2_f32 fetches its A operand for 4 consecutive steps with ONE
// ds_read_b128, the rows of a 16-lane read group fall on 16 different bank slots, and a tile stage is still a dense
// run of 16-byte chunks for global_load_lds.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int M,
                                                           int Mpad, int Cs, int Kred, int KredPad, int wsm, int wsc,
                                                           int zero_tail, TapTable taps) {
    __shared__ int s_wofs[MAX_TAPS];
    for (int i = threadIdx.x; i < MAX_TAPS; i += blockDim.x) s_wofs[i] = i < taps.n ? taps.wofs[i] : 0;
    __syncthreads();
    const int core = KredPad * Mpad, total = core + zero_tail;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int blk = idx / (Mpad * 4), within = idx - blk * (Mpad * 4);
        const int m = within >> 2, s = within & 3;
        const int kk = 8 * (blk >> 1) + 2 * s + (blk & 1);
        float v = 0.f;
        if (idx < core && m < M && kk < Kred) {
            const int t = kk / Cs, ch = kk - t * Cs;
            v = w[(size_t)m * wsm + (size_t)ch * wsc + s_wofs[t]];
        }
        wp[idx] = v;
    }
}

struct IgemmParams {
    const float* src0; const float* src1; int C0, C1, Hs, Ws;
    const float* wp; int Mpad, M, Kred;
    const float* zero;   // >= 16 readable bytes of zeros (tail of the packed-weight buffer)
    long long* tl;       // optional timeline buffer (nemar_tune_ptr): per-stage s_memtime stamps of workgroup 0
    int dbg;             // ablation switches (nemar_tune key 2): 1 = skip staging, 2 = skip MFMAs, 4 = skip barriers
    const float* bias;
    float* dst0; float* dst1; int M0;
    int OH, OW, OHf, OWf, osy, ooy, osx, oox;
    int N, P;
    int sy, sx, border, act, pad;
    float slope;
    // ring mode (ring_p > 0): the "pixels" of this launch are the border ring of width ring_p around a ring_H x ring_W
    // image, in padded coordinates; OH*OW = ring length; results are atomically ADDED at the reflected in-image position
    int ring_p, ring_H, ring_W, ksplit;
    // ksplit > 1 (reduction split over grid.z: few, deep tiles): split z stores its partial tile to slab z of `part`
    // (same indexing as dst0) and nemar_sum_partials adds the slabs in order — no atomics, bitwise reproducible
    float* part; long long part_stride;
    // reflect data gradient without a ring launch (wave-specialised 16-byte-load kernel, 3x3 pad 1): pre-folded border rows
    // [2][3][N][K][Ws] and border column groups [2][N][K][Hs][4] of the source (reflect_aux_kernel)
    int rf; const float* rf_row; const float* rf_col;
    int xcd;             // XCD-aware workgroup -> tile mapping of the wave-specialised kernel (grid.x % 8 == 0)
    FastDiv fd_ohw, fd_ow, fd_cs;
    TapTable taps;
};

// pixel index inside one image -> (oy, ox).  Ring mode enumerates, in padded coordinates of a (H+2p) x (W+2p) plane:
// top band (p rows), bottom band (p rows), then for each image row the p left and p right columns.
__device__ __forceinline__ void decode_ring(unsigned rp, unsigned H, unsigned W, unsigned rem, unsigned& oy, unsigned& ox) {
    const unsigned Wp = W + 2 * rp, band = rp * Wp;
    if (rem < 2 * band) {
        const unsigned q = rem < band ? rem : rem - band;
        const unsigned r = q / Wp;
        oy = rem < band ? r : H + rp + r;
        ox = q - r * Wp;
    } else {
        const unsigned q = rem - 2 * band;
        const unsigned r = q / (2 * rp), e = q - r * (2 * rp);
        oy = rp + r;
        ox = e < rp ? e : W + e;
    }
}
__device__ __forceinline__ void decode_pixel(const IgemmParams& p, unsigned rem, unsigned& oy, unsigned& ox) {
    if (p.ring_p == 0) {
        oy = fd_div(rem, p.fd_ow);
        ox = rem - oy * (unsigned)p.OW;
        return;
    }
    decode_ring((unsigned)p.ring_p, (unsigned)p.ring_H, (unsigned)p.ring_W, rem, oy, ox);
}

// WM x WN waves, each TM x TN MFMA tiles of 32x32 (workgroup = WM*WN*64 threads).  FAST: Cs % BK == 0 && C0 % BK == 0,
// so a whole BK-deep stage shares one tap and one source tensor (address math once per stage instead of per element).
//
// Staging is direct global -> LDS (global_load_lds_*): no VGPR round trip, no ds_write pass, and no register hazards for
// the compiler to guard with early waits.  The stage for step k+1 is issued before the MFMAs of step k and drained
// (s_waitcnt vmcnt(0) + barrier) after them.  A tile: packed weights, 16 B per lane, dense LDS rows (one wave
// instruction = 1 KiB).  B tile: one 4-byte gather per lane, 64 consecutive pixels of one reduction row per wave
// instruction; taps that fall outside a zero-padded source (and tile tails) are pointed at a zero page.
template <int WM, int WN, int TM, int TN, bool FAST, int NBUF = 2>
__global__ __launch_bounds__(WM * WN * 64) void igemm_kernel(IgemmParams p) {
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDB = BN + 4;
    constexpr int SEGS = BN / 64;               // 64-pixel segments per B row
    constexpr int RGRP = NW / SEGS;             // waves working on the same segment (each takes every RGRP-th row)
    constexpr int BROWS = BK / RGRP;            // B rows per wave per stage
    constexpr int A_INSTR = BK * BM / 256;      // 1 KiB wave-instructions per A stage
    constexpr int A_PER = (A_INSTR + NW - 1) / NW;
    static_assert(BN % 64 == 0 && NW % SEGS == 0 && BK % RGRP == 0, "tile");

    // ONE __shared__ object: with several, hipcc guards every ds_read of a stage with s_waitcnt vmcnt(0) while a
    // global_load_lds is in flight (it cannot tell which object the DMA writes), which would serialise the pipeline
    constexpr int A_FLOATS = BK * BM, B_FLOATS = BK * LDB;
    constexpr int LOADS = A_PER + BROWS;          // global->LDS instructions per wave per stage (uniform when NBUF > 2)
    static_assert(NBUF == 2 || (FAST && A_INSTR % NW == 0), "counted waits need the same number of loads in every wave");
    static_assert(NBUF >= 2 && NBUF <= 4 && 2 * LOADS < 64, "ring depth");
    __shared__ __attribute__((aligned(16))) float smem[NBUF * A_FLOATS + NBUF * B_FLOATS + MAX_TAPS];
    float* const As0 = smem;
    float* const Bs0 = smem + NBUF * A_FLOATS;
    int* const s_tap = reinterpret_cast<int*>(smem + NBUF * A_FLOATS + NBUF * B_FLOATS);

    const int tid = threadIdx.x;
    const int wid = tid >> 6, lane = tid & 63;
    for (int i = tid; i < MAX_TAPS; i += NT)
        s_tap[i] = i < p.taps.n ? (((int)p.taps.dy[i] << 16) | ((int)p.taps.dx[i] & 0xffff)) : 0;

    const int m0 = blockIdx.y * BM;
    const int p0 = blockIdx.x * BN;
    const int Cs = p.C0 + p.C1;
    const int HW = p.Hs * p.Ws;

    // ---- this lane's pixel column of the B tile (fixed for the whole reduction) ----------------------------------
    const int seg = wid % SEGS, rgrp = wid / SEGS;
    const int pc = seg * 64 + lane;
    const int pix = p0 + pc;
    const bool pvalid = pix < p.P;
    int by = 0, bx = 0;
    const float* s0n = p.src0;
    const float* s1n = p.src1;
    {
        const unsigned upix = pvalid ? (unsigned)pix : 0u;
        const unsigned n = fd_div(upix, p.fd_ohw);
        const unsigned rem = upix - n * (unsigned)(p.OH * p.OW);
        unsigned oy, ox;
        decode_pixel(p, rem, oy, ox);
        by = (int)oy * p.sy;
        bx = (int)ox * p.sx;
        s0n = p.src0 + (size_t)n * p.C0 * HW;
        if (p.C1) s1n = p.src1 + (size_t)n * p.C1 * HW;
    }
    // ---- this lane's 16-byte slots of the A tile -----------------------------------------------------------------------
    const float* wsrc[A_PER];
    int a_lds[A_PER];
#pragma unroll
    for (int q = 0; q < A_PER; ++q) {
        const int e = (wid + q * NW) * 256 + lane * 4;      // float index inside the dense [4 blocks][BM][4] stage
        const int blk = e / (BM * 4), m = (e - blk * (BM * 4)) >> 2;
        wsrc[q] = p.wp + ((size_t)blk * p.Mpad + m0 + m) * 4;
        a_lds[q] = (wid + q * NW) * 256;                    // wave-uniform LDS base (floats)
    }
    __syncthreads();  // s_tap visible

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int f_tap = -1, f_off = 0;      // FAST staging: tap whose border arithmetic is cached, this lane's y*Ws+x for it
    bool f_inb = false;
#define IGEMM_ISSUE_STAGE(k0_, buf_)                                                                                 \
    {                                                                                                                \
        const int k0 = (k0_);                                                                                        \
        _Pragma("unroll") for (int q = 0; q < A_PER; ++q) {                                                          \
            if (A_INSTR % NW == 0 || wid + q * NW < A_INSTR)                                                         \
                glds_b128(wsrc[q] + (size_t)k0 * p.Mpad, As0 + (buf_) * A_FLOATS + a_lds[q]);                                       \
        }                                                                                                            \
        if (FAST) {                                                                                                  \
            const int t = (int)fd_div((unsigned)k0, p.fd_cs);                                                        \
            const int ch0 = k0 - t * Cs;                                                                             \
            if (t != f_tap) {          /* border arithmetic + tap-table read once per tap, not per stage */          \
                f_tap = t;                                                                                           \
                const int tp = s_tap[t];                                                                             \
                int y = by + (tp >> 16), x = bx + (int)(short)(tp & 0xffff);                                         \
                f_inb = pvalid;                                                                                      \
                if (p.border == BORDER_REFLECT) {                                                                    \
                    y = reflect(y, p.Hs);                                                                            \
                    x = reflect(x, p.Ws);                                                                            \
                } else {                                                                                             \
                    f_inb = f_inb && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;                   \
                }                                                                                                    \
                f_off = f_inb ? y * p.Ws + x : 0;                                                                    \
            }                                                                                                        \
            const bool inb = f_inb && k0 < p.Kred;                                                                   \
            const float* base = (ch0 < p.C0) ? s0n + (size_t)ch0 * HW : s1n + (size_t)(ch0 - p.C0) * HW;           \
            base += (size_t)rgrp * HW + f_off;                                                                       \
            _Pragma("unroll") for (int i = 0; i < BROWS; ++i)                                                        \
                glds_b32(inb ? base + (size_t)(RGRP * i) * HW : p.zero,                                              \
                         Bs0 + (buf_) * B_FLOATS + (rgrp + RGRP * i) * LDB + seg * 64);       \
        } else {                                                                                                     \
            _Pragma("unroll") for (int i = 0; i < BROWS; ++i) {                                                      \
                const int kk = k0 + rgrp + RGRP * i;                                                                 \
                const float* src = p.zero;                                                                           \
                if (pvalid && kk < p.Kred) {                                                                         \
                    const unsigned t = fd_div((unsigned)kk, p.fd_cs);                                                \
                    const int ch = kk - (int)t * Cs;                                                                 \
                    const int tp = s_tap[t];                                                                         \
                    int y = by + (tp >> 16), x = bx + (int)(short)(tp & 0xffff);                                     \
                    bool inb = true;                                                                                 \
                    if (p.border == BORDER_REFLECT) {                                                                \
                        y = reflect(y, p.Hs);                                                                        \
                        x = reflect(x, p.Ws);                                                                        \
                    } else {                                                                                         \
                        inb = (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;                          \
                    }                                                                                                \
                    if (inb) src = ((ch < p.C0) ? s0n + (size_t)ch * HW : s1n + (size_t)(ch - p.C0) * HW) + y * p.Ws + x; \
                }                                                                                                    \
                glds_b32(src, Bs0 + (buf_) * B_FLOATS + (rgrp + RGRP * i) * LDB + seg * 64);                                                 \
            }                                                                                                        \
        }                                                                                                            \
    }

    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ring mode splits the reduction over blockIdx.z (the atomic epilogue sums the partial results)
    const int nk_all = (p.Kred + BK - 1) / BK;
    const int nk_per = (nk_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int ks0 = blockIdx.z * nk_per, nk = min(nk_all, ks0 + nk_per);
    if (ks0 >= nk) return;
    // Ring mode: most (tile, tap) pairs have no source texel in range (a top-band ring pixel is reached by the r = 0 taps
    // only), i.e. an all-zero B tile.  The workgroup collects the set of taps that reach ANY of its pixels and skips the
    // stages of the others (~2/3 of them for a 3x3 layer).
    unsigned long long amask = ~0ull;
    if (p.ring_p) {
        unsigned lo = 0, hi = 0;
        for (int t = 0; t < p.taps.n; ++t) {
            const int tp = s_tap[t];
            const int y = by + (tp >> 16), x = bx + (int)(short)(tp & 0xffff);
            const bool hit = pvalid && (p.border == BORDER_REFLECT ||
                                        ((unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws));
            if (__ballot(hit) != 0ull) { if (t < 32) lo |= 1u << t; else hi |= 1u << (t - 32); }
        }
        unsigned* const s_mask = reinterpret_cast<unsigned*>(s_tap + MAX_TAPS - 2);   // taps.n <= 49: last slots are free
        if (tid == 0) { s_mask[0] = 0u; s_mask[1] = 0u; }
        __syncthreads();
        if (lane == 0) { atomicOr(&s_mask[0], lo); atomicOr(&s_mask[1], hi); }
        __syncthreads();
        amask = ((unsigned long long)s_mask[1] << 32) | s_mask[0];
    }
#define IGEMM_STAGE_ACTIVE(ks_)                                                                                   \
    (amask == ~0ull ||                                                                                                \
     ((amask >> fd_div((unsigned)((ks_) * BK), p.fd_cs)) &                                                            \
      ((2ull << (fd_div((unsigned)min((ks_) * BK + BK - 1, p.Kred - 1), p.fd_cs) - fd_div((unsigned)((ks_) * BK), p.fd_cs))) - 1ull)) != 0ull)
#define IGEMM_NEXT_ACTIVE(var_)  while ((var_) < nk && !IGEMM_STAGE_ACTIVE(var_)) ++(var_)
    // NBUF-deep LDS ring: up to NBUF-2 further stages stay in flight while one is consumed (counted s_waitcnt vmcnt);
    // NBUF = 2 is plain double buffering.  One barrier per stage: it publishes the stage and retires the buffer consumed
    // before it.  `slot` counts consumed stages, `n_iss` issued ones (inactive stages take no slot).
    int iss = ks0;
    IGEMM_NEXT_ACTIVE(iss);
    int first = iss, n_iss = 0;
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d)
        if (iss < nk) {
            IGEMM_ISSUE_STAGE(iss * BK, n_iss % NBUF);
            ++n_iss;
            ++iss;
            IGEMM_NEXT_ACTIVE(iss);
        }
    int slot = 0;
    for (int ks = first; ks < nk; ++slot) {
        const int buf = slot % NBUF;
        if (NBUF == 2) {
            wait_vmem();
        } else {
            const int ahead = n_iss - slot - 1;       // stages issued beyond this one
            if (ahead >= 2 && NBUF >= 4) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * LOADS) & 15) | (((2 * LOADS) >> 4) << 14));
            else if (ahead >= 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (LOADS & 15) | ((LOADS >> 4) << 14));
            else wait_vmem();
        }
        if (!(p.dbg & 4)) __syncthreads();
        if (iss < nk && !(p.dbg & 1)) {
            IGEMM_ISSUE_STAGE(iss * BK, n_iss % NBUF);
            ++n_iss;
            ++iss;
            IGEMM_NEXT_ACTIVE(iss);
        }
        ++ks;
        IGEMM_NEXT_ACTIVE(ks);
        if (!(p.dbg & 2)) {
            // all MFMA operands of the stage into registers first (one LDS round trip per stage), then the MFMAs
            // back to back: reading fragments just-in-time makes hipcc reuse the operand registers, and the
            // write-after-read wait on them puts an LDS latency between every pair of MFMAs
            f32x4 a[BK / 8][TM];
            float b[BK / 2][TN];
#pragma unroll
            for (int kg = 0; kg < BK / 8; ++kg)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[kg][i] = *reinterpret_cast<const f32x4*>(
                        As0 + buf * A_FLOATS + ((kg * 2 + lhi) * BM + (wm * TM + i) * 32 + l31) * 4);
#pragma unroll
            for (int k2 = 0; k2 < BK / 2; ++k2) {
                const int kr = 2 * k2 + lhi;
#pragma unroll
                for (int j = 0; j < TN; ++j) b[k2][j] = Bs0[buf * B_FLOATS + kr * LDB + (wn * TN + j) * 32 + l31];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the fragment loads above, in their own registers
#pragma unroll
            for (int k2 = 0; k2 < BK / 2; ++k2)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2 >> 2][i][k2 & 3], b[k2][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of `buf` are done before the next barrier
        }
    }
#undef IGEMM_ISSUE_STAGE
#undef IGEMM_STAGE_ACTIVE
#undef IGEMM_NEXT_ACTIVE

    // ---- epilogue: bias + activation, NCHW store (lane&31 runs along pixels => coalesced rows) -----------------
    const size_t oplane = (size_t)p.OHf * p.OWf;
    const int M1 = p.M - p.M0;
    // split reduction (tiny, deep problems: a 2x2-pixel 128->128 layer is 72 serial stages in one or two workgroups): this
    // split's partial tile goes to its own slab (no bias / activation in this mode: the launcher guarantees it)
    float* const d0 = gridDim.z > 1 ? p.part + (size_t)blockIdx.z * (size_t)p.part_stride : p.dst0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int opix = p0 + (wn * TN + j) * 32 + l31;
        if (opix >= p.P) continue;
        const unsigned n = fd_div((unsigned)opix, p.fd_ohw);
        const unsigned rem = (unsigned)opix - n * (unsigned)(p.OH * p.OW);
        unsigned oy, ox;
        decode_pixel(p, rem, oy, ox);
        if (p.ring_p) {
            // ring results go to a compact [n][m][ring] scratch (lanes = consecutive ring positions: coalesced stores);
            // ring_gather_kernel adds them to the texels the reflect padding mirrored
            const size_t rlen = (size_t)p.OH * p.OW;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (m < p.M) d0[((size_t)n * p.M + m) * rlen + rem] = acc[i][j][r];
                }
            continue;
        }
        const size_t sp = (size_t)((int)oy * p.osy + p.ooy) * p.OWf + ((int)ox * p.osx + p.oox);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < p.M) {
                    float v = acc[i][j][r];
                    if (p.bias && blockIdx.z == 0) v += p.bias[m];       // split reductions: slab 0 carries the bias
                    v = apply_act(v, p.act, p.slope);
                    if (m < p.M0) {
                        if (d0) d0[((size_t)n * p.M0 + m) * oplane + sp] = v;
                    } else {
                        p.dst1[((size_t)n * M1 + (m - p.M0)) * oplane + sp] = v;
                    }
                }
            }
        }
    }
}

// ---- wave-specialised 128x128 tile (FAST shapes): 8 MFMA waves + 2 loader waves ---------------------------------------
// Measured on the 4-/8-wave kernels above (tools/timeline_conv.py, ablation switches): the MFMA phase of a stage and the
// global->LDS staging of the next one do not overlap — every wave first spends ~1-3k cycles getting its
// global_load_lds accepted by the memory pipeline and only then starts its MFMAs, and all waves of a workgroup do so
// together after each barrier.  Here the MFMA waves issue no vector-memory instructions at all: two extra waves own
// the staging (each: one 64-pixel segment of every B row + half of the A rows), run TWO stages ahead through a 3-deep
// LDS ring, and keep one stage in flight across the barrier with a counted s_waitcnt vmcnt(N).
constexpr int WS_NC = 8, WS_NL = 2, WS_NT = (WS_NC + WS_NL) * 64, WS_NBUF = 3;
__global__ __launch_bounds__(WS_NT) void igemm_ws_kernel(IgemmParams p) {
    constexpr int BM = 128, BN = 128, LDB = BN + 4, TM = 2, WN = 4;
    constexpr int A_FLOATS = BK * BM, B_FLOATS = BK * LDB;
    constexpr int A_PER_LOADER = BK * BM / 256 / WS_NL;      // 1 KiB wave-instructions of A per loader per stage (4)
    constexpr int LOADS_PER_STAGE = A_PER_LOADER + BK;       // + one 64-pixel segment of each of the BK rows (16)
    __shared__ __attribute__((aligned(16))) float smem[WS_NBUF * (A_FLOATS + B_FLOATS)];
    float* const As0 = smem;
    float* const Bs0 = smem + WS_NBUF * A_FLOATS;

    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * BM, p0 = blockIdx.x * BN;
    const int Cs = p.C0 + p.C1, HW = p.Hs * p.Ws;
    const int nk = (p.Kred + BK - 1) / BK;

    if (wid >= WS_NC) {
        // ================================ loader waves ================================
        const int seg = wid - WS_NC;                       // 64-pixel segment of the B tile owned by this wave
        const int pix = p0 + seg * 64 + lane;
        const bool pvalid = pix < p.P;
        const unsigned upix = pvalid ? (unsigned)pix : 0u;
        const unsigned n = fd_div(upix, p.fd_ohw);
        const unsigned rem = upix - n * (unsigned)(p.OH * p.OW);
        const unsigned oy = fd_div(rem, p.fd_ow);
        const int by = (int)oy * p.sy, bx = (int)(rem - oy * (unsigned)p.OW) * p.sx;
        const float* s0n = p.src0 + (size_t)n * p.C0 * HW;
        const float* s1n = p.C1 ? p.src1 + (size_t)n * p.C1 * HW : p.src0;
        const float* wsrc[A_PER_LOADER];
        int a_lds[A_PER_LOADER];
#pragma unroll
        for (int q = 0; q < A_PER_LOADER; ++q) {
            const int inst = seg * A_PER_LOADER + q;
            const int e = inst * 256 + lane * 4;
            const int blk = e / (BM * 4), m = (e - blk * (BM * 4)) >> 2;
            wsrc[q] = p.wp + ((size_t)blk * p.Mpad + m0 + m) * 4;
            a_lds[q] = inst * 256;
        }
        auto issue = [&](int ks) {
            const int k0 = ks * BK, buf = ks % WS_NBUF;
#pragma unroll
            for (int q = 0; q < A_PER_LOADER; ++q) glds_b128(wsrc[q] + (size_t)k0 * p.Mpad, As0 + buf * A_FLOATS + a_lds[q]);
            const unsigned t = fd_div((unsigned)k0, p.fd_cs);
            const int ch0 = k0 - (int)t * Cs;
            int y = by + p.taps.dy[t], x = bx + p.taps.dx[t];
            bool inb = pvalid;
            if (p.border == BORDER_REFLECT) {
                y = reflect(y, p.Hs);
                x = reflect(x, p.Ws);
            } else {
                inb = inb && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
            }
            const float* base = (ch0 < p.C0) ? s0n + (size_t)ch0 * HW : s1n + (size_t)(ch0 - p.C0) * HW;
            base += inb ? y * p.Ws + x : 0;
#pragma unroll
            for (int r = 0; r < BK; ++r)
                glds_b32(inb ? base + (size_t)r * HW : p.zero, Bs0 + buf * B_FLOATS + r * LDB + seg * 64);
        };
        issue(0);
        if (nk > 1) issue(1);
        // stage 0 must have landed before the first barrier; stage 1 may stay in flight
        if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (LOADS_PER_STAGE & 15) | ((LOADS_PER_STAGE >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        for (int ks = 0; ks < nk; ++ks) {
            // buffer (ks+2)%3 == (ks-1)%3 was released by the barrier that ended stage ks-1
            if (ks + 2 < nk) {
                issue(ks + 2);
                __builtin_amdgcn_s_waitcnt(0x0F70 | (LOADS_PER_STAGE & 15) | ((LOADS_PER_STAGE >> 4) << 14));  // ks+1 landed
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ================================ MFMA waves ================================
    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __builtin_amdgcn_s_barrier();      // stage 0 is in LDS
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks % WS_NBUF;
        f32x4 a[BK / 8][TM];
        float b[BK / 2];
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[kg][i] = *reinterpret_cast<const f32x4*>(
                    As0 + buf * A_FLOATS + ((kg * 2 + lhi) * BM + (wm * TM + i) * 32 + l31) * 4);
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2) b[k2] = Bs0[buf * B_FLOATS + (2 * k2 + lhi) * LDB + wn * 32 + l31];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2 >> 2][i][k2 & 3], b[k2], acc[i], 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): all LDS reads of this buffer returned before releasing it
        __builtin_amdgcn_s_barrier();
    }

    const size_t oplane = (size_t)p.OHf * p.OWf;
    const int M1 = p.M - p.M0;
    const int opix = p0 + wn * 32 + l31;
    if (opix >= p.P) return;
    const unsigned n = fd_div((unsigned)opix, p.fd_ohw);
    const unsigned rem = (unsigned)opix - n * (unsigned)(p.OH * p.OW);
    const unsigned oy = fd_div(rem, p.fd_ow);
    const unsigned ox = rem - oy * (unsigned)p.OW;
    const size_t sp = (size_t)((int)oy * p.osy + p.ooy) * p.OWf + ((int)ox * p.osx + p.oox);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < p.M) {
                float v = acc[i][r];
                if (p.bias) v += p.bias[m];
                v = apply_act(v, p.act, p.slope);
                if (m < p.M0) {
                    if (p.dst0) p.dst0[((size_t)n * p.M0 + m) * oplane + sp] = v;
                } else {
                    p.dst1[((size_t)n * M1 + (m - p.M0)) * oplane + sp] = v;
                }
            }
        }
    }
}

// ---- wave-specialised 128x128 tile, second generation: 4 MFMA waves of 32 channels x 128 pixels + 2 loader waves -------
// Same staging as igemm_ws_kernel (the loader code is shared in spirit: A dense 16 B/lane, B one gathered texel per lane),
// different consumer:
//   * B fragments are read with ds_read_b128: lane (J, kslot) takes pixels 4J..4J+3 of reduction row 2s+kslot and feeds
//     component t to MFMA tile t, i.e. the four 32-pixel MFMA tiles of a wave interleave the 128 pixels (tile t owns
//     pixels == t mod 4).  One 16-byte read per MFMA step serves four MFMAs; a 16-lane read group covers one 256-byte run
//     of a row, so it is conflict-free for any row pitch; and in the epilogue every lane owns 4 consecutive pixels of
//     each of its channel rows: 16-byte stores.
//   * A fragments: one ds_read_b128 per 4 MFMA steps (k-interleaved packing, see pack_weights_kernel).
//   * fragments of the next 8 reduction rows are prefetched into a second register set before the 16 MFMAs of the
//     current 8 rows are issued; the stage barrier sits between the two halves of a stage, when the reads of the current
//     buffer have all been issued, so neither LDS latency nor the barrier idles the matrix pipe.
// MT = MFMA waves = 32-channel row tiles (BM = 32*MT: 128 / 64 / 32 output channels per workgroup).
// VEC = the B tile is staged with 16-byte global->LDS loads: lane = 4 consecutive output pixels of one reduction row
// (stride-1 layers with OW % 4 == 0 and |dx| <= 1: the four source texels are consecutive in memory, at an address that
// is only 4-byte aligned — global_load_lds_dwordx4 takes that).  A group whose shifted window would stick out of the
// source row by one texel is loaded from the clamped address instead, and the MFMA wave that consumes it rotates the
// three good texels into place and inserts the mirrored texel (reflect) or 0 (zero padding) — a handful of
// v_cndmask per stage on the two border lanes of a row.  8 wave-instructions per stage instead of 32.
// SPB = stages per barrier.  1: 3-deep stage ring, one workgroup barrier per 16 reduction rows.  2: 4 stage slots used as
// two 32-row halves — the loaders fill one half while the MFMA waves consume the other, one barrier per 32 rows.
// ADIR = the A operand (packed weights) bypasses LDS.  A wave's A fragments are private to it (wave w owns channel rows
// 32w..32w+31 of the tile; only the pixel tile B is shared by the four MFMA waves), and the packed layout IS the fragment
// layout — lane (m, kslot) needs one 16-byte word per 8 reduction rows — so each MFMA wave fetches its own two words per stage
// straight from global memory (L2-resident: the whole packed tensor is 2.4 MB) into registers, one stage ahead, in the
// issue shadow of its MFMAs.  The loader waves then move HALF the bytes per stage (the B tile only): they are the critical path
// of this kernel, and global->LDS DMA throughput per CU is what bounds them.
// RING = LDS ring depth of the one-stage-per-barrier variant: the loaders run RING - 1 stages ahead of the MFMA waves (3: two
// stage times = ~1.7 us for a global->LDS copy to land; 4 / 5 trade LDS (16 KiB per stage) for more latency tolerance).
template <int MT, bool VEC, int SPB = 1, int NL = 2, bool ADIR = false, int RING = 3>   // NL = loader waves (4 only with VEC)
__global__ __launch_bounds__((MT + NL) * 64) void igemm_ws2_kernel(IgemmParams p) {
    static_assert(NL == 2 || NL == 4, "loader split");
    static_assert(!ADIR || SPB == 1, "direct A operands: one-stage barrier variant only");
    static_assert(RING >= 3 && RING <= 5, "ring depth");
    constexpr int W2_NBUF = SPB == 1 ? RING : 4;
    constexpr int BM = 32 * MT, BN = 128, LDB = VEC ? BN : BN + 4;
    constexpr int A_FLOATS = ADIR ? 0 : BK * BM, B_FLOATS = BK * LDB;
    constexpr int A_PER_LOADER = ADIR ? 0 : BK * BM / 256 / NL;   // 1 KiB wave-instructions of A per loader per stage
    constexpr int B_PER_LOADER = VEC ? BK / 2 / NL : BK / (NL / 2);   // VEC: 2 rows x 128 px per instruction; else 1 row x 64 px
    constexpr int LOADS_PER_STAGE = A_PER_LOADER + B_PER_LOADER;
    static_assert(BK == 16, "a stage is two 8-row fragment groups");
    __shared__ __attribute__((aligned(16))) float smem[W2_NBUF * (A_FLOATS + B_FLOATS)];
    float* const As0 = smem;
    float* const Bs0 = smem + W2_NBUF * A_FLOATS;

    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    // Workgroup -> tile.  Consecutive workgroup ids land on consecutive XCDs (id % 8), each with its own L2; neighbouring
    // pixel tiles share half of their source rows (a 3x3 tile of two image rows reads four), so with the identity mapping
    // every XCD fetches the halo rows its neighbours already hold.  p.xcd != 0: XCD k takes a CONTIGUOUS run of pixel tiles
    // (and all channel tiles of each), so halo rows and the second channel tile's B tile are L2 hits.
    int bx = blockIdx.x, by_ = blockIdx.y;
    if (p.xcd) {
        const int gxx = gridDim.x, gyy = gridDim.y;
        const int b = bx + gxx * by_;
        const int k = b & 7, j = b >> 3;
        bx = k * (gxx >> 3) + j / gyy;
        by_ = j - (j / gyy) * gyy;
    }
    const int m0 = by_ * BM, p0 = bx * BN;
    const int Cs = p.C0 + p.C1, HW = p.Hs * p.Ws;
    // grid.z > 1: the reduction is split over workgroups (few, deep tiles — D's 256->512 k4 data gradient is 128 tiles x 512
    // stages); partial results meet in the zero-filled destination through atomics (no bias / activation in that mode)
    const int nk_all = (p.Kred + BK - 1) / BK;
    const int nk_per = (nk_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int ks0 = blockIdx.z * nk_per;
    const int nk = min(nk_all, ks0 + nk_per) - ks0;          // stages of this workgroup (indices below are relative)
    if (nk <= 0) return;
    const int k_start = ks0 * BK;
    // experiment (nemar_tune 2=256): workgroups 256 apart in dispatch order share a CU; start every other one late
    if ((p.dbg & 256) && (((blockIdx.x + gridDim.x * blockIdx.y) >> 8) & 1)) __builtin_amdgcn_s_sleep(20);
    const int tap_start = (int)fd_div((unsigned)k_start, p.fd_cs), ch_start = k_start - tap_start * Cs;

    if (wid >= MT) {
        // ================================ loader waves ================================
        const int ldr = wid - MT;
        // VEC: lane = (pixel group lane&31, row parity lane>>5), this loader's rows are ldr*8 .. ldr*8+7
        // else: lane = pixel of 64-pixel segment `ldr`, rows 0..15
        const int seg = ldr & 1, row0 = (ldr >> 1) * B_PER_LOADER;     // non-VEC: 64-pixel segment and first row of this loader
        const int pix = VEC ? p0 + 4 * (lane & 31) : p0 + seg * 64 + lane;
        const bool pvalid = pix < p.P;
        const unsigned upix = pvalid ? (unsigned)pix : 0u;
        const unsigned n = fd_div(upix, p.fd_ohw);
        const unsigned rem = upix - n * (unsigned)(p.OH * p.OW);
        const unsigned oy = fd_div(rem, p.fd_ow);
        const int by = (int)oy * p.sy, bx = (int)(rem - oy * (unsigned)p.OW) * p.sx;
        const int rofs = VEC ? ldr * (BK / NL) + (lane >> 5) : row0;  // first reduction row (channel offset in the stage) of this lane
        const float* s0n = p.src0 + ((size_t)n * p.C0 + rofs) * HW;
        const float* s1n = p.C1 ? p.src1 + ((size_t)n * p.C1 + rofs) * HW : s0n;
        const float* wsrc[A_PER_LOADER > 0 ? A_PER_LOADER : 1];
        int a_lds[A_PER_LOADER > 0 ? A_PER_LOADER : 1];
#pragma unroll
        for (int q = 0; q < A_PER_LOADER; ++q) {
            const int inst = ldr * A_PER_LOADER + q;
            const int e = inst * 256 + lane * 4;
            const int blk = e / (BM * 4), m = (e - blk * (BM * 4)) >> 2;
            wsrc[q] = p.wp + ((size_t)blk * p.Mpad + m0 + m) * 4;
            a_lds[q] = inst * 256;
        }
        // The loader is a state machine over consecutive stages: everything that repeats is a pointer bump.  (Per-stage
        // tap-table lookups in the kernel arguments, divisions and border arithmetic between the loads made the loader
        // the last wave at the stage barrier: two extra scalar branches in this loop cost the whole kernel 7 %.)
        int a_buf = 0;                                        // LDS ring slot of the next stage
        int tap_t = tap_start, ch0 = ch_start;                // tap / channel offset of the next stage
        // Per tap, every lane resolves ONCE where its texels of channel 0 live (`lbase`) and how far apart consecutive
        // channels are (`lstride`); per stage only `lp` moves.  Lanes whose tap falls outside a zero-padded source read the
        // zero page with stride 0.  Reflect data gradient (p.rf, see reflect_aux_kernel): lanes on a border row / in a
        // border column group read the pre-folded side buffers instead of gy, with those buffers' channel pitch.
        const float* lp = p.zero;                             // this lane's address for reduction row `rofs` of the next stage
        size_t lstride = 0;                                   // floats between consecutive channels at that address
        int sp_off = 0;                                       // y * Ws + x of this lane for the current tap (in-range lanes)
        bool inb = false;
        int ndyx = p.taps.dyx[__builtin_amdgcn_readfirstlane(tap_start)];   // offsets of the tap about to start
#pragma unroll
        for (int q = 0; q < A_PER_LOADER; ++q) wsrc[q] += (size_t)k_start * p.Mpad;
#define WS2_ENTER_TAP(ch_)                                                                                           \
        {                                                                                                            \
            const int dy = ndyx >> 16, dx = (int)(short)(ndyx & 0xffff);                                             \
            int y = by + dy, x = bx + dx;                                                                            \
            inb = pvalid;                                                                                            \
            if (p.border == BORDER_REFLECT) y = reflect(y, p.Hs);                                                    \
            else inb = inb && (unsigned)y < (unsigned)p.Hs;                                                          \
            if (VEC) x = (p.dbg & 2048) ? (min(max(x, 0), p.Ws - 4) & ~3) : min(max(x, 0), p.Ws - 4);  /* 2048: aligned-B ablation */ \
            else if (p.border == BORDER_REFLECT) x = reflect(x, p.Ws);                                               \
            else inb = inb && (unsigned)x < (unsigned)p.Ws;                                                          \
            sp_off = inb ? y * p.Ws + x : 0;                                                                         \
            const float* lbase = inb ? s0n + sp_off : p.zero;                                                        \
            lstride = inb ? (size_t)HW : (size_t)0;                                                                  \
            if (VEC && p.rf && pvalid) {                                                                             \
                const bool top = dy == 1 && by == 1, bot = dy == -1 && by == p.Hs - 2;                               \
                if (top || bot) {            /* whole row pre-folded: [top|bot][dx+1][n][k][Ws] */                   \
                    lstride = (size_t)p.Ws;                                                                          \
                    lbase = p.rf_row + ((((size_t)(bot ? 3 : 0) + (dx + 1)) * p.N + n) * p.C0 + rofs) * p.Ws + x;     \
                } else if (inb && ((dx == 1 && bx == 0) || (dx == -1 && bx == p.Ws - 4))) {                         \
                    /* border column group pre-folded: [left|right][n][k][Hs][4] */                                  \
                    lstride = (size_t)p.Hs * 4;                                                                      \
                    lbase = p.rf_col + ((((size_t)(dx == 1 ? 0 : 1) * p.N + n) * p.C0 + rofs) * p.Hs + y) * 4;       \
                }                                                                                                    \
            }                                                                                                        \
            lp = lbase + (size_t)(ch_) * lstride;                                                                    \
            ndyx = p.taps.dyx[__builtin_amdgcn_readfirstlane(min(tap_t + 1, p.taps.n - 1))];                         \
        }
        // the first tap may be entered mid-way (split reductions): ch_start channels in, possibly already in source 1
        WS2_ENTER_TAP(ch0 < p.C0 ? ch0 : 0);
        if (ch0 >= p.C0 && p.C1) lp = inb ? s1n + sp_off + (size_t)(ch0 - p.C0) * HW : p.zero;
#define WS2_ISSUE_NEXT()                                                                                             \
        {                                                                                                            \
            /* ablations (nemar_tune key 2): 1024 = no A loads, 512 = no B loads (results are garbage; timing only) */       \
            if (!(p.dbg & 1024)) {                                                                                   \
                _Pragma("unroll") for (int q = 0; q < A_PER_LOADER; ++q) {                                           \
                    glds_b128(wsrc[q], As0 + a_buf * A_FLOATS + a_lds[q]);                                           \
                    wsrc[q] += (size_t)BK * p.Mpad;                                                                  \
                }                                                                                                    \
            }                                                                                                        \
            if (p.dbg & 512) {                                                                                       \
            } else if (VEC) {                                                                                        \
                _Pragma("unroll") for (int i = 0; i < B_PER_LOADER; ++i)                                             \
                    glds_b128(lp + (size_t)(2 * i) * lstride, Bs0 + a_buf * B_FLOATS + (ldr * B_PER_LOADER + i) * 256); \
            } else {                                                                                                 \
                _Pragma("unroll") for (int r = 0; r < B_PER_LOADER; ++r)                                             \
                    glds_b32(lp + (size_t)r * lstride, Bs0 + a_buf * B_FLOATS + (row0 + r) * LDB + seg * 64);        \
            }                                                                                                        \
            a_buf = a_buf + 1 == W2_NBUF ? 0 : a_buf + 1;                                                            \
            ch0 += BK;                                                                                               \
            lp += (size_t)BK * lstride;                                                                              \
            if (ch0 == p.C0 && p.C1) lp = inb ? s1n + sp_off : p.zero;                                               \
            if (ch0 == Cs) {                                                                                         \
                ch0 = 0;                                                                                             \
                ++tap_t;                                                                                             \
                WS2_ENTER_TAP(0);                                                                                    \
            }                                                                                                        \
        }
#define WS2_ISSUE(ks_) WS2_ISSUE_NEXT()     /* stages are issued strictly in order */
#define WS2_WAIT_ONE_IN_FLIGHT() \
        __builtin_amdgcn_s_waitcnt(0x0F70 | (LOADS_PER_STAGE & 15) | ((LOADS_PER_STAGE >> 4) << 14))
        if (SPB == 2) {
            WS2_ISSUE(0);
            if (nk > 1) WS2_ISSUE(1);
            wait_vmem();
            __builtin_amdgcn_s_barrier();             // stages 0, 1 are in LDS
            if (nk > 2) WS2_ISSUE(2);
            if (nk > 3) WS2_ISSUE(3);
            for (int S = 0; 2 * S < nk; ++S) {
                wait_vmem();                          // stages 2S+2, 2S+3 have landed (a whole 32-row half to do so)
                __builtin_amdgcn_s_barrier();         // and every MFMA wave has finished reading stages 2S, 2S+1
                if (2 * S + 4 < nk) WS2_ISSUE(2 * S + 4);
                if (2 * S + 5 < nk) WS2_ISSUE(2 * S + 5);
            }
            return;
        }
        // RING - 1 stages are issued before anything is consumed; the barrier of iteration ks needs stage ks + 1 landed, the
        // (up to RING - 2) stages issued after it may stay in flight (counted s_waitcnt: loads retire in issue order)
#define WS2_WAIT_IN_FLIGHT(n_)                                                                                        \
        {                                                                                                            \
            const int ns_ = (p.dbg & (512 | 1024)) ? 0 : (n_);     /* ablated loads: the counts below would be wrong */ \
            if (ns_ <= 0) wait_vmem();                                                                               \
            else if (ns_ == 1) WS2_WAIT_ONE_IN_FLIGHT();                                                             \
            else if (ns_ == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * LOADS_PER_STAGE) & 15) | (((2 * LOADS_PER_STAGE) >> 4) << 14)); \
            else __builtin_amdgcn_s_waitcnt(0x0F70 | ((3 * LOADS_PER_STAGE) & 15) | (((3 * LOADS_PER_STAGE) >> 4) << 14)); \
        }
        static_assert(3 * LOADS_PER_STAGE < 64, "vmcnt is a 6-bit counter");
        int issued = 0;
        for (; issued < W2_NBUF - 1 && issued < nk; ++issued) WS2_ISSUE(issued);
        WS2_WAIT_IN_FLIGHT(issued - 1);
        __builtin_amdgcn_s_barrier();                 // stage 0 is in LDS
        if (issued < nk) { WS2_ISSUE(issued); ++issued; }
#ifdef NEMAR_TIMELINE
        long long lts[4][4];
        const bool lprobe = p.tl != nullptr && bx == 0 && by_ == 0;
#define WS2_LSTAMP(i_) if (lprobe && ks >= 40 && ks < 44) lts[ks - 40][i_] = clock64();
#else
#define WS2_LSTAMP(i_)
#endif
        for (int ks = 0; ks < nk; ++ks) {
            // stages ks + 2 .. issued - 1 may stay in flight; stage ks + 1 must have landed
            WS2_LSTAMP(0)
            WS2_WAIT_IN_FLIGHT(issued - (ks + 2));
            WS2_LSTAMP(1)
            if (!(p.dbg & 4)) __builtin_amdgcn_s_barrier();   // also: every MFMA wave has finished reading buffer ks % NBUF
            WS2_LSTAMP(2)
            if (issued < nk) { WS2_ISSUE(issued); ++issued; }
            WS2_LSTAMP(3)
        }
#ifdef NEMAR_TIMELINE
        if (lprobe && lane == 0 && nk >= 44) {
            long long* o = p.tl + (MT + ldr) * 24;          // behind the MFMA waves' 4 x 6 stamps per stage
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) o[i * 6 + j] = lts[i][j];
        }
#endif
#undef WS2_LSTAMP
#undef WS2_WAIT_IN_FLIGHT
#undef WS2_ISSUE
#undef WS2_ISSUE_NEXT
#undef WS2_ENTER_TAP
#undef WS2_WAIT_ONE_IN_FLIGHT
        return;
    }

    // ================================ MFMA waves ================================
    if (p.dbg & 128) __builtin_amdgcn_s_setprio(3);           // experiment: issue priority over the loader waves
    const int l31 = lane & 31, lhi = lane >> 5;
    const int a_off = (lhi * BM + wid * 32 + l31) * 4;        // + kg * 2*BM*4 floats for the second 8-row group
    const int b_off = lhi * LDB + 4 * l31;                    // + (8*kg + 2*s) * LDB
    // VEC: is this lane's 4-pixel group the first / last of a source row?  (those are the groups the loaders clamp)
    bool first_grp = false, last_grp = false;
    if (VEC) {
        const unsigned gp = (unsigned)min(p0 + 4 * l31, p.P - 1);
        const unsigned gn = fd_div(gp, p.fd_ohw);
        const unsigned grem = gp - gn * (unsigned)(p.OH * p.OW);
        const unsigned gox = grem - fd_div(grem, p.fd_ow) * (unsigned)p.OW;
        first_grp = gox == 0;
        last_grp = (int)gox == p.OW - 4;
    }
    const bool refl = p.border == BORDER_REFLECT;
    // sign of every tap's dx, 2 bits per tap (<= 32 taps on the VEC path: |dx| <= 1 means at most a 3x3 footprint... any
    // tap count up to 32 is representable); current tap / channel offset advance with the stages
    unsigned long long dxbits = 0ull;
    int tap_i = tap_start, tap_ch = ch_start;
    if (VEC)
        for (int t = 0; t < p.taps.n && t < 32; ++t)
            dxbits |= (unsigned long long)(p.taps.dx[t] < 0 ? 1u : p.taps.dx[t] > 0 ? 2u : 0u) << (2 * t);
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 a0, a1, b0[4], b1[4];

#define WS2_READ(buf_, kg_, A_, B_)                                                                        \
    {   /* (an ablation switch around these reads makes hipcc count the LDS waits of the MFMA blocks conservatively: measured \
           once — MFMA-only skeleton 302 us vs 370 us for the full kernel, gpurun_out/abl2 — and removed again) */  \
        const float* sb = Bs0 + (buf_) * B_FLOATS + (kg_) * (8 * LDB) + b_off;                                 \
        if (!ADIR) A_ = *reinterpret_cast<const f32x4*>(As0 + (buf_) * A_FLOATS + (kg_) * (2 * BM * 4) + a_off); \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) B_[s] = *reinterpret_cast<const f32x4*>(sb + 2 * s * LDB); \
    }
#define WS2_MFMA(A_, B_)                                                                                   \
    if (!(p.dbg & 2)) {                                                                                        \
        if (VEC && fix_l) {                                                                                    \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                    \
                const f32x4 v = B_[s];                                                                         \
                B_[s][0] = first_grp ? (refl ? v[1] : 0.f) : v[0];                                             \
                B_[s][1] = first_grp ? v[0] : v[1];                                                            \
                B_[s][2] = first_grp ? v[1] : v[2];                                                            \
                B_[s][3] = first_grp ? v[2] : v[3];                                                            \
            }                                                                                                  \
        }                                                                                                      \
        if (VEC && fix_r) {                                                                                    \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                    \
                const f32x4 v = B_[s];                                                                         \
                B_[s][0] = last_grp ? v[1] : v[0];                                                             \
                B_[s][1] = last_grp ? v[2] : v[1];                                                             \
                B_[s][2] = last_grp ? v[3] : v[2];                                                             \
                B_[s][3] = last_grp ? (refl ? v[2] : 0.f) : v[3];                                              \
            }                                                                                                  \
        }                                                                                                      \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                          \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                      \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[s], B_[s][t], acc[t], 0, 0, 0);               \
    }
    // timeline probe (nemar_tune_ptr): s_memtime stamps of stages 40..43 of every MFMA wave of workgroup (0,0), kept in
    // scalar registers and written once at the end: 6 stamps per stage = loop top | reads issued | MFMA block 1 issued |
    // lgkmcnt(0) | barrier passed | MFMA block 2 issued
    long long ts[4][6];
    const bool probe = p.tl != nullptr && bx == 0 && by_ == 0;
#ifdef NEMAR_TIMELINE      /* the probe's s_memtime + waits perturb the loop's wait counts: compiled in on demand only */
#define WS2_STAMP(i_)                                                     \
    if (probe && ks >= 40 && ks < 44) {                                       \
        const long long c_ = clock64();                                       \
        if (ks == 40) ts[0][i_] = c_; else if (ks == 41) ts[1][i_] = c_;      \
        else if (ks == 42) ts[2][i_] = c_; else ts[3][i_] = c_;               \
    }
#else
#define WS2_STAMP(i_)
#endif
    if (SPB == 2) {
        bool fix_l = false, fix_r = false;
#define WS2_STAGE_FLAGS()                                                       \
        {                                                                       \
            const unsigned dc_ = VEC ? (unsigned)(dxbits >> (2 * tap_i)) & 3u : 0u; \
            fix_l = dc_ == 1u;                                                  \
            fix_r = dc_ == 2u;                                                  \
            if (VEC) {                                                          \
                tap_ch += BK;                                                   \
                if (tap_ch >= Cs) { tap_ch -= Cs; ++tap_i; }                    \
            }                                                                   \
        }
        __builtin_amdgcn_s_barrier();                 // stages 0, 1 are in LDS
        WS2_READ(0, 0, a0, b0);
        for (int s0 = 0; s0 < nk; s0 += 2) {
            const int slot0 = s0 & 3, slot1 = (s0 + 1) & 3;
            const bool has1 = s0 + 1 < nk;
            WS2_STAGE_FLAGS();                        // border flags of stage s0
            WS2_READ(slot0, 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            WS2_MFMA(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (has1) {
                WS2_READ(slot1, 0, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                WS2_MFMA(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                WS2_STAGE_FLAGS();                    // border flags of stage s0 + 1
                WS2_READ(slot1, 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                WS2_MFMA(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): this wave is done reading both stages of the half
            __builtin_amdgcn_s_barrier();             // the next half has landed; this one goes back to the loaders
            if (s0 + 2 < nk) WS2_READ((s0 + 2) & 3, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            WS2_MFMA(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WS2_STAGE_FLAGS
    } else {
    // ADIR: this lane's A words of the two 8-row groups of a stage: gA + stage * (4 * Mpad * 4) + {0, 2 * Mpad * 4} floats
    const float* gA = p.wp + ((size_t)(ks0 * 4 + lhi) * p.Mpad + m0 + wid * 32 + l31) * 4;
    const size_t gA_kg = (size_t)2 * p.Mpad * 4, gA_stage = (size_t)4 * p.Mpad * 4;
    f32x4 na0, na1, a1n;
    if (ADIR) {
        na0 = *reinterpret_cast<const f32x4*>(gA);
        na1 = *reinterpret_cast<const f32x4*>(gA + gA_kg);
        gA += gA_stage;
    }
    __builtin_amdgcn_s_barrier();                     // stage 0 is in LDS
    WS2_READ(0, 0, a0, b0);
    if (ADIR) {
        a0 = na0; a1n = na1;
        if (nk > 1) {
            na0 = *reinterpret_cast<const f32x4*>(gA);
            na1 = *reinterpret_cast<const f32x4*>(gA + gA_kg);
            gA += gA_stage;
        }
    }
    int buf = 0;
    for (int ks = 0; ks < nk; ++ks) {
        // the tap of this stage decides which border groups need patching (wave-uniform, from registers: a table
        // lookup in the kernel arguments here costs a scalar-memory round trip per stage)
        const unsigned dcode = VEC ? (unsigned)(dxbits >> (2 * tap_i)) & 3u : 0u;
        const bool fix_l = dcode == 1u, fix_r = dcode == 2u;
        if (VEC) {
            tap_ch += BK;
            if (tap_ch >= Cs) { tap_ch -= Cs; ++tap_i; }
        }
        WS2_STAMP(0)
        WS2_READ(buf, 1, a1, b1);
        if (ADIR) a1 = a1n;
        __builtin_amdgcn_sched_barrier(0);
        WS2_STAMP(1)
        WS2_MFMA(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        WS2_STAMP(2)
        __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): this wave is done reading buffer `buf`
        WS2_STAMP(3)
        if (!(p.dbg & 4)) __builtin_amdgcn_s_barrier();   // stage ks+1 has landed; buffer `buf` goes back to the loaders (ablation 4: none)
        WS2_STAMP(4)
        buf = buf + 1 == W2_NBUF ? 0 : buf + 1;
        if (ks + 1 < nk) {
            WS2_READ(buf, 0, a0, b0);
            if (ADIR) {                               // next stage's A words (fetched a stage ago); fetch the one after
                a0 = na0; a1n = na1;
                if (ks + 2 < nk) {
                    na0 = *reinterpret_cast<const f32x4*>(gA);
                    na1 = *reinterpret_cast<const f32x4*>(gA + gA_kg);
                    gA += gA_stage;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        WS2_MFMA(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        WS2_STAMP(5)
    }
    }   // SPB == 1
#undef WS2_READ
#undef WS2_MFMA
#undef WS2_STAMP
#ifdef NEMAR_TIMELINE
    if (SPB == 1 && probe && lane == 0 && nk >= 44) {
        long long* o = p.tl + wid * 24;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 6; ++j) o[i * 6 + j] = ts[i][j];
    }
#else
    (void)ts; (void)probe;
#endif

    // ---- epilogue: lane owns pixels p0 + 4*l31 + {0..3} (tile t -> pixel t) of 16 channel rows ---------------------
    const size_t oplane = (size_t)p.OHf * p.OWf;
    const int M1 = p.M - p.M0;
    const int opix0 = p0 + 4 * l31;
    if (opix0 >= p.P) return;
    // split reduction: this split's partial tile goes to its own slab (no bias / activation / second destination then)
    float* const d0 = gridDim.z > 1 ? p.part + (size_t)blockIdx.z * (size_t)p.part_stride : p.dst0;
    // 16-byte stores when the 4 pixels are consecutive, in range and aligned in the destination
    const bool vec = p.osx == 1 && p.osy == 1 && p.oox == 0 && p.ooy == 0 && (p.OW & 3) == 0 &&
                     p.OW == p.OWf && p.OH == p.OHf && opix0 + 3 < p.P &&
                     ((reinterpret_cast<uintptr_t>(d0) | reinterpret_cast<uintptr_t>(p.dst1)) & 15) == 0;
    if (vec) {
        const unsigned n = fd_div((unsigned)opix0, p.fd_ohw);
        const unsigned rem = (unsigned)opix0 - n * (unsigned)(p.OH * p.OW);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < p.M) {
                const float bv = (p.bias && blockIdx.z == 0) ? p.bias[m] : 0.f;   // split reductions: slab 0 carries the bias
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = apply_act(acc[t][r] + bv, p.act, p.slope);
                float* dst = (m < p.M0) ? (d0 ? d0 + ((size_t)n * p.M0 + m) * oplane + rem : nullptr)
                                        : p.dst1 + ((size_t)n * M1 + (m - p.M0)) * oplane + rem;
                if (dst) *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int opix = opix0 + t;
        if (opix >= p.P) continue;
        const unsigned n = fd_div((unsigned)opix, p.fd_ohw);
        const unsigned rem = (unsigned)opix - n * (unsigned)(p.OH * p.OW);
        const unsigned oy = fd_div(rem, p.fd_ow);
        const unsigned ox = rem - oy * (unsigned)p.OW;
        const size_t sp = (size_t)((int)oy * p.osy + p.ooy) * p.OWf + ((int)ox * p.osx + p.oox);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < p.M) {
                float v = acc[t][r];
                if (p.bias && blockIdx.z == 0) v += p.bias[m];
                v = apply_act(v, p.act, p.slope);
                if (m < p.M0) {
                    if (d0) d0[((size_t)n * p.M0 + m) * oplane + sp] = v;
                } else {
                    p.dst1[((size_t)n * M1 + (m - p.M0)) * oplane + sp] = v;
                }
            }
        }
    }
}

static int g_ring = 3;           // tuning switch (key 18): LDS ring depth of the wave-specialised 16-byte-load kernel (3, 4, 5)
static int g_mt8 = 0;            // tuning switch (key 17): 256x128 tiles in the wave-specialised kernel: 0 off, 1 = 2 loader waves, 2 = 4 loader waves
static int g_adir = 0;           // tuning switch (key 16): MFMA waves fetch their A fragments straight from global memory (1; measured 3 %
                                 // SLOWER: 370.8 vs 358.4 us on the 256->256 3x3 layer, gpurun_out/r2g) / through LDS (0, default)
static int g_xcd_map = 1;        // tuning switch (key 15): XCD-aware workgroup -> tile mapping in the wave-specialised igemm
// key 20: 3x3 / stride-1 layers with >= 128 output channels run on the bf16 matrix pipe with three-way split operands
// (conv_split16.hip) whenever the caller has registered a scratch arena large enough for the split source planes
static int g_split16 = 1;
// key 23: the split-16 route needs work to amortise its extra launches (max pass, split pass, slab sum): layers below this many
// million multiply-adds (default 2000 = 4 GFLOP, ~40 us on the exact-fp32 kernels) stay on those — BASELINE config 1's 32x32
// batch-1 resblocks (0.6 GMAC) lost 2 ms per step to launch overhead on the split-16 route
static long long g_split16_min_mmac = 2000;
static int g_split16_variant = 4;   // key 21: 4 fp16 x 3 products (default), 3 bf16 x 6 products, 0 bf16 x 6 on the first-generation
                                // kernel with loader waves (kept for the A/B numbers in DESIGN.md)
// which kernel family served the last conv call of this thread (nemar_last_route; tests and tools): 0 exact-fp32 implicit GEMM,
// 1 narrow (<= 4 channel) VALU kernels, 2 split-16 kernel of the wide residual-block layers, 3 general 16-bit-pipe kernels
static thread_local int g_last_route = 0;
static int g_config_epoch = 0;      // bumped by every nemar_tune / nemar_set_scratch: routes (and packed-weight formats) may have changed
static int g_k7 = 1;               // key 33: the 7x7 stem / head layers (<= 4 channels on one side) on the 16-bit matrix pipe (conv_k7.hip)
static int g_s16g = 1;             // key 24: general layers on the 16-bit matrix pipe with the in-kernel operand split (conv_s16g.hip)
static int g_s16g_wgrad_first = 0;  // key 26: 1 = the in-kernel-split weight gradient also takes the wide residual-block layers (stand-alone 374 vs
                                    // 393 us per call, but 44.3 vs 41.8 ms per step inside the bench: off)
static int g_s16g_wgrad = 1;        // key 29: weight gradients on the in-kernel-split kernels (conv_s16g_wgrad.hip)
static int g_s16g_fold = 1;         // key 30: stride-1 reflect data gradients on the padded domain + fold
static long long g_s16g_min_mmac = 30;   // key 25: ... above this many million multiply-adds (tiny layers are launch-bound either way)
static thread_local void* t_scratch = nullptr;        // nemar_conv2d_*_ex: this call's scratch arena (nemar_conv_extras.scratch)
static thread_local size_t t_scratch_bytes = 0;
static thread_local void* t_gy_planes_out = nullptr;  // bwd_data_ex: where the pass that splits gy also leaves the weight gradient's planes
static thread_local size_t t_gy_planes_bytes = 0;
static thread_local const void* t_src2_planes = nullptr;      // bwd_weight_ex: those planes
static int g_split_act = 0;          // key 36: reduction-split forward layers with a fused ReLU / LeakyReLU (activation in the sum pass)
static int g_dual_gy = 1;            // key 35: the data-gradient call's split pass also writes the weight gradient's gy planes
static thread_local int t_gy_planes_written = 0;      // did the last bwd_data_ex call on this thread fill gy_planes_out?
#define g_scratch t_scratch
#define g_scratch_bytes t_scratch_bytes
static int g_reflect_aux = 1;    // tuning switch (key 8): 3x3 reflect data gradient folds the border into the main launch (1) / ring launch (0)
static int g_deterministic = 1;  // tuning switch (key 14): 1 = split reductions go through per-split slabs summed in order (bitwise
                                 // reproducible backward pass), 0 = fp32 atomics in the weight / bias gradients (round-1 scheme)
static int g_ksplit = 1;         // tuning switch (key 12): allow reduction splits in the wave-specialised data gradient
static int g_nl4_scalar = 1;     // tuning switch (key 11): 4 loader waves for the gathered-B wave-specialised kernel
static int g_deep64 = 0;         // tuning switch (key 10): 4-deep LDS ring for every FAST 64x64 launch (default: ring launches only)
static int g_ws2_mt = 0;         // tuning switch (key 7): force the wave-specialised kernel's channel tile (1, 2, 4 x 32)
static int g_min_blocks = 384;   // tuning switch (key 6): workgroups below which the pixel/channel tile shrinks
static int g_cfg128 = 0;   // tuning switch (nemar_tune key 0): 0 = ws2 (all FAST shapes), 5 = ws2 without 16-byte B loads; 128x128 only: 1 = 4-wave, 2 = 8-wave, 3 = 256x128, 4 = ws gen 1
static int g_narrow = 1;   // tuning switch (key 3): route <=4-channel layers to the VALU kernels
static int g_dbg = 0;
static long long* g_tl = nullptr;
static int g_wgrad = 0;    // tuning switch (key 4): 0 = wave-specialised wide weight gradient, 1 = VGPR-staged kernel, 2 = wave-specialised without 16-byte source loads
static int g_wgrad_blocks = 512;   // tuning switch (key 5): workgroups targeted by the pixel split
static int g_lds_pad = 0;  // tuning switch: extra dynamic LDS bytes per workgroup (limits workgroups per CU)

template <int WM, int WN, int TM, int TN>
void launch_igemm_cfg(const IgemmParams& p, bool fast, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    dim3 grid(nemar_cdiv(p.P, BN), nemar_cdiv(p.M, BM), p.ksplit), block(WM * WN * 64);
    if (fast)
        hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, true>), grid, block, g_lds_pad, st, p);
    else
        hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, false>), grid, block, g_lds_pad, st, p);
}

template <int MT>
void launch_ws2(const IgemmParams& p, bool vec, hipStream_t st) {
    dim3 grid(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 32 * MT), p.ring_p ? 1 : p.ksplit), block((MT + 2) * 64);
    if (vec && g_adir) hipLaunchKernelGGL((igemm_ws2_kernel<MT, true, 1, 2, true>), grid, block, g_lds_pad, st, p);
    else if (vec && g_ring == 4) hipLaunchKernelGGL((igemm_ws2_kernel<MT, true, 1, 2, false, 4>), grid, block, g_lds_pad, st, p);
    else if (vec && g_ring == 5) hipLaunchKernelGGL((igemm_ws2_kernel<MT, true, 1, 2, false, 5>), grid, block, g_lds_pad, st, p);
    else if (vec) hipLaunchKernelGGL((igemm_ws2_kernel<MT, true>), grid, block, g_lds_pad, st, p);
    else hipLaunchKernelGGL((igemm_ws2_kernel<MT, false>), grid, block, g_lds_pad, st, p);
}

// Tile selection.  The channel tile follows M; the pixel tile shrinks when the grid would leave most of the 256 CUs
// idle (the small-spatial discriminator / bottleneck layers): ~2 workgroups per CU is the target.
struct TileChoice { int bm, bn; };
TileChoice igemm_tile(int M, int P, int stages, int ksplit = 1) {
    const int kMinBlocks = g_min_blocks;
    TileChoice t;
    if (M > 64) {
        // 128x128 (wave-specialised for FAST shapes) needs ~1.5 workgroups per CU — or, measured on D's k4 layers, just
        // ~1 per CU when the reduction is deep enough (>= 64 stages) to amortise the lock-step prologue/epilogue
        t.bm = 128; t.bn = 128;
        const long long tiles = (long long)nemar_cdiv(M, 128) * nemar_cdiv(P, 128) * ksplit;
        const int need = (stages / ksplit >= 64 && kMinBlocks > 200) ? 200 : kMinBlocks;
        if (tiles < need) { t.bm = 64; t.bn = 64; }
    } else if (M > 32) {
        t.bm = 64; t.bn = 128;
        if ((long long)nemar_cdiv(P, 128) < kMinBlocks) t.bn = 64;
    } else {
        t.bm = 32; t.bn = 256;
        if ((long long)nemar_cdiv(P, 256) < kMinBlocks) t.bn = 128;
    }
    return t;
}
int igemm_mpad(int M) { return M > 32 ? nemar_cdiv(M, 256) * 256 : 32; }

// Does this launch go to the wave-specialised kernel (128 pixels x 32*MT channels), and with 16-byte B loads?  Measured:
// MT = 4 beats every generic configuration on layers big enough for 128x128 tiles; MT = 1, 2 (fewer MFMAs per staged B
// tile) lose to the generic 64x64 / 32x256 kernels and are only reachable through the tuning switch.
bool route_ws2(const IgemmParams& p, bool* vec_out) {
    const int Cs = p.C0 + p.C1;
    const bool fast = (Cs % BK == 0) && (p.C0 % BK == 0);
    const TileChoice t = igemm_tile(p.M, p.P, nemar_cdiv(p.Kred, BK), p.ring_p ? 1 : p.ksplit);
    if (!(fast && !p.ring_p && (g_cfg128 == 0 || (g_cfg128 >= 5 && g_cfg128 <= 7)) && (t.bm == 128 || g_ws2_mt))) return false;
    bool vec = g_cfg128 != 5 && p.sx == 1 && (p.OW & 3) == 0 && p.Ws == p.OW && p.Ws >= 4 && p.taps.n <= 32;
    for (int i = 0; i < p.taps.n && vec; ++i) vec = p.taps.dx[i] >= -1 && p.taps.dx[i] <= 1;
    *vec_out = vec;
    return true;
}

void launch_igemm(const IgemmParams& p, hipStream_t st) {
    const int Cs = p.C0 + p.C1;
    const bool fast = (Cs % BK == 0) && (p.C0 % BK == 0);
    TileChoice t = igemm_tile(p.M, p.P, nemar_cdiv(p.Kred, BK), p.ring_p ? 1 : p.ksplit);
    if (p.ring_p) {   // few pixels, full reduction depth: small tiles so the launch spreads over the CUs (generic kernel only)
        t.bm = p.M > 32 ? 64 : 32;
        t.bn = p.M > 32 ? 64 : 128;
    }
    bool vec = false;
    if (route_ws2(p, &vec)) {
        int mt = g_ws2_mt ? g_ws2_mt : 4;
        if (32 * mt > p.Mpad) mt = p.Mpad / 32;       // a forced tile must not read past the packed rows (M <= 32 packs 32 rows)
        IgemmParams q = p;
        q.xcd = (g_xcd_map && nemar_cdiv(p.P, 128) % 8 == 0) ? 1 : 0;
        const IgemmParams& p = q;
        if (mt == 4 && !vec && g_nl4_scalar && g_adir)     // gathered (non-VEC) B tile: 4 loader waves share the 32 4-byte loads
            hipLaunchKernelGGL((igemm_ws2_kernel<4, false, 1, 4, true>), dim3(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 128), p.ksplit),
                               dim3(8 * 64), g_lds_pad, st, p);
        else if (mt == 4 && !vec && g_nl4_scalar)
            hipLaunchKernelGGL((igemm_ws2_kernel<4, false, 1, 4>), dim3(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 128), p.ksplit),
                               dim3(8 * 64), g_lds_pad, st, p);
        else if (mt == 4 && vec && g_cfg128 == 7)     // experiment: 4 loader waves
            hipLaunchKernelGGL((igemm_ws2_kernel<4, true, 1, 4>), dim3(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 128)), dim3(8 * 64),
                               g_lds_pad, st, p);
        else if (mt == 4 && vec && g_cfg128 == 6)     // experiment: one barrier per 32 reduction rows
            hipLaunchKernelGGL((igemm_ws2_kernel<4, true, 2>), dim3(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 128)), dim3(6 * 64),
                               g_lds_pad, st, p);
        else if (mt == 4 && vec && g_mt8 && p.M > 128 && p.ksplit == 1) {
            // 256 channels x 128 pixels per workgroup (8 MFMA waves): one workgroup per CU, the B tile staged once per pixel
            // tile instead of once per 128-channel half, 24 instead of 32 KiB of global->LDS traffic per 2 x 16 reduction rows
            const dim3 g8(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 256));
            if (g_mt8 == 2) hipLaunchKernelGGL((igemm_ws2_kernel<8, true, 1, 4>), g8, dim3(12 * 64), g_lds_pad, st, p);
            else hipLaunchKernelGGL((igemm_ws2_kernel<8, true, 1, 2>), g8, dim3(10 * 64), g_lds_pad, st, p);
        }
        else if (mt == 4) launch_ws2<4>(p, vec, st);
        else if (mt == 2) launch_ws2<2>(p, vec, st);
        else launch_ws2<1>(p, vec, st);
        return;
    }
    if (t.bm == 128 && g_cfg128 == 3 && p.M >= 256) launch_igemm_cfg<4, 2, 2, 2>(p, fast, st);   // 256 x 128, 8 waves of 64x64
    else if (t.bm == 128 && fast && g_cfg128 == 4)                                     // 128 x 128, 8 MFMA + 2 loader waves
        hipLaunchKernelGGL(igemm_ws_kernel, dim3(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 128)), dim3(WS_NT), g_lds_pad, st, p);
    else if (t.bm == 128 && g_cfg128 == 1) launch_igemm_cfg<2, 2, 2, 2>(p, fast, st); // 128 x 128, 4 waves of 64x64
    else if (t.bm == 128) launch_igemm_cfg<2, 4, 2, 1>(p, fast, st);                  // 128 x 128, 8 waves of 64x32
    else if (t.bm == 64 && t.bn == 128) launch_igemm_cfg<1, 4, 2, 1>(p, fast, st);    // 64 x 128
    else if (t.bm == 64 && fast && (p.ring_p || g_deep64))                            // 64 x 64, 4-deep LDS ring
        hipLaunchKernelGGL((igemm_kernel<2, 2, 1, 1, true, 4>), dim3(nemar_cdiv(p.P, 64), nemar_cdiv(p.M, 64), p.ksplit),
                           dim3(256), g_lds_pad, st, p);
    else if (t.bm == 64) launch_igemm_cfg<2, 2, 1, 1>(p, fast, st);                   // 64 x 64
    else if (t.bn == 256) launch_igemm_cfg<1, 4, 1, 2>(p, fast, st);                  // 32 x 256
    else launch_igemm_cfg<1, 4, 1, 1>(p, fast, st);                                   // 32 x 128
}

// packed weights are padded to a multiple of 128 channels (32 when M <= 32) so every tile config can read them
constexpr int ZERO_PAGE = 64;   // floats of zeros appended to every packed-weight buffer (target of masked gathers)
size_t packed_core_floats(int M, int Kred) { return (size_t)nemar_cdiv(Kred, BK) * BK * (size_t)igemm_mpad(M); }
size_t packed_floats(int M, int Kred) { return packed_core_floats(M, Kred) + ZERO_PAGE; }

// the same pack as a job of a weight-pack plan (pack_plan.h): arguments from device memory, grid.z = job
struct ExactPackArgs {
    const float* w; float* wp;
    int M, Mpad, Cs, Kred, KredPad, wsm, wsc, zero_tail;
    int gx, gy;
    int wofs[MAX_TAPS];
};
__device__ __forceinline__ void exact_pack_body(const ExactPackArgs& a, int bx, int, int gx) {
    const int core = a.KredPad * a.Mpad, total = core + a.zero_tail;
    for (int idx = bx * 256 + threadIdx.x; idx < total; idx += gx * 256) {
        const int blk = idx / (a.Mpad * 4), within = idx - blk * (a.Mpad * 4);
        const int m = within >> 2, s = within & 3;
        const int kk = 8 * (blk >> 1) + 2 * s + (blk & 1);
        float v = 0.f;
        if (idx < core && m < a.M && kk < a.Kred) {
            const int t = kk / a.Cs, ch = kk - t * a.Cs;
            v = a.w[(size_t)m * a.wsm + (size_t)ch * a.wsc + a.wofs[t]];
        }
        a.wp[idx] = v;
    }
}
NEMAR_PACK_MULTI(exact_pack_multi_kernel, ExactPackArgs, exact_pack_body, 256)
void exact_pack_multi(const void* jobs, int njobs, int gx, int gy, hipStream_t st) {
    hipLaunchKernelGGL(exact_pack_multi_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, (const ExactPackArgs*)jobs);
}
struct RegExactPack {
    RegExactPack() { nemar_pack_register(PACK_FAM_EXACT, sizeof(ExactPackArgs), exact_pack_multi); }
} g_reg_exact_pack;

void launch_pack(const float* w, float* wp, int M, int Cs, int wsm, int wsc, const TapTable& taps, hipStream_t st) {
    const int Kred = taps.n * Cs;
    const int KredPad = nemar_cdiv(Kred, BK) * BK;
    const int Mpad = igemm_mpad(M);
    const int total = KredPad * Mpad + ZERO_PAGE;
    if (nemar_pack_recording()) {
        ExactPackArgs a;
        a.w = w; a.wp = wp; a.M = M; a.Mpad = Mpad; a.Cs = Cs; a.Kred = Kred; a.KredPad = KredPad; a.wsm = wsm; a.wsc = wsc;
        a.zero_tail = ZERO_PAGE; a.gx = nemar_stream_grid(total, 256); a.gy = 1;
        for (int i = 0; i < MAX_TAPS; ++i) a.wofs[i] = i < taps.n ? taps.wofs[i] : 0;
        nemar_pack_record_job(PACK_FAM_EXACT, &a, a.gx, 1);
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, w, wp, M, Mpad, Cs,
                       Kred, KredPad, wsm, wsc, ZERO_PAGE, taps);
}

// ---- weight gradient ----------------------------------------------------------------------------------------
constexpr int WBK = 32;  // pixels per LDS stage

struct WgradParams {
    const float* src0; const float* src1; int C0, C1, Hs, Ws;
    const float* gy; int K, OH, OW;
    float* gw; int J;  // J = Cs * R * S columns, j = c*R*S + r*S + s (the tensor's own memory order)
    float* gb;         // optional [K]: += sum_pixels gy (bias gradient), folded into the A-tile loads of column-tile 0
    float* part;       // [splits][K*J] slabs then [splits][K] bias slabs (nullptr: fp32 atomics into gw / gb)
    float* partb;
    int N, P, sy, sx, R, S, pad, border;
    int pix_per_split;
    int dbg;   // ablation (nemar_tune key 2): 1 = skip staging loads, 2 = skip MFMAs, 8 = skip the atomic epilogue
    FastDiv fd_ohw, fd_ow;
};

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDA = BM + 1, LDB = BN + 1;  // odd stride: conflict-free transposing stores
    constexpr int ACOLS = BM / 8, BCOLS = BN / 8;
    __shared__ float As[2][WBK][LDA];
    __shared__ float Bs[2][WBK][LDB];
    __shared__ int s_jc[BN];   // source channel of column j (or -1: out of range)
    __shared__ int s_jt[BN];   // packed (dy << 16) | (dx & 0xffff)

    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * BM;
    const int j0 = blockIdx.y * BN;
    const int RS = p.R * p.S;
    const int HW = p.Hs * p.Ws, OHW = p.OH * p.OW;
    for (int i = tid; i < BN; i += 256) {
        const int j = j0 + i;
        int c = -1, tp = 0;
        if (j < p.J) {
            c = j / RS;
            const int t = j - c * RS;
            const int r = t / p.S, s = t - r * p.S;
            tp = ((r - p.pad) << 16) | ((s - p.pad) & 0xffff);
        }
        s_jc[i] = c;
        s_jt[i] = tp;
    }
    const int pbeg = blockIdx.z * p.pix_per_split;
    const int pend = min(p.P, pbeg + p.pix_per_split);
    const int prow = tid & 31, cgrp = tid >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();

    float ra[ACOLS], rb[BCOLS];
    float bsum[ACOLS];
#pragma unroll
    for (int i = 0; i < ACOLS; ++i) bsum[i] = 0.f;
    const bool do_bias = p.gb != nullptr && blockIdx.y == 0;
    auto load_stage = [&](int pb) {
        const int pix = pb + prow;
        const bool pv = pix < pend;
        const unsigned upix = pv ? (unsigned)pix : 0u;
        const unsigned n = fd_div(upix, p.fd_ohw);
        const unsigned rem = upix - n * (unsigned)OHW;
        const unsigned oy = fd_div(rem, p.fd_ow);
        const unsigned ox = rem - oy * (unsigned)p.OW;
        const float* g = p.gy + (size_t)n * p.K * OHW + rem;
#pragma unroll
        for (int i = 0; i < ACOLS; ++i) {
            const int m = m0 + cgrp + 8 * i;
            ra[i] = (pv && m < p.K) ? g[(size_t)m * OHW] : 0.f;
            bsum[i] += ra[i];
        }
        const int by = (int)oy * p.sy, bx = (int)ox * p.sx;
        const float* s0n = p.src0 + (size_t)n * p.C0 * HW;
        const float* s1n = p.C1 ? p.src1 + (size_t)n * p.C1 * HW : p.src0;
#pragma unroll
        for (int i = 0; i < BCOLS; ++i) {
            const int col = cgrp + 8 * i;
            const int c = s_jc[col];
            const int tp = s_jt[col];
            float v = 0.f;
            if (pv && c >= 0) {
                int y = by + (tp >> 16), x = bx + (int)(short)(tp & 0xffff);
                bool inb = true;
                if (p.border == BORDER_REFLECT) {
                    y = reflect(y, p.Hs);
                    x = reflect(x, p.Ws);
                } else {
                    inb = (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
                }
                if (inb) {
                    const float* base = (c < p.C0) ? s0n + (size_t)c * HW : s1n + (size_t)(c - p.C0) * HW;
                    v = base[y * p.Ws + x];
                }
            }
            rb[i] = v;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ACOLS; ++i) As[buf][prow][cgrp + 8 * i] = ra[i];
#pragma unroll
        for (int i = 0; i < BCOLS; ++i) Bs[buf][prow][cgrp + 8 * i] = rb[i];
    };

    const int wid = tid >> 6, lane = tid & 63;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nk = (pend - pbeg + WBK - 1) / WBK;
    if (nk > 0) {
        load_stage(pbeg);
        store_stage(0);
    }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk && !(p.dbg & 1)) load_stage(pbeg + (ks + 1) * WBK);
        if (!(p.dbg & 2))
#pragma unroll
        for (int k2 = 0; k2 < WBK / 2; ++k2) {
            const int kr = 2 * k2 + lhi;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[buf][kr][(wm * TM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[buf][kr][(wn * TN + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nk) store_stage(buf ^ 1);
        __syncthreads();
    }
    if (nk <= 0 || (p.dbg & 8)) return;
    float* const gw = p.part ? p.part + (size_t)blockIdx.z * ((size_t)p.K * p.J) : p.gw;
    if (do_bias) {
        // this thread summed gy over its pixel rows for channels cgrp + 8i; fold the 32 pixel lanes of each half-wave
        float* const gb = p.part ? p.partb + (size_t)blockIdx.z * p.K : p.gb;
#pragma unroll
        for (int i = 0; i < ACOLS; ++i) {
            float v = bsum[i];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const int m = m0 + cgrp + 8 * i;
            if (prow == 0 && m < p.K) {
                if (p.part) gb[m] = v;
                else atomicAdd(gb + m, v);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int jj = j0 + (wn * TN + j) * 32 + l31;
        if (jj >= p.J) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < p.K) {
                    if (p.part) gw[(size_t)m * p.J + jj] = acc[i][j][r];
                    else atomicAdd(gw + (size_t)m * p.J + jj, acc[i][j][r]);
                }
            }
    }
}

// Reflect data gradient, 3x3 / pad 1, without a ring launch.  The gradient of reflect-pad + conv w.r.t. texel (h,w) is the
// zero-padded data gradient plus, for the texels one step inside the border, the gradient of the padded texel they were
// mirrored to:  gx[1][w] gets  w[r=0] * gy[0]  on top of  w[r=0] * gy[2]  — i.e. tap dy = +1 must read gy[2] + gy[0] at output
// row 1 (and tap dy = -1 reads gy[H-3] + gy[H-1] at row H-2; columns likewise for dx = +-1 at columns 1 / W-2).  Those sums
// depend on the tap, so they cannot live in gy itself; they are pre-folded into two small side buffers that the loader lanes of
// igemm_ws2_kernel read INSTEAD of gy when their (tap, row / column group) is one of the special ones:
//   rows [v][d][n][k][xs]    v = 0: rows 2 + 0 (top), 1: rows H-3 + H-1 (bottom);  d = dx + 1 selects the column fold baked
//                            into that row: d = 2 adds texel 0 to texel 2, d = 0 adds texel W-1 to texel W-3
//   cols [e][n][k][y][j]     e = 0 (dx = +1): source columns 1..4 of row y with column 0 added to column 2;
//                            e = 1 (dx = -1): source columns W-5..W-2 with column W-1 added to column W-3
// (one 16-byte group = what the first / last 4-pixel output group of a row loads for that tap).
__global__ __launch_bounds__(256) void reflect_aux_kernel(const float* __restrict__ gy, float* __restrict__ rows,
                                                          float* __restrict__ cols, int N, int K, int H, int W) {
    const long long nrow = 6ll * N * K * W, ncol = 8ll * N * K * H;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < nrow + ncol;
         idx += (long long)gridDim.x * blockDim.x) {
        if (idx < nrow) {
            const int xs = (int)(idx % W);
            long long t = idx / W;
            const int k = (int)(t % K); t /= K;
            const int n = (int)(t % N);
            const int vd = (int)(t / N), v = vd / 3, d = vd - 3 * v;
            const float* g = gy + ((size_t)n * K + k) * H * W;
            const float* ra = g + (size_t)(v == 0 ? 2 : H - 3) * W;
            const float* rb = g + (size_t)(v == 0 ? 0 : H - 1) * W;
            float val = ra[xs] + rb[xs];
            if (d == 2 && xs == 2) val += ra[0] + rb[0];
            if (d == 0 && xs == W - 3) val += ra[W - 1] + rb[W - 1];
            rows[idx] = val;
        } else {
            const long long c = idx - nrow;
            const int j = (int)(c & 3);
            long long t = c >> 2;
            const int y = (int)(t % H); t /= H;
            const int k = (int)(t % K); t /= K;
            const int n = (int)(t % N);
            const int e = (int)(t / N);
            const float* r = gy + (((size_t)n * K + k) * H + y) * W;
            float val;
            if (e == 0) val = r[1 + j] + (j == 1 ? r[0] : 0.f);
            else val = r[W - 5 + j] + (j == 2 ? r[W - 1] : 0.f);
            cols[c] = val;
        }
    }
}

// Inverse of decode_ring: padded position (py, px) of the border ring -> index in the compact ring layout.
__device__ __forceinline__ unsigned encode_ring(unsigned rp, unsigned H, unsigned W, unsigned py, unsigned px) {
    const unsigned Wp = W + 2 * rp, band = rp * Wp;
    if (py < rp) return py * Wp + px;
    if (py >= H + rp) return band + (py - H - rp) * Wp + px;
    return 2 * band + (py - rp) * (2 * rp) + (px < rp ? px : px - W);
}

// gx[n,c,ty,tx] += sum of the ring texels that mirror onto (ty,tx), in a fixed order (gather: one thread per affected texel,
// no atomics).  Affected texels: rows 1..pad and H-1-pad..H-2 (whole rows), and columns 1..pad, W-1-pad..W-2 of every row.
// The launch enumerates, per (n,c) plane, `nrows` listed rows x W columns, then H rows x `ncols` listed columns (a texel of
// the second part that lies in a listed row was handled by the first part and is skipped).
struct RingBand { int nrows, ncols; int rows[8], cols[8]; };
__global__ __launch_bounds__(256) void ring_gather_kernel(const float* __restrict__ ring, float* __restrict__ gx, int H, int W,
                                                          int pad, int ring_len, RingBand band, long long planes) {
    const int per_plane = band.nrows * W + H * band.ncols;
    const long long total = planes * per_plane;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long nc = idx / per_plane;
        int e = (int)(idx - nc * per_plane), ty, tx;
        if (e < band.nrows * W) {
            ty = band.rows[e / W];
            tx = e % W;
        } else {
            e -= band.nrows * W;
            ty = e / band.ncols;
            tx = band.cols[e % band.ncols];
            bool listed = false;
            for (int i = 0; i < band.nrows; ++i) listed = listed || band.rows[i] == ty;
            if (listed) continue;
        }
        // padded rows / columns that mirror onto ty / tx (interior candidate first)
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = ty + pad;
        if (ty >= 1 && ty <= pad) ys[ny++] = pad - ty;
        if (ty <= H - 2 && ty >= H - 1 - pad) ys[ny++] = 2 * (H - 1) - ty + pad;
        xs[nx++] = tx + pad;
        if (tx >= 1 && tx <= pad) xs[nx++] = pad - tx;
        if (tx <= W - 2 && tx >= W - 1 - pad) xs[nx++] = 2 * (W - 1) - tx + pad;
        const float* r = ring + nc * ring_len;
        float sum = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b)
                if (a | b) sum += r[encode_ring((unsigned)pad, (unsigned)H, (unsigned)W, (unsigned)ys[a], (unsigned)xs[b])];
        gx[nc * (long long)H * W + (long long)ty * W + tx] += sum;
    }
}

// w2[c][k][r][s] = w[k][c][R-1-r][S-1-s]: the data gradient of a stride-1 convolution is the correlation of gy with
// these weights (and padding R-1-pad), which lets a layer with <= 4 INPUT channels use the narrow forward kernel
__global__ __launch_bounds__(256) void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ w2, int K,
                                                             int C, int R, int S) {
    const int total = K * C * R * S;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int s = idx % S, r = (idx / S) % R, k = (idx / (S * R)) % K, c = idx / (S * R * K);
        w2[idx] = w[(((size_t)k * C + c) * R + (R - 1 - r)) * S + (S - 1 - s)];
    }
}

// ... as a job of a weight-pack plan (pack_plan.h)
struct FlipTArgs {
    const float* w; float* w2;
    int K, C, R, S;
    int gx, gy;
};
__device__ __forceinline__ void flipt_body(const FlipTArgs& a, int bx, int, int gx) {
    const int total = a.K * a.C * a.R * a.S;
    for (int idx = bx * 256 + threadIdx.x; idx < total; idx += gx * 256) {
        const int s = idx % a.S, r = (idx / a.S) % a.R, k = (idx / (a.S * a.R)) % a.K, c = idx / (a.S * a.R * a.K);
        a.w2[idx] = a.w[(((size_t)k * a.C + c) * a.R + (a.R - 1 - r)) * a.S + (a.S - 1 - s)];
    }
}
NEMAR_PACK_MULTI(flipt_multi_kernel, FlipTArgs, flipt_body, 256)
void flipt_multi(const void* jobs, int njobs, int gx, int gy, hipStream_t st) {
    hipLaunchKernelGGL(flipt_multi_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, (const FlipTArgs*)jobs);
}
struct RegFlipT {
    RegFlipT() { nemar_pack_register(PACK_FAM_FLIPT, sizeof(FlipTArgs), flipt_multi); }
} g_reg_flipt;

// round-1 form (nemar_tune(14, 0)): one atomic per workgroup
__global__ __launch_bounds__(256) void bias_grad_atomic_kernel(const float* __restrict__ g, float* __restrict__ gb, int N, int C,
                                                               int HW, int chunk) {
    __shared__ float red[16];
    const int c = blockIdx.x, n = blockIdx.y;
    const int beg = blockIdx.z * chunk, end = min(HW, beg + chunk);
    const float* q = g + ((size_t)n * C + c) * HW;
    float acc = 0.f;
    for (int i = beg + threadIdx.x; i < end; i += blockDim.x) acc += q[i];
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(gb + c, t);
}

// part[(n * chunks + z) * C + c] = sum of g[n,c, chunk z of the plane]; grid (C, N, chunks): fixed tree per workgroup, one
// plain store; nemar_sum_partials adds the N * chunks slabs into gb in order (bitwise reproducible bias gradient)
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ g, float* __restrict__ part, int N, int C,
                                                        int HW, int chunk) {
    __shared__ float red[16];
    const int c = blockIdx.x, n = blockIdx.y;
    const int beg = blockIdx.z * chunk, end = min(HW, beg + chunk);
    const float* q = g + ((size_t)n * C + c) * HW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if ((((uintptr_t)(q + beg)) & 15) == 0) {
        // 16-byte loads, four independent sums, two loads in flight per thread (the one-dword-one-accumulator loop ran at 1.7 TB/s)
        const float4* q4 = reinterpret_cast<const float4*>(q + beg);
        const int n4 = (end - beg) >> 2;
        int i = threadIdx.x;
        for (; i + 256 < n4; i += 512) {
            const float4 u = q4[i], v = q4[i + 256];
            a0 += u.x + v.x; a1 += u.y + v.y; a2 += u.z + v.z; a3 += u.w + v.w;
        }
        if (i < n4) { const float4 u = q4[i]; a0 += u.x; a1 += u.y; a2 += u.z; a3 += u.w; }
        for (int k = beg + (n4 << 2) + threadIdx.x; k < end; k += 256) a0 += q[k];
    } else {
        for (int i = beg + threadIdx.x; i < end; i += blockDim.x) a0 += q[i];
    }
    const float t = block_sum((a0 + a1) + (a2 + a3), red);
    if (threadIdx.x == 0) part[((size_t)n * gridDim.z + blockIdx.z) * C + c] = t;
}

// gx[n,c,h,w] = sum of the padded-domain gradient gp over every padded position that mirrors onto (h,w)
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ gp, float* __restrict__ gx, int H,
                                                           int W, int pad, long long total) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(idx % W);
        const long long t = idx / W;
        const int h = (int)(t % H);
        const long long nc = t / H;
        const float* q = gp + nc * (long long)Hp * Wp;
        // padded rows that map to h: h+pad always; pad-h if 1<=h<=pad; 2(H-1)-h+pad if H-1-pad<=h<=H-2 (columns likewise).  No index
        // lists in private arrays (dynamically indexed ones live in scratch memory): up to three rows x three columns, spelled out
        const int y0 = h + pad, x0 = w + pad;
        const int y1 = (h >= 1 && h <= pad) ? pad - h : -1, y2 = (h <= H - 2 && h >= H - 1 - pad) ? 2 * (H - 1) - h + pad : -1;
        const int x1 = (w >= 1 && w <= pad) ? pad - w : -1, x2 = (w <= W - 2 && w >= W - 1 - pad) ? 2 * (W - 1) - w + pad : -1;
        auto rowsum = [&](int y) {
            const float* r = q + (long long)y * Wp;
            float v = r[x0];
            if (x1 >= 0) v += r[x1];
            if (x2 >= 0) v += r[x2];
            return v;
        };
        float s = rowsum(y0);
        if (y1 >= 0) s += rowsum(y1);
        if (y2 >= 0) s += rowsum(y2);
        gx[idx] = s;
    }
}

// Reduction split for tiny, deep problems on the generic kernels (the launch would otherwise be a handful of workgroups
// each walking the whole reduction at memory latency): enough splits for ~256 workgroups, >= 4 stages each.
int small_problem_split(int M, int P, int Kred) {
    const TileChoice t = igemm_tile(M, P, nemar_cdiv(Kred, BK));
    if (t.bm == 128) return 1;
    const long long tiles = (long long)nemar_cdiv(M, t.bm) * nemar_cdiv(P, t.bn);
    const int stages = nemar_cdiv(Kred, BK);
    if (tiles >= 128 || stages < 16) return 1;
    int ks = nemar_cdiv(256, (int)tiles);
    if (ks > stages / 4) ks = stages / 4;
    return ks < 1 ? 1 : ks;
}

void fwd_taps(TapTable& t, int R, int S, int pad) {
    t.n = R * S;
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s) {
            t.dy[r * S + s] = (short)(r - pad);
            t.dx[r * S + s] = (short)(s - pad);
            t.dyx[r * S + s] = ((r - pad) << 16) | ((s - pad) & 0xffff);
            t.wofs[r * S + s] = r * S + s;
        }
}

// taps of output-pixel parity class (ph, pw) of a stride-`stride` data gradient: r with (ph + pad - r) % stride == 0
void dgrad_taps(TapTable& t, int R, int S, int pad, int stride, int ph, int pw) {
    t.n = 0;
    for (int r = 0; r < R; ++r) {
        if ((ph + pad - r) % stride != 0) continue;
        for (int s = 0; s < S; ++s) {
            if ((pw + pad - s) % stride != 0) continue;
            t.dy[t.n] = (short)((ph + pad - r) / stride);
            t.dx[t.n] = (short)((pw + pad - s) / stride);
            t.dyx[t.n] = ((int)t.dy[t.n] << 16) | ((int)t.dx[t.n] & 0xffff);
            t.wofs[t.n] = r * S + s;
            t.n++;
        }
    }
}


// every split of a reduction must own at least one stage (its slab is summed unconditionally)
int normalize_ksplit(int Kred, int ksplit) {
    if (ksplit <= 1) return 1;
    const int nk_all = nemar_cdiv(Kred, BK);
    const int nk_per = nemar_cdiv(nk_all, ksplit);
    return nemar_cdiv(nk_all, nk_per);
}

static bool split16_worth_it(int N, int OH, int OW, int K, int C, int R, int S) {
    return (long long)N * OH * OW * K * C * R * S >= g_split16_min_mmac * 1000000ll;
}

// ---- routing to conv_s16g.hip (general layers on the 16-bit matrix pipe) ------------------------------------------------------
void s16g_set_class(S16gProblem& q, int c, const TapTable& t, int OHc, int OWc, int ooy, int oox) {
    q.ntaps[c] = t.n;
    for (int i = 0; i < t.n && i < S16G_MAX_TAPS; ++i) { q.dy[c][i] = t.dy[i]; q.dx[c][i] = t.dx[i]; q.wofs[c][i] = t.wofs[i]; }
    q.OH[c] = OHc; q.OW[c] = OWc; q.ooy[c] = ooy; q.oox[c] = oox;
}
bool s16g_worth_it(long long macs) { return g_s16g && macs >= g_s16g_min_mmac * 1000000ll; }

// forward: geometry only (pointers are filled by the operator); false = not this route
bool s16g_fwd_problem(S16gProblem& q, S16gPlan& pl, int N, int C0, int C1, int H, int W, int K, int R, int S, int stride, int pad,
                      int pad_mode, int act, float slope) {
    const int C = C0 + C1;
    if (R * S > S16G_MAX_TAPS || stride > 2 || stride < 1) return false;
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    if (OH <= 0 || OW <= 0 || !s16g_worth_it((long long)N * OH * OW * K * C * R * S)) return false;
    q = S16gProblem();
    q.C0 = C0; q.C1 = C1; q.Hs = H; q.Ws = W; q.N = N; q.M = K; q.M0 = K;
    q.act = act; q.slope = slope; q.border = pad_mode; q.sstride = stride;
    q.OHf = OH; q.OWf = OW; q.osy = 1; q.osx = 1; q.ncls = 1;
    TapTable t;
    fwd_taps(t, R, S, pad);
    s16g_set_class(q, 0, t, OH, OW, 0, 0);
    pl = nemar_s16g_plan(q);
    return pl.ok != 0;
}

// data gradient / transposed convolution (zero padding): one class per output parity
bool s16g_dgrad_problem(S16gProblem& q, S16gPlan& pl, int N, int C, int mskip, int H, int W, int K, int OH, int OW, int R, int S,
                        int stride, int pad, int act, float slope) {
    if (R * S > S16G_MAX_TAPS || stride > 2 || stride < 1) return false;
    if (!s16g_worth_it((long long)N * OH * OW * K * (C - mskip) * R * S)) return false;
    q = S16gProblem();
    q.C0 = K; q.C1 = 0; q.Hs = OH; q.Ws = OW; q.N = N; q.M = C - mskip;
    q.act = act; q.slope = slope; q.border = BORDER_ZERO; q.sstride = 1;
    q.OHf = H; q.OWf = W; q.osy = stride; q.osx = stride; q.ncls = 0;
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw) {
            TapTable t;
            dgrad_taps(t, R, S, pad, stride, ph, pw);
            const int OHc = (H - ph + stride - 1) / stride, OWc = (W - pw + stride - 1) / stride;
            if (t.n == 0 || OHc <= 0 || OWc <= 0 || t.n > (stride > 1 ? S16G_CLS_TAPS : S16G_MAX_TAPS)) return false;
            s16g_set_class(q, q.ncls++, t, OHc, OWc, ph, pw);
        }
    pl = nemar_s16g_plan(q);
    return pl.ok != 0;
}

// ---- 7x7 / pad-3 layers with <= 4 channels on the OUTPUT side (the translation net's RGB head; the data gradient of its stem) -------
// out[k][y][x] = sum_{c, dy, dx} w[k][c][dy][dx] src[c][y + dy][x + dx] with 3 rows would waste 29 of 32 MFMA rows.  Instead the rows of the
// GEMM are the (k, dx) PAIRS (4 x 8 = 32 pseudo-channels): P[(k, dx)][y][x'] = sum_{c, dy} w[k][c][dy][dx] src[c][y + dy][x'] is a SEVEN-TAP
// VERTICAL convolution with 32 output channels — the general 16-bit-pipe kernel (conv_s16g.hip) takes it as it is, over the halo columns
// x' as well — and out[k][y][x] = sum_dx P[(k, dx)][y][x + dx] is a horizontal shift-sum (k7_mf_sum_kernel: + bias, activation, and
// for a reflect-padded data gradient the fold of the padded domain).  Workspace: [re-arranged weights][their packed image][P].
__global__ __launch_bounds__(256) void k7_mf_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int Ks, int Cb, int dgrad) {
    // wt[(ks * 8 + dx)][cb][dy]: forward w[ks][cb][dy][dx] (w = [Ks][Cb][7][7]); data gradient w[cb][ks][6 - dy][6 - dx] (w = [Cb][Ks][7][7])
    const int total = 32 * Cb * 7;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int dy = i % 7, cb = (i / 7) % Cb, m = i / (7 * Cb), ks = m >> 3, dx = m & 7;
        float v = 0.f;
        if (ks < Ks && dx < 7)
            v = dgrad ? w[(((size_t)cb * Ks + ks) * 7 + (6 - dy)) * 7 + (6 - dx)] : w[(((size_t)ks * Cb + cb) * 7 + dy) * 7 + dx];
        wt[i] = v;
    }
}
struct K7MfWtArgs {
    const float* w; float* wt;
    int Ks, Cb, dgrad;
    int gx, gy;
};
__device__ __forceinline__ void k7_mf_wt_body(const K7MfWtArgs& a, int bx, int, int gx) {
    const int total = 32 * a.Cb * 7;
    for (int i = bx * 256 + threadIdx.x; i < total; i += gx * 256) {
        const int dy = i % 7, cb = (i / 7) % a.Cb, m = i / (7 * a.Cb), ks = m >> 3, dx = m & 7;
        float v = 0.f;
        if (ks < a.Ks && dx < 7)
            v = a.dgrad ? a.w[(((size_t)cb * a.Ks + ks) * 7 + (6 - dy)) * 7 + (6 - dx)] : a.w[(((size_t)ks * a.Cb + cb) * 7 + dy) * 7 + dx];
        a.wt[i] = v;
    }
}
NEMAR_PACK_MULTI(k7_mf_wt_multi_kernel, K7MfWtArgs, k7_mf_wt_body, 256)
void k7_mf_wt_multi(const void* jobs, int njobs, int gx, int gy, hipStream_t st) {
    hipLaunchKernelGGL(k7_mf_wt_multi_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, (const K7MfWtArgs*)jobs);
}
struct RegK7MfWt {
    RegK7MfWt() { nemar_pack_register(PACK_FAM_PRE, sizeof(K7MfWtArgs), k7_mf_wt_multi); }
} g_reg_k7_mf_wt;

// out[n][k][y][x] = act(bias[k] + sum over the P positions (Y, X) that belong to (y, x) of sum_dx P[n][k * 8 + dx][Y][X + dx]).
// fold == 0: (Y, X) = (y, x).  fold == 1 (reflect-padded data gradient, P on the padded domain): every padded position that mirrors onto
// (y, x): rows y + 3, 3 - y (1 <= y <= 3), 2 (H - 1) - y + 3 (H - 4 <= y <= H - 2), columns likewise — fixed order, no atomics.
__global__ __launch_bounds__(256) void k7_mf_sum_kernel(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ out,
                                                        int Ks, int H, int W, int PH, int PW, int fold, int act, float slope, long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int x = (int)(idx % W);
        long long t = idx / W;
        const int y = (int)(t % H);
        t /= H;
        const int k = (int)(t % Ks);
        const long long n = t / Ks;
        const float* Pk = P + ((size_t)n * 32 + (size_t)k * 8) * PH * PW;
        const size_t pplane = (size_t)PH * PW;
        auto at = [&](int Y, int X) {
            const float* r = Pk + (size_t)Y * PW + X;
            float s = 0.f;
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) s += r[(size_t)dx * pplane + dx];
            return s;
        };
        float v;
        if (!fold) {
            v = at(y, x);
        } else {
            const int y0 = y + 3, x0 = x + 3;
            const int y1 = (y >= 1 && y <= 3) ? 3 - y : -1, y2 = (y <= H - 2 && y >= H - 4) ? 2 * (H - 1) - y + 3 : -1;
            const int x1 = (x >= 1 && x <= 3) ? 3 - x : -1, x2 = (x <= W - 2 && x >= W - 4) ? 2 * (W - 1) - x + 3 : -1;
            auto row = [&](int Y) {
                float s = at(Y, x0);
                if (x1 >= 0) s += at(Y, x1);
                if (x2 >= 0) s += at(Y, x2);
                return s;
            };
            v = row(y0);
            if (y1 >= 0) v += row(y1);
            if (y2 >= 0) v += row(y2);
        }
        if (bias) v += bias[k];
        out[idx] = apply_act(v, act, slope);
    }
}

// geometry of the vertical convolution: P = [N][32][PH][PW]; src = [N][Cb][Hs][Ws] seen through a border of `spad` (3: forward / zero-padded
// gradient; 6, zero: reflect gradient on the padded domain)
struct K7MfPlan {
    S16gProblem q;
    S16gPlan pl;
    size_t wt_off, pack_off, p_off, total;      // floats
    int PH, PW;
    bool ok;
};
K7MfPlan k7_mf_plan(int N, int Cb, int Hs, int Ws, int spad, int border) {
    K7MfPlan m;
    m.ok = false;
    m.PH = Hs + 2 * spad - 6;
    m.PW = Ws + 2 * spad;
    S16gProblem& q = m.q;
    q = S16gProblem();
    q.C0 = Cb; q.C1 = 0; q.Hs = Hs; q.Ws = Ws; q.N = N; q.M = 32; q.M0 = 32;
    q.act = ACT_NONE; q.slope = 0.f; q.border = border; q.sstride = 1;
    q.OHf = m.PH; q.OWf = m.PW; q.osy = 1; q.osx = 1; q.ncls = 1;
    TapTable t;
    t.n = 7;
    for (int i = 0; i < 7; ++i) { t.dy[i] = (short)(i - spad); t.dx[i] = (short)(-spad); t.dyx[i] = ((i - spad) << 16) | ((-spad) & 0xffff); t.wofs[i] = i; }
    s16g_set_class(q, 0, t, m.PH, m.PW, 0, 0);
    m.pl = nemar_s16g_plan(q);
    if (!m.pl.ok) return m;
    m.wt_off = 0;
    m.pack_off = ((size_t)32 * Cb * 7 + 3) & ~(size_t)3;
    m.p_off = (m.pack_off + (nemar_s16g_pack_bytes(q, m.pl) + 3) / 4 + 3) & ~(size_t)3;
    m.total = m.p_off + (size_t)N * 32 * m.PH * m.PW;
    m.ok = true;
    return m;
}
bool k7_mf_eligible(int Cb, int Ks, int R, int S, int stride, int pad) {
    return g_k7 && R == 7 && S == 7 && stride == 1 && pad == 3 && Ks >= 1 && Ks <= 4 && Cb >= 16 && Cb % 16 == 0;
}
// src -> out through the three launches (weights re-arranged + packed first unless prepacked)
void k7_mf_run(const K7MfPlan& m_, const float* src, const float* w, int Ks, int Cb, int dgrad, const float* bias, float* out, int N, int H, int W,
               int fold, int act, float slope, float* wsf, int prepacked, hipStream_t st) {
    K7MfPlan m = m_;
    float* wt = wsf + m.wt_off;
    void* packed = wsf + m.pack_off;
    float* P = wsf + m.p_off;
    if (!prepacked) {
        K7MfWtArgs a{w, wt, Ks, Cb, dgrad, nemar_stream_grid(32 * Cb * 7, 256), 1};
        if (nemar_pack_recording()) nemar_pack_record_job(PACK_FAM_PRE, &a, a.gx, 1);
        hipLaunchKernelGGL(k7_mf_weights_kernel, dim3(a.gx), dim3(256), 0, st, w, wt, Ks, Cb, dgrad);
        nemar_s16g_pack(m.q, m.pl, wt, (long long)Cb * 7, 7, packed, st);
    }
    m.q.src0 = src; m.q.src1 = nullptr; m.q.dst0 = P; m.q.dst1 = nullptr; m.q.bias = nullptr; m.q.dbg = 0; m.q.tl = nullptr;
    nemar_s16g_conv(m.q, m.pl, packed, st);
    const long long total = (long long)N * Ks * H * W;
    hipLaunchKernelGGL(k7_mf_sum_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, (const float*)P, bias, out, Ks, H, W, m.PH, m.PW, fold,
                       act, slope, total);
}

// Workspace layout of nemar_conv2d_bwd_data (floats), shared by the size query and the operator:
//   [packed weights x stride^2 parity classes][padded-domain scratch (strided reflect)][flipped weights (C <= 4)]
//   [compact border-ring gradient (stride-1 reflect)][ksplit slabs of the gradient (split reductions)]
struct DgradLayout {
    size_t pack_stride, padded_off, w2_off, ring_off, ring_slab_off, slab_off, aux_rows_off, aux_cols_off, total;
    int ring_len, ksplit, ring_ksplit;
    bool ring, fold, fold16;
};
DgradLayout dgrad_layout(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int pad_mode) {
    DgradLayout L;
    const bool refl = pad_mode == BORDER_REFLECT && pad > 0;
    L.ring = refl && stride == 1;
    L.fold = refl && !L.ring;
    L.pack_stride = packed_floats(C, K * R * S);             // upper bound over parity classes and channel skips
    if (nemar_split16_eligible(N, H, W, C, K, R, S, stride, pad, SPLIT16_ZERO, 4)) {     // room for either packed image
        const size_t b = (nemar_split16_pack_bytes(C, K, R) + 3) / 4;
        if (b > L.pack_stride) L.pack_stride = b;
    }
    {
        S16gProblem q;
        S16gPlan pl;
        const int OHd = (H + 2 * pad - R) / stride + 1, OWd = (W + 2 * pad - S) / stride + 1;
        if (!refl && OHd > 0 && OWd > 0 && s16g_dgrad_problem(q, pl, N, C, 0, H, W, K, OHd, OWd, R, S, stride, pad, ACT_NONE, 0.f)) {
            const size_t b = ((nemar_s16g_pack_bytes(q, pl) + 3) / 4 + stride * stride - 1) / (stride * stride);
            if (b > L.pack_stride) L.pack_stride = b;
        }
    }
    // stride-1 reflect layers the general 16-bit-pipe kernel takes: the data gradient of the PADDED input (a plain zero-padded full
    // correlation on the (H + 2p) x (W + 2p) domain) into scratch, then the fold — one extra pass over the gradient, but the
    // implicit GEMM runs at several times the exact-fp32 rate (R net: 175 -> ~95 us per call)
    L.fold16 = false;
    if (L.ring && g_s16g_fold && R * S <= 9) {          // (49-tap layers: the 32-row tile would cost more than it saves)
        S16gProblem q;
        S16gPlan pl;
        const int OHd = H + 2 * pad - R + 1, OWd = W + 2 * pad - S + 1;
        if (OHd > 0 && OWd > 0 && s16g_dgrad_problem(q, pl, N, C, 0, H + 2 * pad, W + 2 * pad, K, OHd, OWd, R, S, 1, 0, ACT_NONE, 0.f)) {
            L.fold16 = true;
            const size_t b = (nemar_s16g_pack_bytes(q, pl) + 3) / 4;
            if (b > L.pack_stride) L.pack_stride = b;
        }
    }
    size_t o = L.pack_stride * (size_t)(stride * stride);
    L.padded_off = o;
    if (L.fold || L.fold16) o += (size_t)N * C * (H + 2 * pad) * (W + 2 * pad);
    L.w2_off = o;
    if (C <= 4) o += (size_t)C * K * R * S;
    L.ring_off = o;
    L.ring_len = L.ring ? 2 * pad * (W + 2 * pad) + 2 * pad * H : 0;
    o += (size_t)N * C * L.ring_len;
    // a ring tile is a few pixels deep in a full-length reduction, and a lone workgroup per CU walks it at memory latency:
    // the reduction is split until ~1.5 workgroups per CU exist (each split = one slab, summed in order)
    L.ring_ksplit = 1;
    L.ring_slab_off = o;
    if (L.ring) {
        const int tiles = nemar_cdiv(N * L.ring_len, 64) * nemar_cdiv(C, 64), stages = nemar_cdiv(K * R * S, BK);
        int ks = nemar_cdiv(384, tiles);
        if (ks > nemar_cdiv(stages, 8)) ks = nemar_cdiv(stages, 8);
        L.ring_ksplit = normalize_ksplit(K * R * S, ks < 1 ? 1 : ks);
        if (L.ring_ksplit > 1) o += (size_t)L.ring_ksplit * N * C * L.ring_len;
    }
    // split reductions (stride 1, single destination, no bias / activation — the operator re-checks those): few, deep
    // 128x128 tiles (D's 256->512 k4 layer: 128 tiles x 512 stages) get one workgroup per CU; tiny deep problems on the
    // generic kernels (the 2x2 .. 32x32-pixel layers of the registration net) ~256 workgroups of >= 4 stages
    L.ksplit = 1;
    if (g_ksplit && stride == 1 && !L.fold && C > 4) {
        const int P = N * H * W, Kred = K * R * S, stages = nemar_cdiv(Kred, BK);
        if (g_cfg128 == 0 && C > 64 && K % BK == 0) {
            const long long tiles = (long long)nemar_cdiv(C, 128) * nemar_cdiv(P, 128);
            if (tiles < 200 && stages >= 256) {
                int ks = nemar_cdiv(256, (int)tiles);
                if (ks > stages / 128) ks = stages / 128;
                if (ks > 1) L.ksplit = ks;
            }
        }
        if (L.ksplit == 1) L.ksplit = small_problem_split(C, P, Kred);
        L.ksplit = normalize_ksplit(Kred, L.ksplit);
    }
    L.slab_off = o;
    if (L.ksplit > 1) o += (size_t)L.ksplit * N * C * H * W;
    // side buffers of the ring-free reflect data gradient (source = gy [N,K,H,W] for a 3x3 / pad 1 layer)
    L.aux_rows_off = o;
    if (L.ring && pad == 1 && R == 3 && S == 3) o += 6ull * N * K * W;
    L.aux_cols_off = o;
    if (L.ring && pad == 1 && R == 3 && S == 3) o += 8ull * N * K * H;
    L.total = o;
    if (k7_mf_eligible(K, C, R, S, stride, pad)) {                       // 7x7 stem (<= 4 input channels): [re-arranged weights | packed image | P]
        const K7MfPlan m = k7_mf_plan(N, K, H, W, refl ? 6 : 3, BORDER_ZERO);
        if (m.ok && m.total > L.total) L.total = m.total;
    }
    if (C > 4 && nemar_k7_fm_eligible(K, C, R, S, stride, pad)) {       // 7x7 head (<= 4 output channels): [packed weights | padded-domain gradient]
        const size_t k7 = ((nemar_k7_fm_pack_floats(C) + 3) & ~(size_t)3) + (refl ? (size_t)N * C * (H + 6) * (W + 6) : 0);
        if (k7 > L.total) L.total = k7;
    }
    return L;
}


// Workspace of nemar_conv2d_fwd (floats): [packed weights][slabs of a split reduction].  Tiny, deep layers (the 2x2 .. 32x32
// maps of the registration net: a 2x2-pixel 128->128 3x3 layer is 72 serial stages in two workgroups) split their reduction
// like the data gradients do; per-split slabs summed in order keep the forward pass bitwise reproducible.
struct FwdLayout { size_t pack, slab_off, total; int ksplit; };
FwdLayout fwd_layout(int N, int H, int W, int K, int C, int R, int S, int stride, int pad) {
    FwdLayout L;
    L.pack = packed_floats(K, C * R * S);
    if (nemar_split16_eligible(N, H, W, K, C, R, S, stride, pad, SPLIT16_ZERO, 4)) {     // room for either packed image
        const size_t b = (nemar_split16_pack_bytes(K, C, R) + 3) / 4;
        if (b > L.pack) L.pack = b;
    }
    {
        S16gProblem q;
        S16gPlan pl;
        if (s16g_fwd_problem(q, pl, N, C, 0, H, W, K, R, S, stride, pad, BORDER_ZERO, ACT_NONE, 0.f)) {
            const size_t b = (nemar_s16g_pack_bytes(q, pl) + 3) / 4;
            if (b > L.pack) L.pack = b;
        }
    }
    if (nemar_k7_fm_eligible(C, K, R, S, stride, pad) && nemar_k7_fm_pack_floats(K) > L.pack) L.pack = nemar_k7_fm_pack_floats(K);
    L.ksplit = 1;
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    if (g_ksplit && OH > 0 && OW > 0 && K > 4) L.ksplit = normalize_ksplit(C * R * S, small_problem_split(K, N * OH * OW, C * R * S));
    L.slab_off = (L.pack + 3) & ~(size_t)3;
    L.total = L.slab_off + (L.ksplit > 1 ? (size_t)L.ksplit * N * K * OH * OW : 0);
    if (k7_mf_eligible(C, K, R, S, stride, pad)) {           // 7x7 head: [re-arranged weights | packed image | P] (either border)
        const K7MfPlan m = k7_mf_plan(N, C, H, W, 3, BORDER_ZERO);
        if (m.ok && m.total > L.total) L.total = m.total;
    }
    return L;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
NEMAR_API size_t nemar_conv2d_fwd_workspace(int N, int H, int W, int K, int C, int R, int S, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || K <= 0 || C <= 0 || R <= 0 || S <= 0 || stride < 1) return 0;
    return sizeof(float) * fwd_layout(N, H, W, K, C, R, S, stride, pad).total;
}

NEMAR_API int nemar_conv2d_fwd(const float* x0, int C0, const float* x1, int C1, const float* w, const float* bias,
                               float* y, int N, int H, int W, int K, int R, int S, int stride, int pad, int pad_mode,
                               int act, float slope, void* workspace, size_t ws_bytes, int prepacked, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x0 && w && y && workspace, "conv2d_fwd: null pointer");
    NEMAR_REQUIRE(C0 > 0 && C1 >= 0 && (C1 == 0 || x1), "conv2d_fwd: bad channel split %d+%d", C0, C1);
    NEMAR_REQUIRE(N > 0 && H > 0 && W > 0 && K > 0 && R > 0 && S > 0 && R * S <= MAX_TAPS, "conv2d_fwd: bad shape");
    NEMAR_REQUIRE(stride >= 1 && pad >= 0 && pad < 32768, "conv2d_fwd: bad stride/pad");
    NEMAR_REQUIRE(pad_mode == BORDER_ZERO || (pad_mode == BORDER_REFLECT && pad < H && pad < W),
                  "conv2d_fwd: reflect pad %d needs pad < H,W (%d,%d)", pad, H, W);
    const int C = C0 + C1;
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    NEMAR_REQUIRE(OH > 0 && OW > 0, "conv2d_fwd: empty output");
    NEMAR_REQUIRE((long long)N * OH * OW < (1ll << 31) && (long long)C * H * W < (1ll << 31) &&
                      (long long)C * R * S < (1 << 20),
                  "conv2d_fwd: problem too large for 32-bit tile indexing");
    const FwdLayout FL = fwd_layout(N, H, W, K, C, R, S, stride, pad);
    const size_t need = sizeof(float) * FL.total;
    if (ws_bytes < need) {
        nemar_set_error("conv2d_fwd: workspace %zu < %zu", ws_bytes, need);
        return NEMAR_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (C1 == 0 && k7_mf_eligible(C, K, R, S, stride, pad)) {
        // 7x7 head (<= 4 output channels): vertical 7-tap convolution with (k, dx) pseudo-channels on the general 16-bit-pipe kernel, then
        // the horizontal shift-sum with bias and activation (tanh included)
        const K7MfPlan m = k7_mf_plan(N, C, H, W, 3, pad_mode);
        if (m.ok) {
            k7_mf_run(m, x0, w, K, C, 0, bias, y, N, H, W, 0, act, slope, (float*)workspace, prepacked, st);
            g_last_route = 4;
            NEMAR_CHECK_LAUNCH("conv2d_fwd (7x7 many -> few, 16-bit pipe)");
            return NEMAR_OK;
        }
    }
    if (g_k7 && C1 == 0 && act != ACT_TANH && nemar_k7_fm_eligible(C, K, R, S, stride, pad)) {
        // 7x7 stem (<= 4 input channels): row-expanded source on the 16-bit matrix pipe, weights in registers (conv_k7.hip)
        if (!prepacked) nemar_k7_fm_pack(w, (long long)C * 49, 49, 0, K, C, workspace, st);
        nemar_k7_fm_conv(x0, C, H, W, 3, pad_mode == BORDER_REFLECT, workspace, bias, y, K, N, H, W, act, slope, 0, st);
        g_last_route = 4;
        NEMAR_CHECK_LAUNCH("conv2d_fwd (7x7, 16-bit pipe)");
        return NEMAR_OK;
    }
    if (nemar_narrow_eligible(K, C1, R, S, stride, N, OH, OW) && g_narrow) {
        // the narrow kernels read the weights in place: the packed-weight workspace doubles as the slab space of their
        // channel-split mode (the split count is capped to what fits, see nemar_narrow_fwd)
        nemar_narrow_fwd(x0, w, bias, y, N, C, H, W, K, R, pad, pad_mode, act, slope,
                         g_deterministic ? (float*)workspace : nullptr, ws_bytes / sizeof(float), st);
        g_last_route = 1;
        NEMAR_CHECK_LAUNCH("conv2d_fwd (narrow)");
        return NEMAR_OK;
    }
    {
        const int mode = pad_mode == BORDER_REFLECT ? SPLIT16_REFLECT : SPLIT16_ZERO;
        if (g_split16 && C1 == 0 && act == ACT_NONE && split16_worth_it(N, OH, OW, K, C, R, S) &&
            nemar_split16_eligible(N, H, W, K, C, R, S, stride, pad, mode, g_split16_variant) &&
            g_scratch && g_scratch_bytes >= nemar_split16_scratch_total(N, H, W, K, C, OH, OW)) {
            if (!prepacked) nemar_split16_pack(w, workspace, K, C, R, 0, g_split16_variant, st);
            nemar_split16_conv(x0, workspace, bias, y, N, H, W, K, C, R, 1, H, W, OH, OW, mode, g_scratch, g_xcd_map, g_split16_variant,
                               g_tl, nullptr, st);
            g_last_route = 2;
            NEMAR_CHECK_LAUNCH("conv2d_fwd (split-16)");
            return NEMAR_OK;
        }
    }
    {
        S16gProblem q;
        S16gPlan pl;
        if (s16g_fwd_problem(q, pl, N, C0, C1, H, W, K, R, S, stride, pad, pad_mode, act, slope)) {
            q.src0 = x0; q.src1 = x1; q.dst0 = y; q.dst1 = nullptr; q.bias = bias; q.dbg = g_dbg; q.tl = g_tl;
            if (!prepacked) nemar_s16g_pack(q, pl, w, (long long)C * R * S, (long long)R * S, workspace, st);
            nemar_s16g_conv(q, pl, workspace, st);
            g_last_route = 3;
            NEMAR_CHECK_LAUNCH("conv2d_fwd (16-bit pipe, in-kernel split)");
            return NEMAR_OK;
        }
    }
    IgemmParams p;
    fwd_taps(p.taps, R, S, pad);
    if (!prepacked) launch_pack(w, (float*)workspace, K, C, C * R * S, R * S, p.taps, st);
    p.src0 = x0; p.src1 = x1; p.C0 = C0; p.C1 = C1; p.Hs = H; p.Ws = W;
    p.wp = (const float*)workspace; p.M = K; p.Mpad = igemm_mpad(K); p.Kred = C * R * S;
    p.zero = p.wp + packed_core_floats(K, C * R * S);
    p.dbg = g_dbg; p.tl = g_tl;
    p.ring_p = 0; p.ring_H = 0; p.ring_W = 0; p.ksplit = 1; p.part = nullptr; p.part_stride = 0;
    p.rf = 0; p.rf_row = nullptr; p.rf_col = nullptr; p.xcd = 0;
    p.bias = bias;
    p.dst0 = y; p.dst1 = nullptr; p.M0 = K;
    p.OH = OH; p.OW = OW; p.OHf = OH; p.OWf = OW; p.osy = 1; p.ooy = 0; p.osx = 1; p.oox = 0;
    p.N = N; p.P = N * OH * OW;
    p.sy = stride; p.sx = stride; p.border = pad_mode; p.act = act; p.slope = slope; p.pad = pad;
    p.fd_ohw = make_fastdiv(OH * OW); p.fd_ow = make_fastdiv(OW); p.fd_cs = make_fastdiv(C);
    // (a split with the activation applied by the sum pass — nemar_sum_partials_act — is wired but off: it changes the summation order of
    // the registration net's decoder layers, and the reduced-width parity test then sits on a different LeakyReLU knife edge of D
    // (tests/step_parity.py); 0.2 ms per step were not worth re-measuring every allowance.  nemar_tune(36, 1) switches it on.)
    const bool split_act = g_split_act && (act == ACT_RELU || act == ACT_LRELU);
    if (FL.ksplit > 1 && (act == ACT_NONE || split_act)) {      // slab 0 carries the bias; an activation follows the sum (ReLU / LeakyReLU)
        p.ksplit = FL.ksplit;
        p.part = (float*)workspace + FL.slab_off;
        p.part_stride = (long long)N * K * OH * OW;
        p.act = ACT_NONE;
    }
    launch_igemm(p, st);
    if (p.ksplit > 1) {
        if (split_act) nemar_sum_partials_act(p.part, p.part_stride, p.ksplit, y, p.part_stride, act == ACT_RELU ? 1 : 2, slope, st);
        else nemar_sum_partials(p.part, p.part_stride, p.ksplit, y, p.part_stride, false, st);
    }
    g_last_route = 0;
    NEMAR_CHECK_LAUNCH("conv2d_fwd");
    return NEMAR_OK;
}

// Data gradient of the conv above: gy [N,K,OH,OW] -> gx [N,C,H,W] (split over gx0[C0] | gx1[C1]; gx0 may be NULL to
// skip its channels).  With bias/act it is also the FORWARD of nn.ConvTranspose2d(K -> C) whose weight is w[K][C][R][S].
NEMAR_API size_t nemar_conv2d_bwd_data_workspace(int N, int C, int H, int W, int K, int R, int S, int stride, int pad,
                                                 int pad_mode) {
    if (N <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride < 1) return 0;
    return sizeof(float) * dgrad_layout(N, C, H, W, K, R, S, stride, pad, pad_mode).total;
}

NEMAR_API int nemar_conv2d_bwd_data(const float* gy, const float* w, const float* bias, int act, float slope,
                                    float* gx0, int C0, float* gx1, int C1, int N, int H, int W, int K, int OH, int OW,
                                    int R, int S, int stride, int pad, int pad_mode, void* workspace, size_t ws_bytes,
                                    int prepacked, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(gy && w && workspace && (gx0 || gx1), "conv2d_bwd_data: null pointer");
    NEMAR_REQUIRE(C0 >= 0 && C1 >= 0 && C0 + C1 > 0 && (C1 == 0 || gx1), "conv2d_bwd_data: bad channel split");
    NEMAR_REQUIRE(N > 0 && H > 0 && W > 0 && K > 0 && OH > 0 && OW > 0 && R * S <= MAX_TAPS && R > 0 && S > 0,
                  "conv2d_bwd_data: bad shape");
    NEMAR_REQUIRE(stride >= 1 && stride <= 4 && pad >= 0, "conv2d_bwd_data: bad stride/pad");
    NEMAR_REQUIRE((H + 2 * pad - R) / stride + 1 == OH && (W + 2 * pad - S) / stride + 1 == OW,
                  "conv2d_bwd_data: gy %dx%d inconsistent with x %dx%d k%d s%d p%d", OH, OW, H, W, R, stride, pad);
    const int C = C0 + C1;
    const bool refl = pad_mode == BORDER_REFLECT && pad > 0;
    NEMAR_REQUIRE(pad_mode != BORDER_REFLECT || (pad < H && pad < W), "conv2d_bwd_data: reflect pad too large");
    NEMAR_REQUIRE(!refl || (!bias && act == ACT_NONE && gx1 == nullptr),
                  "conv2d_bwd_data: reflect mode supports a single destination without bias/activation");
    NEMAR_REQUIRE(!refl || stride > 1 || pad <= 4, "conv2d_bwd_data: reflect pad %d > 4 unsupported", pad);
    const DgradLayout L = dgrad_layout(N, C, H, W, K, R, S, stride, pad, pad_mode);
    if (ws_bytes < sizeof(float) * L.total) {
        nemar_set_error("conv2d_bwd_data: workspace %zu < %zu", ws_bytes, sizeof(float) * L.total);
        return NEMAR_EWORKSPACE;
    }
    NEMAR_REQUIRE((long long)N * (H + 2 * pad) * (W + 2 * pad) < (1ll << 31) && (long long)K * OH * OW < (1ll << 31) &&
                      (long long)K * R * S < (1 << 20),
                  "conv2d_bwd_data: problem too large for 32-bit tile indexing");
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)workspace;
    if (C1 == 0 && gx0 && !bias && act == ACT_NONE && k7_mf_eligible(K, C, R, S, stride, pad)) {
        // 7x7 stem (<= 4 input channels): gx = the many -> few correlation of gy with flipped, transposed weights; reflect border: on the
        // padded domain (gy through a 6-texel zero border), folded back inside the shift-sum pass
        const K7MfPlan m = k7_mf_plan(N, K, OH, OW, refl ? 6 : 3, BORDER_ZERO);
        if (m.ok) {
            k7_mf_run(m, gy, w, C, K, 1, nullptr, gx0, N, H, W, refl ? 1 : 0, ACT_NONE, 0.f, wsf, prepacked, st);
            g_last_route = 4;
            NEMAR_CHECK_LAUNCH("conv2d_bwd_data (7x7 many -> few, 16-bit pipe)");
            return NEMAR_OK;
        }
    }
    if (g_k7 && C1 == 0 && gx0 && !bias && act == ACT_NONE && C > 4 && nemar_k7_fm_eligible(K, C, R, S, stride, pad)) {
        // 7x7 head (<= 4 output channels): the data gradient is a few -> many convolution of gy with flipped, transposed weights
        // (conv_k7.hip).  Reflect border: on the padded (H + 6) x (W + 6) domain (gy through a 6-texel zero border), then the fold.
        if (!prepacked) nemar_k7_fm_pack(w, 49, (long long)C * 49, 1, C, K, workspace, st);
        if (refl && nemar_k7_fm_fold_ok(H, W)) {             // mirrored contributions accumulated in the kernel: no padded tensor, no fold pass
            nemar_k7_fm_conv(gy, K, OH, OW, 6, 0, workspace, nullptr, gx0, C, N, H, W, ACT_NONE, 0.f, 1, st);
        } else if (refl) {
            float* padded = wsf + ((nemar_k7_fm_pack_floats(C) + 3) & ~(size_t)3);
            nemar_k7_fm_conv(gy, K, OH, OW, 6, 0, workspace, nullptr, padded, C, N, H + 6, W + 6, ACT_NONE, 0.f, 0, st);
            const long long total = (long long)N * C * H * W;
            hipLaunchKernelGGL(reflect_fold_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, (const float*)padded, gx0, H, W, pad,
                               total);
        } else {
            nemar_k7_fm_conv(gy, K, OH, OW, 3, 0, workspace, nullptr, gx0, C, N, H, W, ACT_NONE, 0.f, 0, st);
        }
        g_last_route = 4;
        NEMAR_CHECK_LAUNCH("conv2d_bwd_data (7x7, 16-bit pipe)");
        return NEMAR_OK;
    }
    {
        // 3x3 stride-1 layers: the data gradient is the same convolution with flipped, transposed weights (conv_split16.hip)
        const int mode = refl ? SPLIT16_DGRAD_REFLECT : SPLIT16_ZERO;
        if (g_split16 && C1 == 0 && gx0 && !bias && act == ACT_NONE && split16_worth_it(N, OH, OW, K, C, R, S) &&
            nemar_split16_eligible(N, H, W, C, K, R, S, stride, pad, mode, g_split16_variant) &&
            g_scratch && g_scratch_bytes >= nemar_split16_scratch_total(N, H, W, C, K, H, W)) {
            if (!prepacked) nemar_split16_pack(w, workspace, K, C, R, 1, g_split16_variant, st);
            void* dual = nullptr;      // the weight gradient of the same layer follows and takes its gy planes from this call's split pass
            if (g_dual_gy && R == 3 && t_gy_planes_out && t_gy_planes_bytes >= nemar_split16_wgrad_g_bytes(N, H, W, K, R) &&
                nemar_split16_wgrad_g_bytes(N, H, W, K, R) > 0 && nemar_split16_wgrad_eligible(N, C, H, W, K, R, S, stride, pad))
                dual = t_gy_planes_out;
            // (whether the planes were written is the split pass's own decision: variant, producer planes, g_dual_gy — ask it)
            t_gy_planes_written = nemar_split16_conv(gy, workspace, nullptr, gx0, N, H, W, C, K, R, R - 1 - pad, OH, OW, H, W, mode, g_scratch,
                                                     g_xcd_map, g_split16_variant, g_tl, dual, st) ? 1 : 0;
            g_last_route = 2;
            NEMAR_CHECK_LAUNCH("conv2d_bwd_data (split-16)");
            return NEMAR_OK;
        }
    }
    if (!refl) {
        const int mskip0 = (gx0 == nullptr) ? C0 : 0;
        S16gProblem q;
        S16gPlan pl;
        if (s16g_dgrad_problem(q, pl, N, C, mskip0, H, W, K, OH, OW, R, S, stride, pad, act, slope)) {
            q.src0 = gy; q.src1 = nullptr; q.bias = bias ? bias + mskip0 : nullptr;
            if (mskip0) { q.dst0 = gx1; q.dst1 = nullptr; q.M0 = q.M; }
            else { q.dst0 = gx0; q.dst1 = gx1; q.M0 = C0; }
            // output row m = input channel m + mskip0, reduction channel = k:  w[k][c][r][s]
            if (!prepacked) nemar_s16g_pack(q, pl, w + (size_t)mskip0 * R * S, (long long)R * S, (long long)C * R * S, workspace, st);
            nemar_s16g_conv(q, pl, workspace, st);
            g_last_route = 3;
            NEMAR_CHECK_LAUNCH("conv2d_bwd_data (16-bit pipe, in-kernel split)");
            return NEMAR_OK;
        }
    }
    if (refl && L.fold16 && gx1 == nullptr && gx0 && !bias && act == ACT_NONE) {
        S16gProblem q;
        S16gPlan pl;
        const int Hp = H + 2 * pad, Wp = W + 2 * pad;
        if (s16g_dgrad_problem(q, pl, N, C, 0, Hp, Wp, K, OH, OW, R, S, 1, 0, ACT_NONE, 0.f)) {
            float* const padded16 = wsf + L.padded_off;
            q.src0 = gy; q.src1 = nullptr; q.bias = nullptr; q.dst0 = padded16; q.dst1 = nullptr; q.M0 = q.M;
            if (!prepacked) nemar_s16g_pack(q, pl, w, (long long)R * S, (long long)C * R * S, workspace, st);
            nemar_s16g_conv(q, pl, workspace, st);
            const long long total = (long long)N * C * H * W;
            hipLaunchKernelGGL(reflect_fold_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, (const float*)padded16, gx0, H,
                               W, pad, total);
            g_last_route = 3;
            NEMAR_CHECK_LAUNCH("conv2d_bwd_data (16-bit pipe on the padded domain + fold)");
            return NEMAR_OK;
        }
    }
    const size_t pack_stride = L.pack_stride;
    // Reflect padding.  The gradient w.r.t. the PADDED input splits into the image interior — exactly the zero-padded
    // data gradient, computed on the unpadded domain — and the border ring, whose texels are mirrors of in-image
    // texels: a second, small launch evaluates the same implicit GEMM at the ring positions only into a compact scratch,
    // and ring_gather_kernel adds each ring texel to the texel it mirrors (stride 1; gather form, no atomics).  Strided
    // reflect convolutions (not on the hot path) keep the simple form: differentiate on the padded domain into
    // scratch, then fold.
    const bool ring = L.ring, fold = L.fold;
    const int Hd = fold ? H + 2 * pad : H, Wd = fold ? W + 2 * pad : W;
    const int padd = fold ? 0 : pad;
    float* padded = fold ? wsf + L.padded_off : nullptr;
    // skipping the first C0 channels when gx0 == NULL: start the M range at C0
    const int mskip = (gx0 == nullptr) ? C0 : 0;
    int cls = 0;
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw, ++cls) {
            IgemmParams p;
            dgrad_taps(p.taps, R, S, padd, stride, ph, pw);
            const int OHc = (Hd - ph + stride - 1) / stride, OWc = (Wd - pw + stride - 1) / stride;
            if (OHc <= 0 || OWc <= 0) continue;
            const int Mc = C - mskip;
            p.src0 = gy; p.src1 = nullptr; p.C0 = K; p.C1 = 0; p.Hs = OH; p.Ws = OW;
            p.M = Mc; p.Mpad = igemm_mpad(Mc); p.Kred = p.taps.n * K;
            p.bias = bias ? bias + mskip : nullptr;
            if (fold) { p.dst0 = padded; p.dst1 = nullptr; p.M0 = Mc; }
            else if (mskip) { p.dst0 = gx1; p.dst1 = nullptr; p.M0 = Mc; }
            else { p.dst0 = gx0; p.dst1 = gx1; p.M0 = C0; }
            p.OH = OHc; p.OW = OWc; p.OHf = Hd; p.OWf = Wd; p.osy = stride; p.ooy = ph; p.osx = stride; p.oox = pw;
            p.N = N; p.P = N * OHc * OWc;
            p.sy = 1; p.sx = 1; p.border = BORDER_ZERO; p.act = act; p.slope = slope;
            p.pad = pad;
            p.ring_p = 0; p.ring_H = 0; p.ring_W = 0; p.ksplit = 1; p.part = nullptr; p.part_stride = 0;
            p.rf = 0; p.rf_row = nullptr; p.rf_col = nullptr; p.xcd = 0;
            p.fd_ohw = make_fastdiv(OHc * OWc); p.fd_ow = make_fastdiv(OWc); p.fd_cs = make_fastdiv(K);
            float* wp = wsf + pack_stride * (size_t)cls;
            p.wp = wp;
            p.zero = wp + packed_core_floats(Mc, K * (p.taps.n > 0 ? p.taps.n : 1));
            p.dbg = g_dbg; p.tl = nullptr;
            if (p.taps.n == 0) {
                // no tap reaches this class (e.g. k1 s2): gradient is bias-only / zero; run with one zero tap
                p.taps.n = 1; p.taps.dy[0] = -32000; p.taps.dx[0] = -32000; p.taps.wofs[0] = 0; p.Kred = K;
                p.taps.dyx[0] = (int)(((unsigned)-32000 << 16) | ((unsigned)-32000 & 0xffffu));
            }
            // A[(t*K + k)][c] = w[k][c + mskip][r][s]
            if (!prepacked) launch_pack(w + (size_t)mskip * R * S, wp, Mc, K, R * S, C * R * S, p.taps, st);
            // <= 4 input channels (the translation net's stem: 29 of 32 MFMA rows would be empty): the zero-padded
            // data gradient is a <= 4-output-channel correlation of gy — the narrow VALU kernel's job
            const bool narrow = g_narrow && stride == 1 && !fold && !bias && act == ACT_NONE && mskip == 0 && gx1 == nullptr &&
                                R - 1 - pad >= 0 && nemar_narrow_eligible(C, 0, R, S, 1, N, H, W);
            bool ring_done = false;
            if (narrow) {
                float* w2 = wsf + L.w2_off;
                if (!prepacked) {
                    if (nemar_pack_recording()) {
                        FlipTArgs a{w, w2, K, C, R, S, nemar_stream_grid((long long)K * C * R * S, 256), 1};
                        nemar_pack_record_job(PACK_FAM_FLIPT, &a, a.gx, 1);
                    }
                    hipLaunchKernelGGL(flip_transpose_kernel, dim3(nemar_stream_grid((long long)K * C * R * S, 256)),
                                       dim3(256), 0, st, w, w2, K, C, R, S);
                }
                nemar_narrow_fwd(gy, w2, nullptr, gx0, N, K, OH, OW, C, R, R - 1 - pad, BORDER_ZERO, ACT_NONE, 0.f, nullptr, 0, st);
            } else {
                // split reductions (see dgrad_layout): each split stores its partial gradient to its own slab, summed in order
                if (L.ksplit > 1 && !bias && act == ACT_NONE && mskip == 0 && (gx1 == nullptr || (gx0 != nullptr && !ring))) {
                    p.ksplit = L.ksplit;
                    p.part = wsf + L.slab_off;
                    p.part_stride = (long long)N * C * H * W;
                    if (gx1) { p.M0 = C; p.dst1 = nullptr; }      // two destinations: the slabs hold all C rows, the sum pass parts them
                }
                // 3x3 reflect layers that run on the wave-specialised 16-byte-load kernel fold the border INTO the main launch
                // (reflect_aux_kernel); everything else adds the border ring with a second launch below
                bool vec = false;
                if (ring && g_reflect_aux && pad == 1 && R == 3 && S == 3 && H >= 4 && W >= 8 && route_ws2(p, &vec) && vec) {
                    float* rows = wsf + L.aux_rows_off;
                    float* cols = wsf + L.aux_cols_off;
                    hipLaunchKernelGGL(reflect_aux_kernel, dim3(nemar_stream_grid(6ll * N * K * W + 8ll * N * K * H, 256)),
                                       dim3(256), 0, st, gy, rows, cols, N, K, H, W);
                    p.rf = 1; p.rf_row = rows; p.rf_col = cols;
                    ring_done = true;
                }
                launch_igemm(p, st);
                if (p.ksplit > 1 && gx1) nemar_sum_partials_two(p.part, p.part_stride, p.ksplit, gx0, gx1, N, C0, C1, H * W, st);
                else if (p.ksplit > 1) nemar_sum_partials(p.part, p.part_stride, p.ksplit, gx0, p.part_stride, false, st);
                p.rf = 0;
            }
            if (ring && !ring_done) {
                // same weights (stride 1: every tap, same order), taps re-based to padded coordinates
                dgrad_taps(p.taps, R, S, 0, 1, 0, 0);
                const int ring_len = L.ring_len;
                p.ring_p = pad; p.ring_H = H; p.ring_W = W;
                p.ksplit = L.ring_ksplit;
                p.part = L.ring_ksplit > 1 ? wsf + L.ring_slab_off : nullptr;
                p.part_stride = (long long)N * Mc * ring_len;
                p.OH = 1; p.OW = ring_len; p.P = N * ring_len;
                p.fd_ohw = make_fastdiv(ring_len); p.fd_ow = make_fastdiv(ring_len);
                float* ring_buf = wsf + L.ring_off;
                p.dst0 = ring_buf; p.dst1 = nullptr; p.M0 = Mc;
                launch_igemm(p, st);
                if (p.ksplit > 1) nemar_sum_partials(p.part, p.part_stride, p.ksplit, ring_buf, p.part_stride, false, st);
                RingBand band;
                band.nrows = band.ncols = 0;
                for (int t = 0; t < H; ++t)
                    if ((t >= 1 && t <= pad) || (t <= H - 2 && t >= H - 1 - pad)) band.rows[band.nrows++] = t;
                for (int t = 0; t < W; ++t)
                    if ((t >= 1 && t <= pad) || (t <= W - 2 && t >= W - 1 - pad)) band.cols[band.ncols++] = t;
                const long long planes = (long long)N * Mc;
                const long long work = planes * ((long long)band.nrows * W + (long long)H * band.ncols);
                if (work > 0)
                    hipLaunchKernelGGL(ring_gather_kernel, dim3(nemar_stream_grid(work, 256)), dim3(256), 0, st,
                                       (const float*)ring_buf, gx0 ? gx0 : gx1, H, W, pad, ring_len, band, planes);
            }
        }
    if (fold) {
        const long long total = (long long)N * C * H * W;
        hipLaunchKernelGGL(reflect_fold_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st,
                           (const float*)padded, gx0 ? gx0 : gx1, H, W, pad, total);
    }
    g_last_route = 0;
    NEMAR_CHECK_LAUNCH("conv2d_bwd_data");
    return NEMAR_OK;
}

// ---- weight gradient ------------------------------------------------------------------------------------------------
namespace {
void legacy_wgrad_plan(int K, int J, int P, int* splits_out, int* pix_per_split_out) {
    const bool wide = K > 32;
    const int BM = wide ? 128 : 32, BN = wide ? 128 : 256;
    const int mt = nemar_cdiv(K, BM), jt = nemar_cdiv(J, BN);
    // split the pixel reduction so that ~4 workgroups per CU exist, but keep >= 8 stages per split
    int splits = nemar_cdiv(1024, mt * jt);
    const int max_splits = nemar_cdiv(P, WBK * 8);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    *pix_per_split_out = nemar_cdiv(nemar_cdiv(P, splits), WBK) * WBK;
    *splits_out = nemar_cdiv(P, *pix_per_split_out);
}
constexpr int BIAS_CHUNK = 4096;

// Layers whose gy planes are not a multiple of 4 floats (the discriminator's 31x31 / 15x15 maps) cannot be read in aligned
// 16-byte chunks; instead of the first-generation VGPR-staged kernel (62 TF on the 256->512 k4 layer) gy is copied once into
// planes of OHv >= OH rows with (OHv * OW) % 4 == 0, zero-filled below row OH, and the wave-specialised kernel runs on the
// virtual OHv x OW map: the extra rows multiply whatever source texel they address by zero.
int padded_rows(int OH, int OW) {
    int ohv = OH;
    while ((ohv * OW) % 4) ++ohv;
    return ohv;
}
__global__ __launch_bounds__(256) void pad_planes_kernel(const float* __restrict__ src, float* __restrict__ dst, int plane,
                                                         int plane_padded, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pl = idx / plane_padded;
        const int r = (int)(idx - pl * plane_padded);
        dst[idx] = r < plane ? src[pl * plane + r] : 0.f;
    }
}
bool wgrad_pad_route(int K, int OH, int OW, int pad_mode) {
    return K > 4 && (OH * OW) % 4 != 0 && pad_mode == BORDER_ZERO && g_wgrad != 1;
}
}  // namespace

// Scratch of the weight / bias gradient: per-split slabs of the fixed-order reduction (max over the kernels the shape can
// be routed to; the routing also depends on the alignment of gy, unknown here).
NEMAR_API size_t nemar_conv2d_bwd_weight_workspace(int N, int C, int H, int W, int K, int OH, int OW, int R, int S,
                                                   int stride, int pad) {
    if (N <= 0 || C <= 0 || K <= 0 || OH <= 0 || OW <= 0 || R <= 0 || S <= 0) return 0;
    const int J = C * R * S, P = N * OH * OW;
    size_t fl = 0;
    int splits, pps;
    nemar_wgrad2_plan(K, J, P, g_wgrad_blocks, &splits, &pps);
    fl = (size_t)splits * ((size_t)K * J + K);
    legacy_wgrad_plan(K, J, P, &splits, &pps);
    const size_t f2 = (size_t)splits * ((size_t)K * J + K);
    if (f2 > fl) fl = f2;
    if ((OH * OW) % 4 != 0 && K > 4) {          // padded-gy route: slabs of the virtual map + the padded copy of gy
        const int ohv = padded_rows(OH, OW);
        nemar_wgrad2_plan(K, J, N * ohv * OW, g_wgrad_blocks, &splits, &pps);
        const size_t f4 = (size_t)splits * ((size_t)K * J + K) + 4 + (size_t)N * K * ohv * OW;
        if (f4 > fl) fl = f4;
    }
    if (K <= 4) {
        const size_t f3 = (size_t)nemar_narrow_wgrad_splits(N, C, OH, OW) * K * J + (size_t)N * nemar_cdiv(OH * OW, BIAS_CHUNK) * K;
        if (f3 > fl) fl = f3;
    }
    if (nemar_s16g_wgrad_eligible(N, C, 0, H, W, K, OH, OW, R, S, stride, pad, BORDER_ZERO)) {      // (slab count: same for any channel split)
        const size_t f6 = (size_t)nemar_s16g_wgrad_slabs_max(N, C, K, OH, W, stride) * ((size_t)K * J + K);
        if (f6 > fl) fl = f6;
    }
    if (nemar_k7_wgrad_eligible(N, C, H, W, K, R, S, stride, pad)) {            // 7x7 stem / head: slabs + max words + bias partials
        const size_t f7 = nemar_k7_wgrad_floats(N, C, H, W, K) + (size_t)N * nemar_cdiv(OH * OW, BIAS_CHUNK) * K;
        if (f7 > fl) fl = f7;
    }
    if (nemar_split16_wgrad_eligible(N, C, H, W, K, R, S, stride, pad)) {       // slabs of the split-16 route + bias partials
        const size_t f5 = (size_t)nemar_split16_wgrad_splits(N, C, H, W, K, R) * K * J + (size_t)N * nemar_cdiv(OH * OW, BIAS_CHUNK) * K;
        if (f5 > fl) fl = f5;
    }
    return sizeof(float) * fl;
}

// gw[K][C][R][S] += d loss / d w, and (gb != NULL) gb[K] += sum_pixels gy   (always accumulate: the caller
// zero-fills once per optimizer step)
NEMAR_API int nemar_conv2d_bwd_weight(const float* x0, int C0, const float* x1, int C1, const float* gy, float* gw,
                                      float* gb, int N, int H, int W, int K, int OH, int OW, int R, int S, int stride,
                                      int pad, int pad_mode, void* workspace, size_t ws_bytes, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x0 && gy && gw, "conv2d_bwd_weight: null pointer");
    NEMAR_REQUIRE(C0 > 0 && C1 >= 0 && (C1 == 0 || x1), "conv2d_bwd_weight: bad channel split");
    NEMAR_REQUIRE(N > 0 && H > 0 && W > 0 && K > 0 && OH > 0 && OW > 0 && R > 0 && S > 0, "conv2d_bwd_weight: bad shape");
    NEMAR_REQUIRE(pad_mode == BORDER_ZERO || (pad < H && pad < W), "conv2d_bwd_weight: reflect pad too large");
    NEMAR_REQUIRE((long long)N * OH * OW < (1ll << 31) && (long long)(C0 + C1) * H * W < (1ll << 31),
                  "conv2d_bwd_weight: problem too large for 32-bit tile indexing");
    float* part = nullptr;
    if (g_deterministic) {
        const size_t need = nemar_conv2d_bwd_weight_workspace(N, C0 + C1, H, W, K, OH, OW, R, S, stride, pad);
        if (!workspace || ws_bytes < need) {
            nemar_set_error("conv2d_bwd_weight: workspace %zu < %zu", workspace ? ws_bytes : (size_t)0, need);
            return NEMAR_EWORKSPACE;
        }
        part = (float*)workspace;
    }
    hipStream_t st = (hipStream_t)stream;
    const int J = (C0 + C1) * R * S;
    if (g_k7 && part && C1 == 0 && nemar_k7_wgrad_eligible(N, C0, H, W, K, R, S, stride, pad)) {
        // 7x7 stem / head (<= 4 channels on one side): reduction over pixels on the 16-bit matrix pipe (conv_k7.hip)
        if (!nemar_k7_wgrad(x0, gy, gw, gb, N, C0, H, W, K, pad_mode, part, st)) {      // (head: K <= 4 planes of gy, its own small reduction)
            const int chunks = nemar_cdiv(OH * OW, BIAS_CHUNK);
            float* pb = part + nemar_k7_wgrad_floats(N, C0, H, W, K);
            hipLaunchKernelGGL(bias_grad_kernel, dim3(K, N, chunks), dim3(256), 0, st, gy, pb, N, K, OH * OW, BIAS_CHUNK);
            nemar_sum_partials(pb, K, N * chunks, gb, K, true, st);
        }
        g_last_route = 4;
        NEMAR_CHECK_LAUNCH("conv2d_bwd_weight (7x7, 16-bit pipe)");
        return NEMAR_OK;
    }
    if (nemar_narrow_eligible(K, C1, R, S, stride, N, OH, OW) && g_narrow) {
        nemar_narrow_wgrad(x0, gy, gw, N, C0, H, W, K, R, pad, pad_mode, part, st);
        if (gb) {
            const int chunks = nemar_cdiv(OH * OW, BIAS_CHUNK);
            float* pb = part ? part + (size_t)nemar_narrow_wgrad_splits(N, C0, OH, OW) * K * J : nullptr;
            if (pb) {
                hipLaunchKernelGGL(bias_grad_kernel, dim3(K, N, chunks), dim3(256), 0, st, gy, pb, N, K, OH * OW, BIAS_CHUNK);
                nemar_sum_partials(pb, K, N * chunks, gb, K, true, st);
            } else {
                hipLaunchKernelGGL(bias_grad_atomic_kernel, dim3(K, N, chunks), dim3(256), 0, st, gy, gb, N, K, OH * OW, BIAS_CHUNK);
            }
        }
        g_last_route = 1;
        NEMAR_CHECK_LAUNCH("conv2d_bwd_weight (narrow)");
        return NEMAR_OK;
    }
    const bool s16g_wg = g_s16g_wgrad && part && s16g_worth_it((long long)N * OH * OW * K * (C0 + C1) * R * S) &&
        nemar_s16g_wgrad_eligible(N, C0, C1, H, W, K, OH, OW, R, S, stride, pad, pad_mode);
    const bool split16_wg = g_split16 && g_split16_variant == 4 && part && C1 == 0 && split16_worth_it(N, OH, OW, K, C0, R, S) &&
        nemar_split16_wgrad_eligible(N, C0, H, W, K, R, S, stride, pad) &&
        (R == 3 || pad_mode == BORDER_ZERO) && g_scratch && g_scratch_bytes >= nemar_split16_wgrad_scratch_bytes(N, C0, H, W, K, R);
    if (s16g_wg && (g_s16g_wgrad_first || !split16_wg)) {
        nemar_s16g_wgrad(x0, C0, x1, C1, gy, gw, gb, N, H, W, K, OH, OW, R, stride, pad_mode, part, g_dbg, st);
        g_last_route = 3;
        NEMAR_CHECK_LAUNCH("conv2d_bwd_weight (16-bit pipe, in-kernel split)");
        return NEMAR_OK;
    }
    if (g_split16 && g_split16_variant == 4 && part && C1 == 0 && split16_worth_it(N, OH, OW, K, C0, R, S) &&
        nemar_split16_wgrad_eligible(N, C0, H, W, K, R, S, stride, pad) &&
        (R == 3 || pad_mode == BORDER_ZERO) && g_scratch && g_scratch_bytes >= nemar_split16_wgrad_scratch_bytes(N, C0, H, W, K, R)) {
        // wide 3x3 stride-1 layers: fp16 x 3 on the 16-bit matrix pipe (conv_split16_wgrad.hip); bias gradient as its own reduction
        nemar_split16_wgrad(x0, gy, gw, N, C0, H, W, K, R, pad_mode == BORDER_REFLECT ? 1 : 0, g_scratch, part, g_xcd_map,
                            R == 3 ? t_src2_planes : nullptr, st);
        if (gb) {
            const int chunks = nemar_cdiv(OH * OW, BIAS_CHUNK);
            float* pb = part + (size_t)nemar_split16_wgrad_splits(N, C0, H, W, K, R) * K * J;
            hipLaunchKernelGGL(bias_grad_kernel, dim3(K, N, chunks), dim3(256), 0, st, gy, pb, N, K, OH * OW, BIAS_CHUNK);
            nemar_sum_partials(pb, K, N * chunks, gb, K, true, st);
        }
        g_last_route = 2;
        NEMAR_CHECK_LAUNCH("conv2d_bwd_weight (split-16)");
        return NEMAR_OK;
    }
    if (g_wgrad != 1 && nemar_wgrad2_eligible(K, OH, OW, gy)) {
        nemar_wgrad2_launch(x0, C0, x1, C1, gy, gw, gb, N, H, W, K, OH, OW, R, S, stride, pad, pad_mode, g_wgrad_blocks,
                            g_wgrad != 2, g_dbg, part, st);
        g_last_route = 0;
        NEMAR_CHECK_LAUNCH("conv2d_bwd_weight (wide)");
        return NEMAR_OK;
    }
    if (part && wgrad_pad_route(K, OH, OW, pad_mode)) {
        const int ohv = padded_rows(OH, OW);
        int splits, pps;
        nemar_wgrad2_plan(K, J, N * ohv * OW, g_wgrad_blocks, &splits, &pps);
        float* gyp = part + (((size_t)splits * ((size_t)K * J + K) + 3) & ~(size_t)3);       // 16-byte aligned, behind the slabs
        const long long total = (long long)N * K * ohv * OW;
        hipLaunchKernelGGL(pad_planes_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, gy, gyp, OH * OW, ohv * OW, total);
        if (nemar_wgrad2_eligible(K, ohv, OW, gyp)) {
            nemar_wgrad2_launch(x0, C0, x1, C1, gyp, gw, gb, N, H, W, K, ohv, OW, R, S, stride, pad, pad_mode, g_wgrad_blocks,
                                g_wgrad != 2, g_dbg, part, st);
            g_last_route = 0;
            NEMAR_CHECK_LAUNCH("conv2d_bwd_weight (wide, padded gy)");
            return NEMAR_OK;
        }
    }
    WgradParams p;
    p.src0 = x0; p.src1 = x1; p.C0 = C0; p.C1 = C1; p.Hs = H; p.Ws = W;
    p.gy = gy; p.K = K; p.OH = OH; p.OW = OW;
    p.gw = gw; p.gb = gb; p.J = J;
    p.N = N; p.P = N * OH * OW; p.sy = stride; p.sx = stride; p.R = R; p.S = S; p.pad = pad; p.border = pad_mode;
    p.fd_ohw = make_fastdiv(OH * OW); p.fd_ow = make_fastdiv(OW);
    p.dbg = g_dbg;
    const bool wide = K > 32;
    const int BM = wide ? 128 : 32, BN = wide ? 128 : 256;
    const int mt = nemar_cdiv(K, BM), jt = nemar_cdiv(p.J, BN);
    int splits;
    legacy_wgrad_plan(K, p.J, p.P, &splits, &p.pix_per_split);
    p.part = part;
    p.partb = part ? part + (size_t)splits * K * J : nullptr;
    dim3 grid(mt, jt, splits), block(256);
    if (wide)
        hipLaunchKernelGGL((wgrad_kernel<2, 2, 2, 2>), grid, block, 0, st, p);
    else
        hipLaunchKernelGGL((wgrad_kernel<1, 4, 1, 2>), grid, block, 0, st, p);
    if (part) {
        nemar_sum_partials(part, (long long)K * J, splits, gw, (long long)K * J, true, st);
        if (gb) nemar_sum_partials(p.partb, K, splits, gb, K, true, st);
    }
    g_last_route = 0;
    NEMAR_CHECK_LAUNCH("conv2d_bwd_weight");
    return NEMAR_OK;
}

NEMAR_API int nemar_tune_ptr(void* p) { g_tl = (long long*)p; return NEMAR_OK; }

// Tuning switches for A/B measurements (not part of the operator contract): key 0 = 128x128 workgroup shape.
NEMAR_API int nemar_last_route(void) { return g_last_route; }
NEMAR_API int nemar_config_epoch(void) { return g_config_epoch; }

NEMAR_API int nemar_tune(int key, int value) {
    ++g_config_epoch;
    if (key == 0) { g_cfg128 = value; return NEMAR_OK; }
    if (key == 1) { g_lds_pad = value; return NEMAR_OK; }
    if (key == 2) { g_dbg = value; return NEMAR_OK; }
    if (key == 3) { g_narrow = value; return NEMAR_OK; }
    if (key == 4) { g_wgrad = value; return NEMAR_OK; }
    if (key == 6) { g_min_blocks = value > 0 ? value : 384; return NEMAR_OK; }
    if (key == 14) { g_deterministic = value != 0; return NEMAR_OK; }
    if (key == 8) { g_reflect_aux = value != 0; return NEMAR_OK; }
    if (key == 15) { g_xcd_map = value != 0; return NEMAR_OK; }
    if (key == 20) { g_split16 = value != 0; return NEMAR_OK; }
    if (key == 24) { g_s16g = value != 0; return NEMAR_OK; }
    if (key == 25) { g_s16g_min_mmac = value < 0 ? 0 : value; return NEMAR_OK; }
    if (key == 26) { g_s16g_wgrad_first = value != 0; return NEMAR_OK; }
    if (key == 27) { nemar_s16g_tune(0, value); return NEMAR_OK; }
    if (key == 29) { g_s16g_wgrad = value != 0; return NEMAR_OK; }
    if (key == 31) { nemar_norm_planes_debug(value); return NEMAR_OK; }
    if (key == 32) { g_split16_ring3 = value != 0; return NEMAR_OK; }
    if (key == 33) { g_k7 = value != 0; return NEMAR_OK; }
    if (key == 36) { g_split_act = value != 0; return NEMAR_OK; }
    if (key == 35) { g_dual_gy = value != 0; return NEMAR_OK; }
    if (key == 34) { nemar_split16_wgrad_tune(value); return NEMAR_OK; }      // wide weight gradient: 1 one gy copy (default), 0 KS shifted copies
    if (key == 30) { g_s16g_fold = value != 0; return NEMAR_OK; }
    if (key == 28) { nemar_s16g_tune(1, value); return NEMAR_OK; }
    if (key == 23) { g_split16_min_mmac = value < 0 ? 0 : value; return NEMAR_OK; }
    if (key == 21) { g_split16_variant = value == 3 ? 3 : 4; return NEMAR_OK; }      // packed images made under the other setting are stale
    if (key == 16) { g_adir = value != 0; return NEMAR_OK; }
    if (key == 17) { g_mt8 = value; return NEMAR_OK; }
    if (key == 19) { extern int g_narrow_fwd4; g_narrow_fwd4 = value != 0; return NEMAR_OK; }
    if (key == 18) { g_ring = (value == 4 || value == 5) ? value : 3; return NEMAR_OK; }
    if (key == 12) { g_ksplit = value != 0; return NEMAR_OK; }
    if (key == 11) { g_nl4_scalar = value != 0; return NEMAR_OK; }
    if (key == 10) { g_deep64 = value != 0; return NEMAR_OK; }
    if (key == 7) { g_ws2_mt = (value == 1 || value == 2 || value == 4) ? value : 0; return NEMAR_OK; }
    if (key == 5) { g_wgrad_blocks = value > 0 ? value : 512; return NEMAR_OK; }
    nemar_set_error("nemar_tune: unknown key %d", key);
    return NEMAR_EINVAL;
}

// ---- the side inputs of the wide-layer route (scratch arena, per-sample max words, producer-written planes) travel WITH the call
// (nemar_conv_extras): nothing is registered process-wide.  Inside the library they are thread-local for the duration of the call.
namespace {
struct ExtrasScope {
    const void* t0 = nullptr;
    const void* t1 = nullptr;
    const void* tp = nullptr;
    ExtrasScope(const nemar_conv_extras* ex, const void* src, const void* src2, int N, int C, int H, int W) {
        if (!ex) return;
        if (ex->scratch && ex->scratch_bytes) { t_scratch = ex->scratch; t_scratch_bytes = ex->scratch_bytes; }
        if (ex->src_max_words && ex->src_max_count > 0) { nemar_split16_set_hint(src, ex->src_max_words, ex->src_max_count); t0 = src; }
        if (src2 && ex->src2_max_words && ex->src2_max_count > 0) { nemar_split16_set_hint(src2, ex->src2_max_words, ex->src2_max_count); t1 = src2; }
        if (ex->src_planes) { nemar_split16_set_planes_hint(src, ex->src_planes, N, C, H, W); tp = src; }
        t_gy_planes_out = ex->gy_planes_out; t_gy_planes_bytes = ex->gy_planes_bytes;
        t_src2_planes = ex->src2_planes;
    }
    ~ExtrasScope() {
        t_scratch = nullptr; t_scratch_bytes = 0;
        t_gy_planes_out = nullptr; t_gy_planes_bytes = 0; t_src2_planes = nullptr;
        if (t0) nemar_split16_set_hint(t0, nullptr, 0);
        if (t1) nemar_split16_set_hint(t1, nullptr, 0);
        if (tp) nemar_split16_set_planes_hint(tp, nullptr, 0, 0, 0, 0);
    }
};
}  // namespace

// bytes of the gy planes the data-gradient call of a layer can leave behind for its weight-gradient call (nemar_conv_extras.gy_planes_out /
// .src2_planes); 0 = the layer's gradients do not both run on the wide route
NEMAR_API size_t nemar_conv2d_gy_planes_bytes(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int pad_mode) {
    if (!g_dual_gy || !g_split16 || g_split16_variant != 4 || R != 3 || S != 3 || stride != 1 || pad != 1) return 0;
    const int mode = pad_mode == BORDER_REFLECT ? SPLIT16_DGRAD_REFLECT : SPLIT16_ZERO;
    if (!split16_worth_it(N, H, W, K, C, R, S) || !nemar_split16_eligible(N, H, W, C, K, R, S, stride, pad, mode, g_split16_variant) ||
        !nemar_split16_wgrad_eligible(N, C, H, W, K, R, S, stride, pad))
        return 0;
    return nemar_split16_wgrad_g_bytes(N, H, W, K, R);
}

// 1 when the last nemar_conv2d_bwd_data_ex call on this thread filled its gy_planes_out buffer (the route it took supports it): only then
// may the buffer be handed to nemar_conv2d_bwd_weight_ex as src2_planes
NEMAR_API int nemar_last_gy_planes(void) { return t_gy_planes_written; }

NEMAR_API int nemar_conv2d_fwd_ex(const float* x0, int C0, const float* x1, int C1, const float* w, const float* bias, float* y, int N,
                                  int H, int W, int K, int R, int S, int stride, int pad, int pad_mode, int act, float slope,
                                  void* workspace, size_t ws_bytes, int prepacked, void* stream, const nemar_conv_extras* extras) {
    nemar_conv_extras e;
    if (extras) { e = *extras; e.gy_planes_out = nullptr; e.gy_planes_bytes = 0; e.src2_planes = nullptr; }
    ExtrasScope scope(extras ? &e : nullptr, x0, nullptr, N, C0 + C1, H, W);
    return nemar_conv2d_fwd(x0, C0, x1, C1, w, bias, y, N, H, W, K, R, S, stride, pad, pad_mode, act, slope, workspace, ws_bytes, prepacked, stream);
}

NEMAR_API int nemar_conv2d_bwd_data_ex(const float* gy, const float* w, const float* bias, int act, float slope, float* gx0, int C0,
                                       float* gx1, int C1, int N, int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad,
                                       int pad_mode, void* workspace, size_t ws_bytes, int prepacked, void* stream,
                                       const nemar_conv_extras* extras) {
    nemar_conv_extras e;
    if (extras) { e = *extras; e.src_planes = nullptr; e.src2_planes = nullptr; }
    ExtrasScope scope(extras ? &e : nullptr, gy, nullptr, N, K, OH, OW);
    t_gy_planes_written = 0;
    return nemar_conv2d_bwd_data(gy, w, bias, act, slope, gx0, C0, gx1, C1, N, H, W, K, OH, OW, R, S, stride, pad, pad_mode, workspace, ws_bytes,
                                 prepacked, stream);
}

NEMAR_API int nemar_conv2d_bwd_weight_ex(const float* x0, int C0, const float* x1, int C1, const float* gy, float* gw, float* gb, int N,
                                         int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad, int pad_mode,
                                         void* workspace, size_t ws_bytes, void* stream, const nemar_conv_extras* extras) {
    nemar_conv_extras e;
    if (extras) { e = *extras; e.src_planes = nullptr; e.gy_planes_out = nullptr; e.gy_planes_bytes = 0; }
    ExtrasScope scope(extras ? &e : nullptr, x0, gy, N, C0 + C1, H, W);
    return nemar_conv2d_bwd_weight(x0, C0, x1, C1, gy, gw, gb, N, H, W, K, OH, OW, R, S, stride, pad, pad_mode, workspace, ws_bytes, stream);
}

// max |t| (finite elements) per sample of a tensor, for callers that feed the same tensor to several split-16 convolution calls
// (forward + weight gradient take x, data + weight gradient take gy): computed once, registered with nemar_absmax_hint, it replaces
// the max pass inside each call.  nemar_absmax = one sample of n elements.
NEMAR_API int nemar_absmax(const float* t, long long n, void* out_word, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(t && out_word && n > 0, "absmax: null pointer");
    nemar_split16_absmax(t, 1, n, out_word, (hipStream_t)stream);
    NEMAR_CHECK_LAUNCH("absmax");
    return NEMAR_OK;
}

NEMAR_API int nemar_absmax_samples(const float* t, int samples, long long per_sample, void* out_words, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(t && out_words && samples > 0 && samples <= 65535 && per_sample > 0, "absmax_samples: bad arguments");
    nemar_split16_absmax(t, samples, per_sample, out_words, (hipStream_t)stream);
    NEMAR_CHECK_LAUNCH("absmax_samples");
    return NEMAR_OK;
}

// bench.py's roofline entry: time the main kernel (igemm_split16_kernel) of every forward / data-gradient call of the wide 3x3
// layers with HIP events recorded on the launch stream, between enable and read
NEMAR_API int nemar_kernel_timer(int enable) {
    nemar_split16_timer(enable);
    return NEMAR_OK;
}

NEMAR_API int nemar_kernel_timer_read(double* total_ms, double* total_flop, int* launches) {
    NEMAR_REQUIRE(total_ms && total_flop && launches, "kernel_timer_read: null pointer");
    *launches = nemar_split16_timer_read(total_ms, total_flop);
    return NEMAR_OK;
}

// Scratch bytes nemar_conv2d_fwd / nemar_conv2d_bwd_data want for this layer (0: the layer never uses the arena)
NEMAR_API size_t nemar_conv2d_scratch(int N, int H, int W, int K, int C, int R, int S, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || K <= 0 || C <= 0) return 0;
    if (!g_split16 || !split16_worth_it(N, H + 2 * pad - R + 1, W + 2 * pad - S + 1, K, C, R, S)) return 0;
    size_t b = 0;
    if (nemar_split16_eligible(N, H, W, K, C, R, S, stride, pad, SPLIT16_ZERO, 4)) b = nemar_split16_scratch_total(N, H, W, K, C, H, W);
    if (nemar_split16_eligible(N, H, W, C, K, R, S, stride, pad, SPLIT16_ZERO, 4)) {
        const size_t d = nemar_split16_scratch_total(N, H, W, C, K, H, W);
        if (d > b) b = d;
    }
    if (nemar_split16_wgrad_eligible(N, C, H, W, K, R, S, stride, pad)) {
        const size_t d = nemar_split16_wgrad_scratch_bytes(N, C, H, W, K, R);
        if (d > b) b = d;
    }
    return b;
}

NEMAR_API size_t nemar_bias_grad_workspace(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return sizeof(float) * (size_t)N * nemar_cdiv(HW, BIAS_CHUNK) * C;
}

// gb[C] += sum over N and the plane of g [N,C,HW]   (bias gradient; also ConvTranspose2d's)
NEMAR_API int nemar_bias_grad(const float* g, float* gb, int N, int C, int HW, void* workspace, size_t ws_bytes,
                              void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(g && gb && N > 0 && C > 0 && HW > 0, "bias_grad: bad arguments");
    const int chunks = nemar_cdiv(HW, BIAS_CHUNK);
    if (g_deterministic) {
        const size_t need = nemar_bias_grad_workspace(N, C, HW);
        if (!workspace || ws_bytes < need) {
            nemar_set_error("bias_grad: workspace %zu < %zu", workspace ? ws_bytes : (size_t)0, need);
            return NEMAR_EWORKSPACE;
        }
        hipLaunchKernelGGL(bias_grad_kernel, dim3(C, N, chunks), dim3(256), 0, (hipStream_t)stream, g, (float*)workspace, N, C,
                           HW, BIAS_CHUNK);
        nemar_sum_partials((const float*)workspace, C, N * chunks, gb, C, true, (hipStream_t)stream);
    } else {
        hipLaunchKernelGGL(bias_grad_atomic_kernel, dim3(C, N, chunks), dim3(256), 0, (hipStream_t)stream, g, gb, N, C, HW,
                           BIAS_CHUNK);
    }
    NEMAR_CHECK_LAUNCH("bias_grad");
    return NEMAR_OK;
}
