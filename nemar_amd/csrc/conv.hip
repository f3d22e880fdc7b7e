// K1 + K2 + K4 + K8: the convolution operators — entry points, routing, workspace layouts.
//
// Replaces nn.Conv2d / nn.ConvTranspose2d (+ the ReflectionPad2d in front of them and the torch.cat that
// feeds them) at reference models/networks.py:349-377,418-439,576-597 and models/stn/layers.py:85,
// models/stn/unet_stn.py:80,97, models/stn/affine_stn.py:69-72,79 — forward, data gradient and weight
// gradient.  nn.Linear of the affine head is the 1x1 case on a 1x1 image.
//
// Each call is routed to one kernel family (nemar_last_route): the 16-bit-pipe kernels with fp32-accurate split operands
// (conv_split16*.hip: the wide residual-block layers; conv_s16g*.hip: general layers; conv_k7.hip: the 7x7 stem / head), the VALU
// kernels for <= 4 channels (conv_narrow.hip), or the exact-fp32 tap-table implicit GEMM (conv_exact.hip), which serves forward AND
// data gradient (and therefore ConvTranspose2d, the data gradient of a strided conv):
//   - a stride-2 data-gradient is four launches, one per output-pixel parity class, each with the subset of
//     taps that lands on integer source positions;
//   - a stride-1 reflect data-gradient is the zero-padded data-gradient on the unpadded domain plus a small launch over
//     the border ring whose results are added to the texels the padding mirrored (nemar_conv2d_bwd_data).
//
// The weight gradient is a second implicit GEMM, dW[k][c,r,s] = sum_pixels gy[k,p] * src[c, p (+) tap], with
// the (huge) pixel reduction split across workgroups into per-split slabs summed in order:
// conv_wgrad.hip (wave-specialised) for everything whose gy planes are 16-byte chunkable, wgrad_kernel below for the rest.
#include "common.h"
#include "conv_exact.h"
#include "conv_split16.h"
#include "conv_s16g.h"
#include "conv_k7.h"
#include "pack_plan.h"

#ifdef NEMAR_AB
extern int g_split16_ring3;                      // conv_split16.hip
extern int g_split16_ksplit_cap;
extern int g_narrow_fwd4;                        // conv_narrow.hip
extern int g_wg_xreg;                            // conv_split16_wgrad.hip
void nemar_norm_planes_debug(int bits);          // norm_planes.hip: ablation bits of the fused producer (measurement only)
#endif

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
void nemar_sum_partials_pair(const float* part_a, long long stride_a, int splits_a, float* dst_a, long long n_a,
                             const float* part_b, long long stride_b, int splits_b, float* dst_b, long long n_b, bool accumulate, hipStream_t st);
void nemar_sum_partials_fold(const float* part, long long stride, int splits, float* gx, long long planes, int H, int W, int pad,
                             const float* addend, hipStream_t st);
void nemar_sum_partials(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate,
                        hipStream_t st);

using namespace nemar_exact;

namespace {

// key 20: 3x3 / stride-1 layers with >= 128 output channels run on the bf16 matrix pipe with three-way split operands
// (conv_split16.hip) whenever the caller has registered a scratch arena large enough for the split source planes
static NEMAR_SWITCH(int, g_split16, 1);
// key 23: the split-16 route needs work to amortise its extra launches (max pass, split pass, slab sum): layers below this many
// million multiply-adds (default 2000 = 4 GFLOP, ~40 us on the exact-fp32 kernels) stay on those — BASELINE config 1's 32x32
// batch-1 resblocks (0.6 GMAC) lost 2 ms per step to launch overhead on the split-16 route
static NEMAR_SWITCH(long long, g_split16_min_mmac, 2000);
static NEMAR_SWITCH(int, g_split16_variant, 4);   // key 21: 4 fp16 x 3 products (default), 3 bf16 x 6 products, 0 bf16 x 6 on the first-generation
                                // kernel with loader waves (kept for the A/B numbers in DESIGN.md)
// which kernel family served the last conv call of this thread (nemar_last_route; tests and tools): 0 exact-fp32 implicit GEMM,
// 1 narrow (<= 4 channel) VALU kernels, 2 split-16 kernel of the wide residual-block layers, 3 general 16-bit-pipe kernels
static thread_local int g_last_route = 0;
static NEMAR_SWITCH(int, g_config_epoch, 0);      // A/B build: bumped by every nemar_tune (routes and packed-weight formats may have changed)
static NEMAR_SWITCH(int, g_k7, 1);               // key 33: the 7x7 stem / head layers (<= 4 channels on one side) on the 16-bit matrix pipe (conv_k7.hip)
static NEMAR_SWITCH(int, g_s16g, 1);             // key 24: general layers on the 16-bit matrix pipe with the in-kernel operand split (conv_s16g.hip)
static NEMAR_SWITCH(int, g_s16g_wgrad_first, 0);  // key 26: 1 = the in-kernel-split weight gradient also takes the wide residual-block layers (stand-alone 374 vs
                                    // 393 us per call, but 44.3 vs 41.8 ms per step inside the bench: off)
static NEMAR_SWITCH(int, g_s16g_wgrad, 1);        // key 29: weight gradients on the in-kernel-split kernels (conv_s16g_wgrad.hip)
static NEMAR_SWITCH(int, g_s16g_fold, 1);         // key 30: stride-1 reflect data gradients on the padded domain + fold
static NEMAR_SWITCH(long long, g_s16g_min_mmac, 30);   // key 25: ... above this many million multiply-adds (tiny layers are launch-bound either way)
static thread_local void* t_scratch = nullptr;        // nemar_conv2d_*_ex: this call's scratch arena (nemar_conv_extras.scratch)
static thread_local size_t t_scratch_bytes = 0;
static thread_local void* t_gy_planes_out = nullptr;  // bwd_data_ex: where the pass that splits gy also leaves the weight gradient's planes
static thread_local size_t t_gy_planes_bytes = 0;
static thread_local const void* t_src2_planes = nullptr;      // bwd_weight_ex: those planes
static thread_local const void* t_x_wplanes = nullptr;        // bwd_weight_ex: the X planes of x a forward producer wrote (extras.src_planes)
static thread_local const float* t_addend = nullptr;          // bwd_data_ex: tensor added to gx0 in the epilogue (extras.addend)
static thread_local void* t_out_max = nullptr;                // bwd_data_ex: per-sample max |gx0| words (extras.out_max_words)
static thread_local int t_fused_epilogue = 0;                 // did the last bwd_data_ex call honour them?
static thread_local int t_bias_rode = 0;                       // ... and did the call reduce them (the wide route)?
static thread_local const float* t_bias_partials = nullptr;   // bwd_weight_ex: per-plane sums of gy [N, K] (extras.bias_partials)
static thread_local int t_addend_done = 0;                    // ... or at least the addend (the fold pass of a small reflect layer: nemar_conv2d_bwd_data_addend_ok)
static NEMAR_SWITCH(int, g_split_act, 1);          // key 36: reduction-split forward layers with a fused ReLU / LeakyReLU (activation in the sum pass)
static NEMAR_SWITCH(int, g_fold_small, 1);         // key 43: stride-1 reflect data gradients of tiny maps on the exact route: padded domain + sum-and-fold pass
static NEMAR_SWITCH(int, g_dual_gy, 1);            // key 35: the data-gradient call's split pass also writes the weight gradient's gy planes
static thread_local int t_gy_planes_written = 0;      // did the last bwd_data_ex call on this thread fill gy_planes_out?
#define g_scratch t_scratch
#define g_scratch_bytes t_scratch_bytes
static NEMAR_SWITCH(int, g_reflect_aux, 1);    // tuning switch (key 8): 3x3 reflect data gradient folds the border into the main launch (1) / ring launch (0)
static NEMAR_SWITCH(int, g_deterministic, 1);  // tuning switch (key 14): 1 = split reductions go through per-split slabs summed in order (bitwise
                                 // reproducible backward pass), 0 = fp32 atomics in the weight / bias gradients (round-1 scheme)
static NEMAR_SWITCH(int, g_ksplit, 1);         // tuning switch (key 12): allow reduction splits in the wave-specialised data gradient
static NEMAR_SWITCH(int, g_narrow, 1);   // tuning switch (key 3): route <=4-channel layers to the VALU kernels
static NEMAR_SWITCH(int, g_wgrad, 0);    // tuning switch (key 4): 0 = wave-specialised wide weight gradient, 1 = VGPR-staged kernel, 2 = wave-specialised without 16-byte source loads
static NEMAR_SWITCH(int, g_wgrad_blocks, 512);   // tuning switch (key 5): workgroups targeted by the pixel split

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
// (+ addend[n,c,h,w] where given: the skip gradient of a ResnetBlock rides in the pass that writes the data gradient of its first convolution)
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ gp, float* __restrict__ gx, int H,
                                                           int W, int pad, long long total, const float* __restrict__ addend) {
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
        gx[idx] = addend ? s + addend[idx] : s;
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
    bool ring, fold, fold16, fold_small;
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
    // Tiny stride-1 reflect layers that stay on the exact-fp32 kernels (the <= 16 x 16 maps of the registration net's ResnetBlocks): the
    // padded-domain form as well — ONE implicit-GEMM launch over the (H + 2p) x (W + 2p) domain with its reduction split into slabs, then ONE
    // pass that sums the slabs and folds the mirrored border (nemar_sum_partials_fold).  The ring form of the same layer is five
    // launches of 4 - 8 us each on the step's critical chain: interior, its slab sum, border ring, its slab sum, gather.
    L.fold_small = L.ring && !L.fold16 && g_fold_small && g_ksplit && C > 4 && (H + 2 * pad) * (W + 2 * pad) <= 1296;
    if (L.fold_small) { L.ring = false; L.fold = true; }
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
    const int Hs = L.fold_small ? H + 2 * pad : H, Wsl = L.fold_small ? W + 2 * pad : W;      // the domain the split launch covers
    if (g_ksplit && stride == 1 && (!L.fold || L.fold_small) && C > 4) {
        const int P = N * Hs * Wsl, Kred = K * R * S, stages = nemar_cdiv(Kred, BK);
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
    if (L.ksplit > 1) o += (size_t)L.ksplit * N * C * Hs * Wsl;
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
    // A layer with a fused ReLU / LeakyReLU splits too: the activation is applied by the sum pass (nemar_sum_partials_act).  The registration
    // net's decoder and first-of-level layers at <= 16 x 16 (72 .. 144 serial stages in one or two workgroups: 33 / 57 us per call) take
    // ~10 us + the sum; same-box A/B 27.41 -> 27.07 ms per step (profiles/r6_d_wgrad_lane_and_split_act_ab.txt).  nemar_tune(36, 0): off.
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
                               total, (const float*)nullptr);
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
            nemar_split16_set_epilogue(t_addend, t_out_max);
            t_gy_planes_written = nemar_split16_conv(gy, workspace, nullptr, gx0, N, H, W, C, K, R, R - 1 - pad, OH, OW, H, W, mode, g_scratch,
                                                     g_xcd_map, g_split16_variant, g_tl, dual, st) ? 1 : 0;
            t_fused_epilogue = nemar_split16_epilogue_done();
            nemar_split16_set_epilogue(nullptr, nullptr);
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
                               W, pad, total, t_addend);
            if (t_addend) t_addend_done = 1;
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
    bool folded = false;                  // fold_small: the slab sum folded the border already
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
                if (L.ksplit > 1 && !bias && act == ACT_NONE && mskip == 0 && (fold ? (gx0 != nullptr && gx1 == nullptr) : (gx1 == nullptr || (gx0 != nullptr && !ring)))) {
                    p.ksplit = L.ksplit;
                    p.part = wsf + L.slab_off;
                    p.part_stride = (long long)N * C * Hd * Wd;       // (fold_small: slabs of the padded domain)
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
                if (p.ksplit > 1 && fold) {                       // slabs of the padded domain -> sum + fold in one pass (gx1 == nullptr: checked above)
                    nemar_sum_partials_fold(p.part, p.part_stride, p.ksplit, gx0, (long long)N * C, H, W, pad, t_addend, st);
                    if (t_addend) t_addend_done = 1;
                    folded = true;
                }
                else if (p.ksplit > 1 && gx1) nemar_sum_partials_two(p.part, p.part_stride, p.ksplit, gx0, gx1, N, C0, C1, H * W, st);
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
    if (fold && !folded) {
        const long long total = (long long)N * C * H * W;
        const float* const add = (gx0 && !gx1) ? t_addend : nullptr;
        hipLaunchKernelGGL(reflect_fold_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st,
                           (const float*)padded, gx0 ? gx0 : gx1, H, W, pad, total, add);
        if (add) t_addend_done = 1;
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
        const bool bias_rides = gb && t_bias_partials;           // the producer's per-plane sums: reduced inside the slab-sum launch
        if (bias_rides) { nemar_split16_wgrad_set_bias(t_bias_partials, gb); t_bias_rode = 1; }
        nemar_split16_wgrad(x0, gy, gw, N, C0, H, W, K, R, pad_mode == BORDER_REFLECT ? 1 : 0, g_scratch, part, g_xcd_map,
                            R == 3 ? t_src2_planes : nullptr, (R == 3 && pad_mode == BORDER_REFLECT) ? t_x_wplanes : nullptr, st);
        if (gb && !bias_rides) {
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
    if (part) nemar_sum_partials_pair(part, (long long)K * J, splits, gw, (long long)K * J, gb ? p.partb : nullptr, K, splits, gb, K, true, st);
    g_last_route = 0;
    NEMAR_CHECK_LAUNCH("conv2d_bwd_weight");
    return NEMAR_OK;
}

NEMAR_API int nemar_last_route(void) { return g_last_route; }
NEMAR_API int nemar_config_epoch(void) { return g_config_epoch; }      // product build: always 0 (no switch can change)

#ifdef NEMAR_AB
// Measurement switches (include/nemar_hip_ab.h): only libnemar_hip_ab.so has these entry points.
NEMAR_API int nemar_tune_ptr(void* p) { g_tl = (long long*)p; return NEMAR_OK; }

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
    if (key == 43) { g_fold_small = value != 0; return NEMAR_OK; }      // tiny stride-1 reflect data gradients: padded domain + sum-and-fold (1) / interior + ring (0)
    if (key == 37) { g_lds_claim = value; return NEMAR_OK; }      // kernel families whose workgroups claim the whole CU's LDS (common.h)
    if (key == 38) { g_wg_xreg = value != 0; return NEMAR_OK; }   // wide 3x3 weight gradient: X pieces through registers
    if (key == 39) { g_split16_ksplit_cap = value < 1 ? 1 : (value > 8 ? 8 : value); return NEMAR_OK; }      // most reduction runs per tile of the wide-layer kernel
    if (key == 35) { g_dual_gy = value != 0; return NEMAR_OK; }
    if (key == 34) { nemar_split16_wgrad_tune(value); return NEMAR_OK; }      // wide weight gradient: 1 one gy copy (default), 0 KS shifted copies
    if (key == 30) { g_s16g_fold = value != 0; return NEMAR_OK; }
    if (key == 28) { nemar_s16g_tune(1, value); return NEMAR_OK; }
    if (key == 40) { nemar_s16g_tune(2, value); return NEMAR_OK; }
    if (key == 42) { nemar_s16g_tune(4, value); return NEMAR_OK; }      // class-fused stride-2 data gradients (conv_s16g.hip CF): 0 off, 1 on (default), 2 on + required
    if (key == 41) { nemar_s16g_tune(3, value); return NEMAR_OK; }      // ... while the grid keeps this many workgroups (256)      // most channel blocks per s16g workgroup (4; 1 = one workgroup per block)
    if (key == 23) { g_split16_min_mmac = value < 0 ? 0 : value; return NEMAR_OK; }
    if (key == 21) { g_split16_variant = value == 3 ? 3 : 4; return NEMAR_OK; }      // packed images made under the other setting are stale
    if (key == 16) { g_adir = value != 0; return NEMAR_OK; }
    if (key == 17) { g_mt8 = value; return NEMAR_OK; }
    if (key == 19) { g_narrow_fwd4 = value != 0; return NEMAR_OK; }
    if (key == 18) { g_ring = (value == 4 || value == 5) ? value : 3; return NEMAR_OK; }
    if (key == 12) { g_ksplit = value != 0; return NEMAR_OK; }
    if (key == 11) { g_nl4_scalar = value != 0; return NEMAR_OK; }
    if (key == 10) { g_deep64 = value != 0; return NEMAR_OK; }
    if (key == 7) { g_ws2_mt = (value == 1 || value == 2 || value == 4) ? value : 0; return NEMAR_OK; }
    if (key == 5) { g_wgrad_blocks = value > 0 ? value : 512; return NEMAR_OK; }
    nemar_set_error("nemar_tune: unknown key %d", key);
    return NEMAR_EINVAL;
}
#endif  // NEMAR_AB

// ---- the side inputs of the wide-layer route (scratch arena, per-sample max words, producer-written planes) travel WITH the call
// (nemar_conv_extras): nothing is registered process-wide.  Inside the library they are thread-local for the duration of the call.
namespace {
struct ExtrasScope {
    const void* t0 = nullptr;
    const void* t1 = nullptr;
    const void* tp = nullptr;
    // planes_kind: the SPLIT16_* content extras.src_planes holds for this call (-1: the call takes no channel-blocked planes)
    ExtrasScope(const nemar_conv_extras* ex, const void* src, const void* src2, int N, int C, int H, int W, int planes_kind) {
        if (!ex) return;
        if (ex->scratch && ex->scratch_bytes) { t_scratch = ex->scratch; t_scratch_bytes = ex->scratch_bytes; }
        if (ex->src_max_words && ex->src_max_count > 0) { nemar_split16_set_hint(src, ex->src_max_words, ex->src_max_count); t0 = src; }
        if (src2 && ex->src2_max_words && ex->src2_max_count > 0) { nemar_split16_set_hint(src2, ex->src2_max_words, ex->src2_max_count); t1 = src2; }
        if (ex->src_planes && planes_kind >= 0) { nemar_split16_set_planes_hint(src, ex->src_planes, N, C, H, W, planes_kind); tp = src; }
        t_gy_planes_out = ex->gy_planes_out; t_gy_planes_bytes = ex->gy_planes_bytes;
        t_src2_planes = ex->src2_planes;
        t_addend = ex->addend; t_out_max = ex->out_max_words;
    }
    ~ExtrasScope() {
        t_scratch = nullptr; t_scratch_bytes = 0;
        t_gy_planes_out = nullptr; t_gy_planes_bytes = 0; t_src2_planes = nullptr;
        t_addend = nullptr; t_out_max = nullptr; t_x_wplanes = nullptr; t_bias_partials = nullptr;
        if (t0) nemar_split16_set_hint(t0, nullptr, 0);
        if (t1) nemar_split16_set_hint(t1, nullptr, 0);
        if (tp) nemar_split16_set_planes_hint(tp, nullptr, 0, 0, 0, 0, 0);
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

// 1 when nemar_conv2d_bwd_data_ex of this layer honours nemar_conv_extras.addend / .out_max_words and takes gy as producer-written planes
// (.src_planes), and its nemar_conv2d_bwd_weight_ex takes both operands as planes: the wide 3x3 route with an unsplit reduction
NEMAR_API int nemar_conv2d_bwd_data_fusable(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int pad_mode) {
    if (nemar_conv2d_gy_planes_bytes(N, C, H, W, K, R, S, stride, pad, pad_mode) == 0) return 0;
    return nemar_split16_ksplit(N, H, W, C, K) == 1 ? 1 : 0;
}

// 1 when nemar_conv2d_bwd_data_ex of this layer (one destination, no bias, no activation) adds nemar_conv_extras.addend to the data gradient:
// the wide route's fused epilogue, or a stride-1 reflect layer whose data gradient ends with a fold pass (the general 16-bit-pipe kernel
// on the padded domain + fold; the tiny maps' split launch + sum-and-fold) — the addend is one more term of that pass.
NEMAR_API size_t nemar_conv2d_scratch(int N, int H, int W, int K, int C, int R, int S, int stride, int pad);
NEMAR_API int nemar_conv2d_bwd_data_addend_ok(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int pad_mode) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
    if (nemar_conv2d_bwd_data_fusable(N, C, H, W, K, R, S, stride, pad, pad_mode)) return 1;
    if (pad_mode != BORDER_REFLECT || pad <= 0 || stride != 1 || pad >= H || pad >= W || R != S) return 0;
    if (nemar_conv2d_scratch(N, H, W, K, C, R, S, stride, pad) > 0) return 0;      // a layer the wide route takes when its arena is given
    if (R == 7) return 0;                                                            // (the 7x7 kernels have their own borders)
    const DgradLayout L = dgrad_layout(N, C, H, W, K, R, S, stride, pad, pad_mode);
    return (L.fold16 || L.fold) ? 1 : 0;
}

// 1 when the last nemar_conv2d_bwd_data_ex call on this thread filled its gy_planes_out buffer (the route it took supports it): only then
// may the buffer be handed to nemar_conv2d_bwd_weight_ex as src2_planes
NEMAR_API int nemar_last_gy_planes(void) { return t_gy_planes_written; }

NEMAR_API int nemar_conv2d_fwd_ex(const float* x0, int C0, const float* x1, int C1, const float* w, const float* bias, float* y, int N,
                                  int H, int W, int K, int R, int S, int stride, int pad, int pad_mode, int act, float slope,
                                  void* workspace, size_t ws_bytes, int prepacked, void* stream, const nemar_conv_extras* extras) {
    nemar_conv_extras e;
    if (extras) { e = *extras; e.gy_planes_out = nullptr; e.gy_planes_bytes = 0; e.src2_planes = nullptr; e.addend = nullptr; e.out_max_words = nullptr; }
    ExtrasScope scope(extras ? &e : nullptr, x0, nullptr, N, C0 + C1, H, W, SPLIT16_REFLECT);
    return nemar_conv2d_fwd(x0, C0, x1, C1, w, bias, y, N, H, W, K, R, S, stride, pad, pad_mode, act, slope, workspace, ws_bytes, prepacked, stream);
}

NEMAR_API int nemar_conv2d_bwd_data_ex(const float* gy, const float* w, const float* bias, int act, float slope, float* gx0, int C0,
                                       float* gx1, int C1, int N, int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad,
                                       int pad_mode, void* workspace, size_t ws_bytes, int prepacked, void* stream,
                                       const nemar_conv_extras* extras) {
    nemar_conv_extras e;
    if (extras) { e = *extras; e.src2_planes = nullptr; }
    // (extras.src_planes: the data-gradient planes of gy nemar_instnorm_bwd_planes wrote, in the content of THIS layer's padding)
    ExtrasScope scope(extras ? &e : nullptr, gy, nullptr, N, K, OH, OW, pad_mode == BORDER_REFLECT ? SPLIT16_DGRAD_REFLECT : SPLIT16_ZERO);
    t_gy_planes_written = 0;
    t_fused_epilogue = 0;
    t_addend_done = 0;
    const int rc = nemar_conv2d_bwd_data(gy, w, bias, act, slope, gx0, C0, gx1, C1, N, H, W, K, OH, OW, R, S, stride, pad, pad_mode, workspace,
                                         ws_bytes, prepacked, stream);
    if (rc == NEMAR_OK && extras && ((extras->addend && !t_fused_epilogue && !t_addend_done) || (extras->out_max_words && !t_fused_epilogue))) {
        nemar_set_error("conv2d_bwd_data_ex: this layer's route has no fused epilogue (addend / out_max_words): ask nemar_conv2d_bwd_data_fusable / "
                        "nemar_conv2d_bwd_data_addend_ok first");
        return NEMAR_EINVAL;
    }
    return rc;
}

NEMAR_API int nemar_conv2d_bwd_weight_ex(const float* x0, int C0, const float* x1, int C1, const float* gy, float* gw, float* gb, int N,
                                         int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad, int pad_mode,
                                         void* workspace, size_t ws_bytes, void* stream, const nemar_conv_extras* extras) {
    nemar_conv_extras e;
    if (extras) { e = *extras; e.gy_planes_out = nullptr; e.gy_planes_bytes = 0; e.addend = nullptr; e.out_max_words = nullptr; }
    // (extras.src_planes: the weight gradient's X planes of x0 nemar_instnorm_fwd_planes wrote — pixel-major, not a channel-blocked hint)
    ExtrasScope scope(extras ? &e : nullptr, x0, gy, N, C0 + C1, H, W, -1);
    t_x_wplanes = extras ? extras->src_planes : nullptr;
    t_bias_partials = extras ? extras->bias_partials : nullptr;
    t_bias_rode = 0;
    const int rc = nemar_conv2d_bwd_weight(x0, C0, x1, C1, gy, gw, gb, N, H, W, K, OH, OW, R, S, stride, pad, pad_mode, workspace, ws_bytes, stream);
    if (rc == NEMAR_OK && extras && extras->bias_partials && gb && !t_bias_rode) {
        nemar_set_error("conv2d_bwd_weight_ex: bias_partials are only taken on the wide route (nemar_conv2d_bwd_data_fusable); gb was reduced from gy");
        return NEMAR_EINVAL;
    }
    return rc;
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
