// K1 for "narrow" layers: convolutions with at most 4 output channels — the translation net's 7x7
// RGB head (reference models/networks.py:376), the STN's 2-channel offset head (models/stn/unet_stn.py:75-77) and the
// discriminator's 1-channel logit conv (models/networks.py:597) — forward and weight gradient.
//
// On the matrix cores these layers waste 29..31 of 32 MFMA rows, so they run on the vector ALUs instead: one output
// pixel per lane, <= 4 accumulators, the source staged through LDS as a halo tile (16 channels x (8+R-1) x (32+R-1)),
// so every source texel is fetched from memory once per workgroup and re-read R*R times from LDS (128 B/clk), not R*R
// times through the texture path.  Weights are wave-uniform (scalar loads).  Border handling (zero / reflect) happens
// once, while the tile is filled.  fp32 fmaf-free multiply-add in (channel, r, s) order.
#include "common.h"

namespace {

constexpr int TW = 32, TH = 8, CH = 16;
constexpr int BORDER_REFLECT = 1;
constexpr int ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3;

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

struct NarrowParams {
    const float* x; const float* w; const float* bias; float* y;
    const float* gy; float* gw;
    int N, C, H, W, OH, OW, pad, border, act;
    float slope;
    int tiles_x, tiles_y, tiles_total;
    int csplit;   // forward: channel ranges per image (grid.z = N * csplit); > 1 => range cs stores its partial sums to slab cs
    float* part;  // slabs of the split modes (forward: [csplit][N*M*OH*OW]; weight gradient: [gridDim.x][K*C*R*R])
};

// Stage a [channels][LH][LWP] halo tile of x (image n) into LDS: element idx = (ch, ly, lx) <- x[c0 + ch][y0 + ly][x0 + lx] with the
// border rule applied (mirror / zero), zero for channels >= cend and for the pitch padding lx >= LWV.  SU loads per thread are
// issued back to back from clamped (always valid) addresses before the first one is consumed.
template <int TOTAL, int PLANE, int LWP, int LWV>
__device__ __forceinline__ void stage_tile(float* tile, const float* __restrict__ xn, int c0, int cend, int y0, int x0, int H, int W,
                                           int border) {
    constexpr int SU = 8;
    const int HW = H * W;
    for (int base = threadIdx.x; base < TOTAL; base += 256 * SU) {
        float v[SU];
        bool ok[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int idx = min(base + u * 256, TOTAL - 1);
            const int ch = idx / PLANE;
            const int rem = idx - ch * PLANE;
            const int ly = rem / LWP, lxx = rem - ly * LWP;
            int iy = y0 + ly, ix = x0 + lxx;
            ok[u] = c0 + ch < cend && lxx < LWV;
            if (border == BORDER_REFLECT) {
                iy = reflect_idx(iy, H);
                ix = reflect_idx(ix, W);
            } else {
                ok[u] = ok[u] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            }
            iy = min(max(iy, 0), H - 1);
            ix = min(max(ix, 0), W - 1);
            v[u] = xn[(size_t)min(c0 + ch, cend - 1) * HW + iy * W + ix];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int idx = base + u * 256;
            if (idx < TOTAL) tile[idx] = ok[u] ? v[u] : 0.f;
        }
    }
}

// fill the LDS halo tile of channels [c0, c0+CH) for the output tile whose origin is (oy0, ox0)
template <int R>
__device__ __forceinline__ void fill_tile(float* tile, const NarrowParams& p, const float* xn, int c0, int oy0, int ox0) {
    constexpr int LH = TH + R - 1, LW = TW + R - 1;
    stage_tile<CH * LH * LW, LH * LW, LW, LW>(tile, xn, c0, p.C, oy0 - p.pad, ox0 - p.pad, p.H, p.W, p.border);
}

template <int M, int R>
__global__ __launch_bounds__(256) void narrow_fwd_kernel(NarrowParams p) {
    constexpr int LH = TH + R - 1, LW = TW + R - 1;
    __shared__ float tile[CH * LH * LW];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    // grid.z = image x channel range.  With csplit > 1 (deep, spatially small layers: the discriminator's logit conv has
    // 32 output tiles for 512 channels) each workgroup reduces its own channel range into slab `cs` of the workspace and
    // nemar_sum_partials adds the slabs in order (bitwise reproducible); range 0 carries the bias (no activation in this mode).
    const int n = blockIdx.z / p.csplit, cs = blockIdx.z - n * p.csplit;
    const int cper = ((p.C + p.csplit - 1) / p.csplit + CH - 1) / CH * CH;
    const int cbeg = cs * cper, cend = min(p.C, cbeg + cper);
    const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const float* xn = p.x + (size_t)n * p.C * p.H * p.W;
    float acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = (p.bias && cs == 0) ? p.bias[m] : 0.f;
    const int CRS = p.C * R * R;
    for (int c0 = cbeg; c0 < cend; c0 += CH) {
        __syncthreads();
        fill_tile<R>(tile, p, xn, c0, oy0, ox0);
        __syncthreads();
        const int nch = min(CH, cend - c0);
        for (int ch = 0; ch < nch; ++ch) {
            const float* t0 = tile + ch * (LH * LW) + ty * LW + tx;
            const float* wc = p.w + (size_t)(c0 + ch) * R * R;     // wave-uniform => scalar loads
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    const float v = t0[r * LW + s];
#pragma unroll
                    for (int m = 0; m < M; ++m) acc[m] += wc[(size_t)m * CRS + r * R + s] * v;
                }
        }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < p.OH && ox < p.OW) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const size_t o = (((size_t)n * M + m) * p.OH + oy) * p.OW + ox;
            if (p.csplit > 1) p.part[(size_t)cs * ((size_t)p.N * M * p.OH * p.OW) + o] = acc[m];
            else p.y[o] = act_apply(acc[m], p.act, p.slope);
        }
    }
}

// ---- register-tiled forward for wide images (OW >= 64): 4 horizontally adjacent output pixels per lane ------------------------
// narrow_fwd_kernel reads one LDS value per M multiply-adds (LDS-issue bound: 26 TF-equivalent on the 7x7 RGB head).  Here a
// lane owns pixels 4*lx .. 4*lx+3 of one row: per (channel, filter row) it reads the 4 + R - 1 source values of that row ONCE
// (16-byte LDS reads, pitch padded to 16 bytes) and spends R * M * 4 multiply-adds on them with scalar-loaded weights — 84
// FMAs per 3 LDS reads for the 64->3 7x7 layer.  Workgroup = 128 x 8 output pixels, 4 channels of halo tile per round.
constexpr int T2W = 128, T2H = 8, CH2 = 4, PX2 = 4;
template <int M, int R>
__global__ __launch_bounds__(256) void narrow_fwd4_kernel(NarrowParams p) {
    constexpr int LH = T2H + R - 1, LWP = (T2W + R - 1 + 3) / 4 * 4;       // LDS row pitch: multiple of 4 floats
    constexpr int NV = (PX2 + R - 1 + 3) / 4;                              // 16-byte reads per (channel, filter row)
    __shared__ __attribute__((aligned(16))) float tile[CH2 * LH * LWP + 8];
    const int lx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = blockIdx.z / p.csplit, cs = blockIdx.z - n * p.csplit;
    const int cper = ((p.C + p.csplit - 1) / p.csplit + CH - 1) / CH * CH;
    const int cbeg = cs * cper, cend = min(p.C, cbeg + cper);
    const int oy0 = blockIdx.y * T2H, ox0 = blockIdx.x * T2W;
    const int HW = p.H * p.W;
    const float* xn = p.x + (size_t)n * p.C * HW;
    float acc[M][PX2];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int j = 0; j < PX2; ++j) acc[m][j] = (p.bias && cs == 0) ? p.bias[m] : 0.f;
    const int CRS = p.C * R * R;
    for (int c0 = cbeg; c0 < cend; c0 += CH2) {
        __syncthreads();
        // halo tile: SU unconditional loads (clamped addresses) in flight per thread, then the selects and the LDS stores — written
        // as one load -> store per iteration the loop ran at one memory latency per element (72 k cycles per 4-channel round)
        stage_tile<CH2 * LH * LWP, LH * LWP, LWP, T2W + R - 1>(tile, xn, c0, cend, oy0 - p.pad, ox0 - p.pad, p.H, p.W, p.border);
        __syncthreads();
        const int nch = min(CH2, cend - c0);
        for (int ch = 0; ch < nch; ++ch) {
            const float* wc = p.w + (size_t)(c0 + ch) * R * R;              // wave-uniform => scalar loads
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float v[NV * 4];
                const float4* row = reinterpret_cast<const float4*>(tile + (ch * LH + ty + r) * LWP + PX2 * lx);
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const float4 t = row[q];
                    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
                }
#pragma unroll
                for (int s = 0; s < R; ++s)
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const float w = wc[(size_t)m * CRS + r * R + s];
#pragma unroll
                        for (int j = 0; j < PX2; ++j) acc[m][j] = fmaf(w, v[j + s], acc[m][j]);   // fused: half the VALU instructions
                    }
            }
        }
    }
    const int oy = oy0 + ty, ox = ox0 + PX2 * lx;
    if (oy < p.OH) {
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int j = 0; j < PX2; ++j) {
                if (ox + j < p.OW) {
                    const size_t o = (((size_t)n * M + m) * p.OH + oy) * p.OW + ox + j;
                    if (p.csplit > 1) p.part[(size_t)cs * ((size_t)p.N * M * p.OH * p.OW) + o] = acc[m][j];
                    else p.y[o] = act_apply(acc[m][j], p.act, p.slope);
                }
            }
    }
}

// gw[k][c][r][s] += sum_pixels gy[k][p] * x[c][p + (r,s) - pad].  blockIdx.y = channel chunk; blockIdx.x strides over
// output tiles, accumulating in registers; each thread owns up to PAIRS (channel, tap) columns x K rows.
template <int K, int R>
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(NarrowParams p) {
    constexpr int LH = TH + R - 1, LW = TW + R - 1;
    constexpr int NPAIR = CH * R * R;
    constexpr int PAIRS = (NPAIR + 255) / 256;
    __shared__ float tile[CH * LH * LW];
    __shared__ float gyt[K * TH * TW];
    const int c0 = blockIdx.y * CH;
    float acc[PAIRS][K];
    int toff[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const int pr = threadIdx.x + q * 256;
        const int ch = pr / (R * R), t = pr - ch * (R * R);
        const int r = t / R, s = t - r * R;
        toff[q] = (pr < NPAIR) ? ch * (LH * LW) + r * LW + s : -1;
#pragma unroll
        for (int k = 0; k < K; ++k) acc[q][k] = 0.f;
    }
    for (int tl = blockIdx.x; tl < p.tiles_total; tl += gridDim.x) {
        const int n = tl / (p.tiles_x * p.tiles_y);
        const int rem = tl - n * (p.tiles_x * p.tiles_y);
        const int tyi = rem / p.tiles_x, txi = rem - tyi * p.tiles_x;
        const int oy0 = tyi * TH, ox0 = txi * TW;
        __syncthreads();
        fill_tile<R>(tile, p, p.x + (size_t)n * p.C * p.H * p.W, c0, oy0, ox0);
        for (int idx = threadIdx.x; idx < K * TH * TW; idx += 256) {
            const int k = idx / (TH * TW), r2 = idx - k * (TH * TW);
            const int oy = oy0 + r2 / TW, ox = ox0 + (r2 - (r2 / TW) * TW);
            gyt[idx] = (oy < p.OH && ox < p.OW) ? p.gy[(((size_t)n * K + k) * p.OH + oy) * p.OW + ox] : 0.f;
        }
        __syncthreads();
        for (int py = 0; py < TH; ++py)
#pragma unroll 4
            for (int px = 0; px < TW; ++px) {
                float g[K];
#pragma unroll
                for (int k = 0; k < K; ++k) g[k] = gyt[k * (TH * TW) + py * TW + px];   // broadcast reads
#pragma unroll
                for (int q = 0; q < PAIRS; ++q) {
                    if (toff[q] >= 0) {
                        const float v = tile[toff[q] + py * LW + px];
#pragma unroll
                        for (int k = 0; k < K; ++k) acc[q][k] += g[k] * v;
                    }
                }
            }
    }
    const int CRS = p.C * R * R;
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const int pr = threadIdx.x + q * 256;
        const int ch = pr / (R * R), t = pr - ch * (R * R);
        if (pr < NPAIR && c0 + ch < p.C) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const size_t o = (size_t)k * CRS + (size_t)(c0 + ch) * R * R + t;
                if (p.part) p.part[(size_t)blockIdx.x * ((size_t)K * CRS) + o] = acc[q][k];
                else atomicAdd(p.gw + o, acc[q][k]);
            }
        }
    }
}

// ---- register-tiled weight gradient: a thread owns one (channel, filter row) and its R taps x K output channels --------------
// narrow_wgrad_kernel spends one LDS read per K multiply-adds (15-18 TF-equivalent on the 64->3 7x7 head).  Here thread (ch, r)
// keeps R * K accumulators and walks the 32 x 8 tile four pixels at a time: per group it reads the 4 + R - 1 source values of
// row py + r once (16-byte reads) and the K x 4 gy values (same address in every lane: LDS broadcast), and spends R * K * 4
// multiply-adds on them.  32 channels x R rows = 224 of 256 threads busy for R = 7.
constexpr int WCH = 32;
template <int K, int R>
__global__ __launch_bounds__(256) void narrow_wgrad4_kernel(NarrowParams p) {
    constexpr int LH = TH + R - 1, LWP = (TW + R - 1 + 3) / 4 * 4;
    constexpr int NV = (4 + R - 1 + 3) / 4;
    __shared__ __attribute__((aligned(16))) float tile[WCH * LH * LWP];
    __shared__ __attribute__((aligned(16))) float gyt[K * TH * TW];
    const int c0 = blockIdx.y * WCH;
    const int ch = threadIdx.x / R, r = threadIdx.x - ch * R;
    const bool active = ch < WCH && c0 + ch < p.C;
    float acc[R][K];
#pragma unroll
    for (int s = 0; s < R; ++s)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[s][k] = 0.f;
    const int HW = p.H * p.W;
    for (int tl = blockIdx.x; tl < p.tiles_total; tl += gridDim.x) {
        const int n = tl / (p.tiles_x * p.tiles_y);
        const int rem = tl - n * (p.tiles_x * p.tiles_y);
        const int tyi = rem / p.tiles_x, txi = rem - tyi * p.tiles_x;
        const int oy0 = tyi * TH, ox0 = txi * TW;
        const float* xn = p.x + (size_t)n * p.C * HW;
        __syncthreads();
        stage_tile<WCH * LH * LWP, LH * LWP, LWP, TW + R - 1>(tile, xn, c0, p.C, oy0 - p.pad, ox0 - p.pad, p.H, p.W, p.border);
        for (int idx = threadIdx.x; idx < K * TH * TW; idx += 256) {
            const int k = idx / (TH * TW), r2 = idx - k * (TH * TW);
            const int oy = oy0 + r2 / TW, ox = ox0 + (r2 - (r2 / TW) * TW);
            gyt[idx] = (oy < p.OH && ox < p.OW) ? p.gy[(((size_t)n * K + k) * p.OH + oy) * p.OW + ox] : 0.f;
        }
        __syncthreads();
        if (active) {
            for (int py = 0; py < TH; ++py) {
                const float4* row = reinterpret_cast<const float4*>(tile + (ch * LH + py + r) * LWP);
#pragma unroll 2
                for (int g4 = 0; g4 < TW / 4; ++g4) {
                    float v[NV * 4];
#pragma unroll
                    for (int q = 0; q < NV; ++q) {
                        const float4 t = row[g4 + q];
                        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
                    }
                    float g[K][4];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const float4 t = *reinterpret_cast<const float4*>(gyt + k * (TH * TW) + py * TW + 4 * g4);   // broadcast
                        g[k][0] = t.x; g[k][1] = t.y; g[k][2] = t.z; g[k][3] = t.w;
                    }
#pragma unroll
                    for (int s = 0; s < R; ++s)
#pragma unroll
                        for (int k = 0; k < K; ++k)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[s][k] = fmaf(g[k][j], v[j + s], acc[s][k]);
                }
            }
        }
    }
    if (active) {
        const int CRS = p.C * R * R;
#pragma unroll
        for (int s = 0; s < R; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const size_t o = (size_t)k * CRS + (size_t)(c0 + ch) * R * R + r * R + s;
                if (p.part) p.part[(size_t)blockIdx.x * ((size_t)K * CRS) + o] = acc[s][k];
                else atomicAdd(p.gw + o, acc[s][k]);
            }
    }
}

template <int R>
void launch_fwd_r(const NarrowParams& p, int M, dim3 grid, hipStream_t st) {
    switch (M) {
        case 1: hipLaunchKernelGGL((narrow_fwd_kernel<1, R>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((narrow_fwd_kernel<2, R>), grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((narrow_fwd_kernel<3, R>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((narrow_fwd_kernel<4, R>), grid, dim3(256), 0, st, p); break;
    }
}
template <int R>
void launch_wgrad4_r(const NarrowParams& p, int K, dim3 grid, hipStream_t st) {
    switch (K) {
        case 1: hipLaunchKernelGGL((narrow_wgrad4_kernel<1, R>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((narrow_wgrad4_kernel<2, R>), grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((narrow_wgrad4_kernel<3, R>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((narrow_wgrad4_kernel<4, R>), grid, dim3(256), 0, st, p); break;
    }
}
template <int R>
void launch_fwd4_r(const NarrowParams& p, int M, dim3 grid, hipStream_t st) {
    switch (M) {
        case 1: hipLaunchKernelGGL((narrow_fwd4_kernel<1, R>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((narrow_fwd4_kernel<2, R>), grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((narrow_fwd4_kernel<3, R>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((narrow_fwd4_kernel<4, R>), grid, dim3(256), 0, st, p); break;
    }
}
template <int R>
void launch_wgrad_r(const NarrowParams& p, int K, dim3 grid, hipStream_t st) {
    switch (K) {
        case 1: hipLaunchKernelGGL((narrow_wgrad_kernel<1, R>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((narrow_wgrad_kernel<2, R>), grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((narrow_wgrad_kernel<3, R>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((narrow_wgrad_kernel<4, R>), grid, dim3(256), 0, st, p); break;
    }
}

}  // namespace

// Eligibility shared with conv.hip's entry points (declared there as extern).
bool nemar_narrow_eligible(int K, int C1, int R, int S, int stride, int N, int OH, int OW) {
    return K >= 1 && K <= 4 && C1 == 0 && R == S && (R == 3 || R == 4 || R == 7) && stride == 1 && N <= 65535 &&
           nemar_cdiv(OH, TH) <= 65535;
}

void nemar_sum_partials(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate,
                        hipStream_t st);
NEMAR_SWITCH(int, g_narrow_fwd4, 1);      // nemar_tune(19): register-tiled narrow forward for wide images (1, default) / one pixel per lane (0)

// channel ranges of the forward split mode: few output tiles and many channels (no activation) — so the launch fills the chip
static int narrow_fwd_csplit(int N, int C, int OH, int OW, int act) {
    const int tiles_total = nemar_cdiv(OW, TW) * nemar_cdiv(OH, TH) * N;
    const int chunks = nemar_cdiv(C, CH);
    int csplit = 1;
    if (act == 0 && tiles_total < 512 && chunks > 1) {
        csplit = nemar_cdiv(1024, tiles_total);
        if (csplit > chunks) csplit = chunks;
    }
    return csplit;
}

// `part` / `part_floats`: slab space of the channel-split mode (nullptr: no split); the split count is capped to what fits
int nemar_narrow_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int K, int R,
                     int pad, int border, int act, float slope, float* part, size_t part_floats, hipStream_t st) {
    NarrowParams p;
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.gy = nullptr; p.gw = nullptr; p.part = part;
    p.N = N; p.C = C; p.H = H; p.W = W; p.OH = H + 2 * pad - R + 1; p.OW = W + 2 * pad - R + 1;
    p.pad = pad; p.border = border; p.act = act; p.slope = slope;
    p.tiles_x = nemar_cdiv(p.OW, TW); p.tiles_y = nemar_cdiv(p.OH, TH); p.tiles_total = p.tiles_x * p.tiles_y * N;
    const size_t out_floats = (size_t)N * K * p.OH * p.OW;
    p.csplit = part ? narrow_fwd_csplit(N, C, p.OH, p.OW, act) : 1;
    if ((size_t)p.csplit * out_floats > part_floats) p.csplit = (int)(part_floats / out_floats);
    if (p.csplit < 1) p.csplit = 1;
    if (p.csplit > 1) {
        // ranges are whole CH-channel chunks: drop ranges that would be empty (every slab must be written)
        const int cper = nemar_cdiv(nemar_cdiv(C, p.csplit), CH) * CH;
        p.csplit = nemar_cdiv(C, cper);
    }
    if (p.OW >= 64 && g_narrow_fwd4) {           // wide images: register-tiled kernel, 128 x 8 output tiles
        dim3 grid4(nemar_cdiv(p.OW, T2W), nemar_cdiv(p.OH, T2H), N * p.csplit);
        if (R == 3) launch_fwd4_r<3>(p, K, grid4, st);
        else if (R == 4) launch_fwd4_r<4>(p, K, grid4, st);
        else launch_fwd4_r<7>(p, K, grid4, st);
    } else {
        dim3 grid(p.tiles_x, p.tiles_y, N * p.csplit);
        if (R == 3) launch_fwd_r<3>(p, K, grid, st);
        else if (R == 4) launch_fwd_r<4>(p, K, grid, st);
        else launch_fwd_r<7>(p, K, grid, st);
    }
    if (p.csplit > 1) nemar_sum_partials(part, (long long)out_floats, p.csplit, y, (long long)out_floats, false, st);
    return 0;
}

// workgroups along x of the narrow weight gradient = number of slabs of its fixed-order reduction
static bool narrow_wgrad_tiled(int C, int OW) { return g_narrow_fwd4 && OW >= 32 && C >= 16; }

int nemar_narrow_wgrad_splits(int N, int C, int OH, int OW) {
    const int tiles_total = nemar_cdiv(OW, TW) * nemar_cdiv(OH, TH) * N;
    const int chunks = nemar_cdiv(C, narrow_wgrad_tiled(C, OW) ? WCH : CH);
    int gx = nemar_cdiv(1024, chunks);           // ~4 workgroups per CU in total
    if (gx > tiles_total) gx = tiles_total;
    return gx < 1 ? 1 : gx;
}

int nemar_narrow_wgrad(const float* x, const float* gy, float* gw, int N, int C, int H, int W, int K, int R, int pad,
                       int border, float* part, hipStream_t st) {
    NarrowParams p;
    p.x = x; p.w = nullptr; p.bias = nullptr; p.y = nullptr; p.gy = gy; p.gw = gw; p.part = part;
    p.N = N; p.C = C; p.H = H; p.W = W; p.OH = H + 2 * pad - R + 1; p.OW = W + 2 * pad - R + 1;
    p.pad = pad; p.border = border; p.act = 0; p.slope = 0.f; p.csplit = 1;
    p.tiles_x = nemar_cdiv(p.OW, TW); p.tiles_y = nemar_cdiv(p.OH, TH); p.tiles_total = p.tiles_x * p.tiles_y * N;
    const bool tiled = narrow_wgrad_tiled(C, p.OW);
    const int chunks = nemar_cdiv(C, tiled ? WCH : CH);
    const int gx = nemar_narrow_wgrad_splits(N, C, p.OH, p.OW);
    dim3 grid(gx, chunks);
    if (tiled) {
        if (R == 3) launch_wgrad4_r<3>(p, K, grid, st);
        else if (R == 4) launch_wgrad4_r<4>(p, K, grid, st);
        else launch_wgrad4_r<7>(p, K, grid, st);
    } else if (R == 3) launch_wgrad_r<3>(p, K, grid, st);
    else if (R == 4) launch_wgrad_r<4>(p, K, grid, st);
    else launch_wgrad_r<7>(p, K, grid, st);
    if (part) {
        const long long KJ = (long long)K * C * R * R;
        nemar_sum_partials(part, KJ, gx, gw, KJ, true, st);
    }
    return 0;
}
