// K9 + K10 + K11: sampling-grid generation fused into bilinear grid_sample.
//
// Replaces, for the NeMAR hot path, the torch call sites
//   F.grid_sample(img, grid, 'bilinear', 'zeros', align_corners=False)   reference models/stn/unet_stn.py:173-174,
//                                                                          models/stn/affine_stn.py:129-130
//   identity_grid.repeat(B) + deformation, .permute(0,2,3,1)             reference models/stn/unet_stn.py:121-129,167
//   F.affine_grid(theta.view(-1,2,3), size)                              reference models/stn/affine_stn.py:105,128
// The grid is never materialised: it is synthesised in registers from the planar offset field
// (UNet STN) or from the six affine parameters (affine STN).
//
// HBM-bound: compulsory traffic is 4*(2C+2) B/px forward, 4*(3C+4) B/px backward with grad_input
// (4*(2C+4) without).  One lane owns VEC consecutive output pixels of one row and all C channels, so
// offset loads and output stores are 16 B/lane coalesced when Wo % 4 == 0; the four-corner gathers hit
// L1/L2 (each source texel is touched by ~4 neighbouring lanes in the near-identity regime).
#include "common.h"

namespace {

constexpr int GRID_EXPLICIT = 0;  // grid [N,Ho,Wo,2]  (x,y) interleaved, normalised coords
constexpr int GRID_UNET = 1;      // offsets [N,2,Ho,Wo] planar; + linspace(-1,1) identity (ch0 = x)
constexpr int GRID_AFFINE = 2;    // dtheta [N,6]; theta = dtheta + [1,0,0,0,1,0]; affine_grid(align_corners=False)

// torch.linspace(-1, 1, n)[i] in fp32: fused multiply-add from the nearer end (ATen RangeFactories
// symmetric form; bit-exact against torch CPU, see tests/test_oracle_torch.py)
__device__ __forceinline__ float linspace_m1_p1(int i, int n) {
    if (n <= 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return (i < n / 2) ? fmaf(step, (float)i, -1.f) : fmaf(-step, (float)(n - 1 - i), 1.f);
}
// affine_grid base coordinate, align_corners=False: (2i+1)/n - 1
__device__ __forceinline__ float affine_base(int i, int n) { return (2.f * (float)i + 1.f) / (float)n - 1.f; }

struct Sample {
    int x0, y0;
    float tx, ty;  // ix - x0, iy - y0
};
__device__ __forceinline__ Sample locate(float gx, float gy, int W, int H) {
    // unnormalise, align_corners=False: ((g + 1) * size - 1) / 2
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    Sample s;
    // clamp far-out-of-range coordinates before the int conversion (all four corners are OOB anyway)
    s.x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f);
    s.y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
    s.tx = ix - fx;
    s.ty = iy - fy;
    return s;
}

template <int MODE>
__device__ __forceinline__ void make_grid(const float* __restrict__ gsrc, int n, int h, int w, int Ho, int Wo,
                                          const float* th, float& gx, float& gy) {
    if (MODE == GRID_EXPLICIT) {
        const float2 g = *reinterpret_cast<const float2*>(gsrc + (((size_t)n * Ho + h) * Wo + w) * 2);
        gx = g.x;
        gy = g.y;
    } else if (MODE == GRID_UNET) {
        const size_t plane = (size_t)Ho * Wo;
        const size_t o = (size_t)n * 2 * plane + (size_t)h * Wo + w;
        gx = linspace_m1_p1(w, Wo) + gsrc[o];
        gy = linspace_m1_p1(h, Ho) + gsrc[o + plane];
    } else {
        const float xb = affine_base(w, Wo), yb = affine_base(h, Ho);
        gx = th[0] * xb + th[1] * yb + th[2];
        gy = th[3] * xb + th[4] * yb + th[5];
    }
}

// ---- forward ---------------------------------------------------------------------------------------
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void grid_sample_fwd_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ gsrc,
                                                              float* __restrict__ out, int C, int H, int W, int Ho,
                                                              int Wo) {
    const int n = blockIdx.y;
    const int wq = Wo / VEC;          // work items per row
    const int items = Ho * wq;
    float th[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (MODE == GRID_AFFINE) {
#pragma unroll
        for (int i = 0; i < 6; ++i) th[i] = gsrc[n * 6 + i] + ((i == 0 || i == 4) ? 1.f : 0.f);
    }
    const float* inN = in + (size_t)n * C * H * W;
    float* outN = out + (size_t)n * C * Ho * Wo;
    const size_t iplane = (size_t)H * W, oplane = (size_t)Ho * Wo;

    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int h = it / wq;
        const int w0 = (it - h * wq) * VEC;
        float gx[VEC], gy[VEC];
        if (MODE == GRID_UNET && VEC == 4) {
            const size_t o = (size_t)n * 2 * oplane + (size_t)h * Wo + w0;
            const float4 dx = *reinterpret_cast<const float4*>(gsrc + o);
            const float4 dy = *reinterpret_cast<const float4*>(gsrc + o + oplane);
            const float yb = linspace_m1_p1(h, Ho);
            gx[0] = linspace_m1_p1(w0 + 0, Wo) + dx.x; gy[0] = yb + dy.x;
            gx[1] = linspace_m1_p1(w0 + 1, Wo) + dx.y; gy[1] = yb + dy.y;
            gx[2] = linspace_m1_p1(w0 + 2, Wo) + dx.z; gy[2] = yb + dy.z;
            gx[3] = linspace_m1_p1(w0 + 3, Wo) + dx.w; gy[3] = yb + dy.w;
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) make_grid<MODE>(gsrc, n, h, w0 + v, Ho, Wo, th, gx[v], gy[v]);
        }
        Sample s[VEC];
        float wnw[VEC], wne[VEC], wsw[VEC], wse[VEC];
        int onw[VEC];
        bool vx0[VEC], vx1[VEC], vy0[VEC], vy1[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            s[v] = locate(gx[v], gy[v], W, H);
            const float ex = 1.f - s[v].tx, ey = 1.f - s[v].ty;
            wnw[v] = ex * ey; wne[v] = s[v].tx * ey; wsw[v] = ex * s[v].ty; wse[v] = s[v].tx * s[v].ty;
            vx0[v] = (unsigned)s[v].x0 < (unsigned)W;
            vx1[v] = (unsigned)(s[v].x0 + 1) < (unsigned)W;
            vy0[v] = (unsigned)s[v].y0 < (unsigned)H;
            vy1[v] = (unsigned)(s[v].y0 + 1) < (unsigned)H;
            onw[v] = s[v].y0 * W + s[v].x0;
        }
        for (int c = 0; c < C; ++c) {
            const float* p = inN + (size_t)c * iplane;
            float r[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float a = (vx0[v] && vy0[v]) ? p[onw[v]] : 0.f;
                const float b = (vx1[v] && vy0[v]) ? p[onw[v] + 1] : 0.f;
                const float cc = (vx0[v] && vy1[v]) ? p[onw[v] + W] : 0.f;
                const float d = (vx1[v] && vy1[v]) ? p[onw[v] + W + 1] : 0.f;
                r[v] = a * wnw[v] + b * wne[v] + cc * wsw[v] + d * wse[v];
            }
            float* q = outN + (size_t)c * oplane + (size_t)h * Wo + w0;
            if (VEC == 4) {
                *reinterpret_cast<float4*>(q) = make_float4(r[0], r[1], r[2], r[3]);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) q[v] = r[v];
            }
        }
    }
}

// ---- backward --------------------------------------------------------------------------------------
// grad_input: scatter-add of w_k * gout into the four corners (global fp32 atomics; the buffer is
// zero-filled by the entry point unless accumulating).  grad_grid: analytic, same pass.
//   UNET    -> ggrid planar [N,2,Ho,Wo] (= d loss / d offsets), written or accumulated
//   EXPLICIT-> ggrid [N,Ho,Wo,2]
//   AFFINE  -> gtheta [N,6] = sum_pix ggrid . [x_j, y_i, 1], block reduction + 6 atomics per block
template <int MODE, bool NEED_GIN>
__global__ __launch_bounds__(256) void grid_sample_bwd_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ gsrc,
                                                              const float* __restrict__ gout,
                                                              float* __restrict__ gin, float* __restrict__ ggrid,
                                                              int accum_ggrid, int C, int H, int W, int Ho, int Wo) {
    __shared__ float red[16];
    const int n = blockIdx.y;
    const int items = Ho * Wo;
    float th[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (MODE == GRID_AFFINE) {
#pragma unroll
        for (int i = 0; i < 6; ++i) th[i] = gsrc[n * 6 + i] + ((i == 0 || i == 4) ? 1.f : 0.f);
    }
    const float* inN = in + (size_t)n * C * H * W;
    float* ginN = NEED_GIN ? gin + (size_t)n * C * H * W : nullptr;
    const float* goN = gout + (size_t)n * C * Ho * Wo;
    const size_t iplane = (size_t)H * W, oplane = (size_t)Ho * Wo;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int h = it / Wo;
        const int w = it - h * Wo;
        float gx, gy;
        make_grid<MODE>(gsrc, n, h, w, Ho, Wo, th, gx, gy);
        const Sample s = locate(gx, gy, W, H);
        const float ex = 1.f - s.tx, ey = 1.f - s.ty;
        const float wnw = ex * ey, wne = s.tx * ey, wsw = ex * s.ty, wse = s.tx * s.ty;
        const bool x0 = (unsigned)s.x0 < (unsigned)W, x1 = (unsigned)(s.x0 + 1) < (unsigned)W;
        const bool y0 = (unsigned)s.y0 < (unsigned)H, y1 = (unsigned)(s.y0 + 1) < (unsigned)H;
        const int o = s.y0 * W + s.x0;
        float gix = 0.f, giy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float g = goN[(size_t)c * oplane + it];
            const float* p = inN + (size_t)c * iplane;
            const float a = (x0 && y0) ? p[o] : 0.f;
            const float b = (x1 && y0) ? p[o + 1] : 0.f;
            const float cc = (x0 && y1) ? p[o + W] : 0.f;
            const float d = (x1 && y1) ? p[o + W + 1] : 0.f;
            gix += g * ((b - a) * ey + (d - cc) * s.ty);
            giy += g * ((cc - a) * ex + (d - b) * s.tx);
            if (NEED_GIN) {
                float* q = ginN + (size_t)c * iplane;
                if (x0 && y0) atomicAdd(q + o, g * wnw);
                if (x1 && y0) atomicAdd(q + o + 1, g * wne);
                if (x0 && y1) atomicAdd(q + o + W, g * wsw);
                if (x1 && y1) atomicAdd(q + o + W + 1, g * wse);
            }
        }
        const float ggx = gix * (0.5f * (float)W), ggy = giy * (0.5f * (float)H);
        if (MODE == GRID_UNET) {
            float* q = ggrid + (size_t)n * 2 * oplane + it;
            if (accum_ggrid) { q[0] += ggx; q[oplane] += ggy; } else { q[0] = ggx; q[oplane] = ggy; }
        } else if (MODE == GRID_EXPLICIT) {
            float2* q = reinterpret_cast<float2*>(ggrid + ((size_t)n * oplane + it) * 2);
            if (accum_ggrid) { float2 t = *q; t.x += ggx; t.y += ggy; *q = t; } else { *q = make_float2(ggx, ggy); }
        } else {
            const float xb = affine_base(w, Wo), yb = affine_base(h, Ho);
            acc[0] += ggx * xb; acc[1] += ggx * yb; acc[2] += ggx;
            acc[3] += ggy * xb; acc[4] += ggy * yb; acc[5] += ggy;
        }
    }
    if (MODE == GRID_AFFINE) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float t = block_sum(acc[i], red);
            if (threadIdx.x == 0) atomicAdd(ggrid + n * 6 + i, t);
        }
    }
}

// ---- backward, grad_input accumulated through an LDS tile ------------------------------------------------------------
// In the NeMAR regime (identity + small offsets / near-identity affine) the four corners an output pixel scatters into
// lie next to the pixel itself.  A workgroup owns a 16 x 64 output tile (2 pixels per lane), accumulates the scatter of
// up to 4 channels at a time in LDS images of the input region around it (tile + 4 texels of halo, ds_add_f32), and flushes
// the region to grad_input with one coalesced row-wise atomic per non-zero texel: 12 scattered global atomics per pixel
// become <= 1.7 coalesced ones per texel.  Corners that fall outside the region (large deformations) go straight to
// global atomics, so the result never depends on the regime.  Same-size input/output only (the training path).
constexpr int TL_W = 64, TL_H = 16, TL_HALO = 4, TL_CH = 4;   // TL_CH channels share one zero / flush round
constexpr int TL_THREADS = 512, TL_ROWS = TL_THREADS / 64, TL_PPT = TL_H / TL_ROWS;   // 2 pixels per lane
constexpr int TL_RW = TL_W + 2 * TL_HALO, TL_RH = TL_H + 2 * TL_HALO;
template <int MODE>
__global__ __launch_bounds__(TL_THREADS) void grid_sample_bwd_tiled_kernel(const float* __restrict__ in,
                                                                    const float* __restrict__ gsrc,
                                                                    const float* __restrict__ gout,
                                                                    float* __restrict__ gin, float* __restrict__ ggrid,
                                                                    int accum_ggrid, int C, int H, int W, int ablate) {
    __shared__ float red[16];
    __shared__ float tile[TL_CH * TL_RH * TL_RW];
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * TL_W, y0 = blockIdx.y * TL_H;
    const int rx0 = x0 - TL_HALO, ry0 = y0 - TL_HALO;          // region origin in input coordinates
    float th[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (MODE == GRID_AFFINE) {
#pragma unroll
        for (int i = 0; i < 6; ++i) th[i] = gsrc[n * 6 + i] + ((i == 0 || i == 4) ? 1.f : 0.f);
    }
    const size_t plane = (size_t)H * W;
    const float* inN = in + (size_t)n * C * plane;
    float* ginN = gin + (size_t)n * C * plane;
    const float* goN = gout + (size_t)n * C * plane;
    const int lx = threadIdx.x & 63, lrow = threadIdx.x >> 6;   // lane column, first of its TL_PPT rows (stride TL_ROWS)
    const int w = x0 + lx;

    // per-pixel sample geometry (kept in registers across the channel loop)
    int o[TL_PPT], li[TL_PPT];
    float wnw[TL_PPT], wne[TL_PPT], wsw[TL_PPT], wse[TL_PPT], tx[TL_PPT], ty[TL_PPT], gix[TL_PPT], giy[TL_PPT];
    unsigned msk[TL_PPT];     // bit0..3: corner in the image; bit4: pixel valid; bit5: whole 2x2 patch inside the LDS region
#pragma unroll
    for (int i = 0; i < TL_PPT; ++i) {
        const int h = y0 + lrow + TL_ROWS * i;
        msk[i] = 0u; o[i] = 0; li[i] = 0; gix[i] = 0.f; giy[i] = 0.f;
        wnw[i] = wne[i] = wsw[i] = wse[i] = tx[i] = ty[i] = 0.f;
        if (h < H && w < W) {
            float gx, gy;
            make_grid<MODE>(gsrc, n, h, w, H, W, th, gx, gy);
            const Sample s = locate(gx, gy, W, H);
            const float ex = 1.f - s.tx, ey = 1.f - s.ty;
            wnw[i] = ex * ey; wne[i] = s.tx * ey; wsw[i] = ex * s.ty; wse[i] = s.tx * s.ty;
            tx[i] = s.tx; ty[i] = s.ty;
            const bool bx0 = (unsigned)s.x0 < (unsigned)W, bx1 = (unsigned)(s.x0 + 1) < (unsigned)W;
            const bool by0 = (unsigned)s.y0 < (unsigned)H, by1 = (unsigned)(s.y0 + 1) < (unsigned)H;
            o[i] = s.y0 * W + s.x0;
            const int px = s.x0 - rx0, py = s.y0 - ry0;
            const bool inreg = px >= 0 && px + 1 < TL_RW && py >= 0 && py + 1 < TL_RH;
            li[i] = py * TL_RW + px;
            msk[i] = (bx0 && by0 ? 1u : 0u) | (bx1 && by0 ? 2u : 0u) | (bx0 && by1 ? 4u : 0u) | (bx1 && by1 ? 8u : 0u) |
                     16u | (inreg ? 32u : 0u);
        }
    }
    for (int c0 = 0; c0 < C; c0 += TL_CH) {
        const int nc = min(TL_CH, C - c0);
        for (int t = threadIdx.x; t < nc * (TL_RH * TL_RW); t += TL_THREADS) tile[t] = 0.f;
        __syncthreads();
        for (int cc_ = 0; cc_ < nc; ++cc_) {
            const int c = c0 + cc_;
            const float* p = inN + (size_t)c * plane;
            float* q = ginN + (size_t)c * plane;
            float* tl = tile + cc_ * (TL_RH * TL_RW);
#pragma unroll
            for (int i = 0; i < TL_PPT; ++i) {
                if (!(msk[i] & 16u)) continue;
                const int h = y0 + lrow + TL_ROWS * i;
                const float g = goN[(size_t)c * plane + (size_t)h * W + w];
                const float a = (msk[i] & 1u) ? p[o[i]] : 0.f;
                const float b = (msk[i] & 2u) ? p[o[i] + 1] : 0.f;
                const float cc = (msk[i] & 4u) ? p[o[i] + W] : 0.f;
                const float d = (msk[i] & 8u) ? p[o[i] + W + 1] : 0.f;
                gix[i] += g * ((b - a) * (1.f - ty[i]) + (d - cc) * ty[i]);
                giy[i] += g * ((cc - a) * (1.f - tx[i]) + (d - b) * tx[i]);
                if (ablate & 2) continue;
                if (msk[i] & 32u) {
                    if (msk[i] & 1u) atomicAdd(&tl[li[i]], g * wnw[i]);
                    if (msk[i] & 2u) atomicAdd(&tl[li[i] + 1], g * wne[i]);
                    if (msk[i] & 4u) atomicAdd(&tl[li[i] + TL_RW], g * wsw[i]);
                    if (msk[i] & 8u) atomicAdd(&tl[li[i] + TL_RW + 1], g * wse[i]);
                } else {
                    if (msk[i] & 1u) atomicAdd(q + o[i], g * wnw[i]);
                    if (msk[i] & 2u) atomicAdd(q + o[i] + 1, g * wne[i]);
                    if (msk[i] & 4u) atomicAdd(q + o[i] + W, g * wsw[i]);
                    if (msk[i] & 8u) atomicAdd(q + o[i] + W + 1, g * wse[i]);
                }
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < nc * (TL_RH * TL_RW); t += TL_THREADS) {
            const float v = tile[t];
            if (v != 0.f && !(ablate & 1)) {
                const int cc_ = t / (TL_RH * TL_RW), r = t - cc_ * (TL_RH * TL_RW);
                const int ry = r / TL_RW, rx = r - ry * TL_RW;
                const int iy = ry0 + ry, ix = rx0 + rx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    atomicAdd(ginN + (size_t)(c0 + cc_) * plane + (size_t)iy * W + ix, v);
            }
        }
        __syncthreads();
    }
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TL_PPT; ++i) {
        if (!(msk[i] & 16u)) continue;
        const int h = y0 + lrow + TL_ROWS * i;
        const size_t it = (size_t)h * W + w;
        const float ggx = gix[i] * (0.5f * (float)W), ggy = giy[i] * (0.5f * (float)H);
        if (MODE == GRID_UNET) {
            float* qq = ggrid + (size_t)n * 2 * plane + it;
            if (accum_ggrid) { qq[0] += ggx; qq[plane] += ggy; } else { qq[0] = ggx; qq[plane] = ggy; }
        } else if (MODE == GRID_EXPLICIT) {
            float2* qq = reinterpret_cast<float2*>(ggrid + ((size_t)n * plane + it) * 2);
            if (accum_ggrid) { float2 t2 = *qq; t2.x += ggx; t2.y += ggy; *qq = t2; } else { *qq = make_float2(ggx, ggy); }
        } else {
            const float xb = affine_base(w, W), yb = affine_base(h, H);
            acc[0] += ggx * xb; acc[1] += ggx * yb; acc[2] += ggx;
            acc[3] += ggy * xb; acc[4] += ggy * yb; acc[5] += ggy;
        }
    }
    if (MODE == GRID_AFFINE) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float t = block_sum(acc[i], red);
            if (threadIdx.x == 0) atomicAdd(ggrid + n * 6 + i, t);
        }
    }
}

template <int MODE>
int launch_fwd(const float* in, const float* gsrc, float* out, int N, int C, int H, int W, int Ho, int Wo,
               hipStream_t st) {
    const bool vec4 = (Wo % 4 == 0) && (((uintptr_t)out & 15) == 0) && (((uintptr_t)gsrc & 15) == 0) &&
                      (MODE != GRID_EXPLICIT);
    const long long items = (long long)Ho * (vec4 ? Wo / 4 : Wo);
    int gx = nemar_cdiv(items, 256);
    const int cap = nemar_cdiv(256 * 8, N);
    if (gx > cap) gx = cap;
    dim3 grid(gx, N), block(256);
    if (vec4)
        hipLaunchKernelGGL((grid_sample_fwd_kernel<MODE, 4>), grid, block, 0, st, in, gsrc, out, C, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL((grid_sample_fwd_kernel<MODE, 1>), grid, block, 0, st, in, gsrc, out, C, H, W, Ho, Wo);
    return 0;
}

// nemar_grid_sample_tune: 0 (default) = grad_input by global fp32 atomics, 1 = through the LDS tile.  Measured (8x3x256^2):
// zero / near-identity field 26 vs 43 us, smooth 3-pixel field 64 vs 45 us, white 1-pixel field 139 vs 44 us (1024^2:
// 385 / 1013 / 2175 vs 533 / 549 / 547 us).  The tile version is bound by ds_add_f32 (~180 cycles per wave-instruction,
// ablation: 29 of its 43 us), so the global-atomic version stays the default for the regime the training step starts
// in; the tile version is the robust choice once deformations are rough or images large.
int g_tiled_scatter = 0;

template <int MODE>
int launch_bwd(const float* in, const float* gsrc, const float* gout, float* gin, float* ggrid, int accum_ggrid,
               int N, int C, int H, int W, int Ho, int Wo, hipStream_t st) {
    const long long items = (long long)Ho * Wo;
    int gx = nemar_cdiv(items, 256);
    const int cap = nemar_cdiv(256 * 8, N);
    if (gx > cap) gx = cap;
    dim3 grid(gx, N), block(256);
    if (gin && H == Ho && W == Wo && g_tiled_scatter && N <= 65535)
        hipLaunchKernelGGL((grid_sample_bwd_tiled_kernel<MODE>), dim3(nemar_cdiv(W, TL_W), nemar_cdiv(H, TL_H), N),
                           dim3(TL_THREADS), 0, st, in, gsrc, gout, gin, ggrid, accum_ggrid, C, H, W, g_tiled_scatter >> 1);
    else if (gin)
        hipLaunchKernelGGL((grid_sample_bwd_kernel<MODE, true>), grid, block, 0, st, in, gsrc, gout, gin, ggrid,
                           accum_ggrid, C, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL((grid_sample_bwd_kernel<MODE, false>), grid, block, 0, st, in, gsrc, gout, gin, ggrid,
                           accum_ggrid, C, H, W, Ho, Wo);
    return 0;
}

}  // namespace

// grad_input scatter variant: 0 = global atomics (default), 1 = LDS tile; bits 1..2 of larger values = ablations
NEMAR_API int nemar_grid_sample_tune(int tiled_scatter) {
    g_tiled_scatter = tiled_scatter;
    return NEMAR_OK;
}

NEMAR_API int nemar_grid_sample_fwd(const float* in, const float* grid_src, int grid_mode, float* out, int N, int C,
                                    int H, int W, int Ho, int Wo, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(in && grid_src && out, "grid_sample_fwd: null pointer");
    NEMAR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "grid_sample_fwd: bad shape");
    NEMAR_REQUIRE((long long)H * W < (1ll << 31) && (long long)Ho * Wo < (1ll << 31) && N <= 65535,
                  "grid_sample_fwd: plane too large");
    hipStream_t st = (hipStream_t)stream;
    switch (grid_mode) {
        case GRID_EXPLICIT: launch_fwd<GRID_EXPLICIT>(in, grid_src, out, N, C, H, W, Ho, Wo, st); break;
        case GRID_UNET: launch_fwd<GRID_UNET>(in, grid_src, out, N, C, H, W, Ho, Wo, st); break;
        case GRID_AFFINE: launch_fwd<GRID_AFFINE>(in, grid_src, out, N, C, H, W, Ho, Wo, st); break;
        default: NEMAR_REQUIRE(false, "grid_sample_fwd: unknown grid_mode %d", grid_mode);
    }
    NEMAR_CHECK_LAUNCH("grid_sample_fwd");
    return NEMAR_OK;
}

NEMAR_API int nemar_grid_sample_bwd(const float* in, const float* grid_src, int grid_mode, const float* gout,
                                    float* gin, int accum_gin, float* ggrid, int accum_ggrid, int N, int C, int H,
                                    int W, int Ho, int Wo, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(in && grid_src && gout && ggrid, "grid_sample_bwd: null pointer");
    NEMAR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "grid_sample_bwd: bad shape");
    NEMAR_REQUIRE((long long)H * W < (1ll << 31) && (long long)Ho * Wo < (1ll << 31) && N <= 65535,
                  "grid_sample_bwd: plane too large");
    hipStream_t st = (hipStream_t)stream;
    if (gin && !accum_gin) NEMAR_HIP_CALL(hipMemsetAsync(gin, 0, sizeof(float) * (size_t)N * C * H * W, st));
    if (grid_mode == GRID_AFFINE && !accum_ggrid)
        NEMAR_HIP_CALL(hipMemsetAsync(ggrid, 0, sizeof(float) * (size_t)N * 6, st));
    switch (grid_mode) {
        case GRID_EXPLICIT:
            launch_bwd<GRID_EXPLICIT>(in, grid_src, gout, gin, ggrid, accum_ggrid, N, C, H, W, Ho, Wo, st); break;
        case GRID_UNET:
            launch_bwd<GRID_UNET>(in, grid_src, gout, gin, ggrid, accum_ggrid, N, C, H, W, Ho, Wo, st); break;
        case GRID_AFFINE:
            launch_bwd<GRID_AFFINE>(in, grid_src, gout, gin, ggrid, accum_ggrid, N, C, H, W, Ho, Wo, st); break;
        default: NEMAR_REQUIRE(false, "grid_sample_bwd: unknown grid_mode %d", grid_mode);
    }
    NEMAR_CHECK_LAUNCH("grid_sample_bwd");
    return NEMAR_OK;
}
