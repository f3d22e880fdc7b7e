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
        int o00[VEC], o01[VEC], o10[VEC], o11[VEC];
        bool v00[VEC], v01[VEC], v10[VEC], v11[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            s[v] = locate(gx[v], gy[v], W, H);
            const float ex = 1.f - s[v].tx, ey = 1.f - s[v].ty;
            wnw[v] = ex * ey; wne[v] = s[v].tx * ey; wsw[v] = ex * s[v].ty; wse[v] = s[v].tx * s[v].ty;
            const bool vx0 = (unsigned)s[v].x0 < (unsigned)W, vx1 = (unsigned)(s[v].x0 + 1) < (unsigned)W;
            const bool vy0 = (unsigned)s[v].y0 < (unsigned)H, vy1 = (unsigned)(s[v].y0 + 1) < (unsigned)H;
            v00[v] = vx0 && vy0; v01[v] = vx1 && vy0; v10[v] = vx0 && vy1; v11[v] = vx1 && vy1;
            // corner addresses CLAMPED into the image: the loads below are unconditional (a conditional load makes hipcc branch around
            // it and wait for every load separately — 48 serialized gathers per thread), zeros are selected afterwards
            const int xa = min(max(s[v].x0, 0), W - 1), xb = min(max(s[v].x0 + 1, 0), W - 1);
            const int ya = min(max(s[v].y0, 0), H - 1), yb = min(max(s[v].y0 + 1, 0), H - 1);
            o00[v] = ya * W + xa; o01[v] = ya * W + xb; o10[v] = yb * W + xa; o11[v] = yb * W + xb;
        }
        for (int c = 0; c < C; ++c) {
            const float* p = inN + (size_t)c * iplane;
            float a[VEC], b[VEC], cc[VEC], d[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                a[v] = p[o00[v]];
                b[v] = p[o01[v]];
                cc[v] = p[o10[v]];
                d[v] = p[o11[v]];
            }
            float r[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                r[v] = (v00[v] ? a[v] : 0.f) * wnw[v] + (v01[v] ? b[v] : 0.f) * wne[v] + (v10[v] ? cc[v] : 0.f) * wsw[v] +
                       (v11[v] ? d[v] : 0.f) * wse[v];
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
                                                              int accum_ggrid, int C, int H, int W, int Ho, int Wo,
                                                              float* __restrict__ gpart) {
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
            if (threadIdx.x == 0) {
                if (gpart) gpart[((size_t)n * gridDim.x + blockIdx.x) * 6 + i] = t;    // summed in block order afterwards
                else atomicAdd(ggrid + n * 6 + i, t);
            }
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

// ---- backward, grad_input as a GATHER per destination tile (default; bitwise reproducible) -------------------------------------
// The scatter  gin[corner_k(p)] += w_k(p) * gout[p]  is inverted: a workgroup owns a 64 x 16 tile of grad_input and finds, for
// every texel of it, the output pixels that touch it.  An output pixel p = (h,w) whose 2x2 corner patch lies within +-GT_R of
// its own position ("near": identity, the reference's linspace zoom, smooth or noisy fields of a few pixels) can only touch
// texels within GT_R of (h,w), so the workgroup stages the sample geometry (corner base, fractional weights) and the gout
// values of the (64+2R) x (16+2R) pixels around its tile in LDS, and every texel scans the (2R+1)^2 pixels around it in a
// fixed order: no atomics at all, one coalesced store per texel, a summation order that never changes.
// The same pass computes d loss / d grid for the tile's own pixels (they are staged anyway).
// "Far" pixels (patch further than GT_R from the pixel: rough / large deformations) are appended to a list and scattered
// by far_scatter_kernel with 64-bit FIXED-POINT atomics — integer addition is associative, so that path is reproducible as
// well: contributions are scaled by 2^(40 - e), e = exponent of max |gout| (found by the first pass), i.e. 40 bits below the
// largest gradient, and far_fold_kernel adds the converted sums to grad_input and returns the accumulator to all-zero.
// Round 3: the window FOLLOWS the field.  A smooth deformation of many pixels is locally a translation, so every destination tile T
// gets an integer offset o_T = the displacement of the pixels that land in it (tile_offset_kernel: two fixed-point steps of
// o <- d(centre(T) - o)), stages the pixels of (T - o_T) +- GT_R instead of T +- GT_R, and a texel scans the window around t - o_T.
// "Near" becomes a property of a (pixel, corner) pair — corner k of pixel p is gathered iff d(p) - o_T(k) lies in the window, T(k) the
// tile of that corner's texel — and only the other corners go through the fixed-point scatter (4-bit mask next to the pixel id in the
// far list).  With o = 0 everywhere this is exactly the round-2 scheme; a smooth 3-pixel field at 1024^2 went from 872 us (most
// pixels far) to the identity regime's cost.
constexpr int GT_W = 64, GT_H = 16, GT_R = 3, GT_RW = GT_W + 2 * GT_R, GT_RH = GT_H + 2 * GT_R, GT_NP = GT_RW * GT_RH;
constexpr int GT_CH = 4;
constexpr int GT_KEY_NONE = 0x7fff7fff;
constexpr int FIX_BITS = 40;

struct GatherWs {
    unsigned* count;              // [1]  != 0: some tile has far pixels      } zeroed by the entry point before every call
    unsigned* maxbits;            // [1]  bit pattern of max |gout|           }
    unsigned* wg_count;           // [N*tiles] far pixels of each tile (written by every tile of the gather pass)
    unsigned* far_list;           // [N*tiles][GT_W*GT_H] packed (n * H + h) * W + w of each tile's far pixels
    long long* acc;               // [N*C*H*W] fixed-point accumulator — ALL-ZERO between calls (far_fold_kernel restores it)
    unsigned* dirty;              // [N*tiles] tile touched by the far scatter — all-zero between calls
    float* gpart;                 // [N][tiles][6] per-workgroup sums of the affine grid gradient
    const int* toff;              // [N*tiles][2] gather-window offset (ox, oy) of every destination tile (tile_offset_kernel)
};

template <int MODE>
__global__ __launch_bounds__(256) void tile_offset_kernel(const float* __restrict__ gsrc, int* __restrict__ toff, int H, int W,
                                                          int tiles_x, int tiles_y, int N, unsigned* __restrict__ any_shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * tiles_x * tiles_y) return;
    const int n = i / (tiles_x * tiles_y), r = i - n * (tiles_x * tiles_y);
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    float th[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (MODE == GRID_AFFINE) {
#pragma unroll
        for (int k = 0; k < 6; ++k) th[k] = gsrc[n * 6 + k] + ((k == 0 || k == 4) ? 1.f : 0.f);
    }
    const int cx = min(tx * GT_W + GT_W / 2, W - 1), cy = min(ty * GT_H + GT_H / 2, H - 1);
    int ox = 0, oy = 0;
    for (int it = 0; it < 2; ++it) {
        const int px = min(max(cx - ox, 0), W - 1), py = min(max(cy - oy, 0), H - 1);
        float gx, gy;
        make_grid<MODE>(gsrc, n, py, px, H, W, th, gx, gy);
        const Sample s = locate(gx, gy, W, H);
        ox = min(max(s.x0 - px, -16384), 16384);
        oy = min(max(s.y0 - py, -16384), 16384);
    }
    // dead zone: the centred window already covers displacements of -GT_R .. GT_R - 1 pixels — identity, the linspace zoom and
    // pixel-level noise keep o = 0, and then the gather pass takes its table-free path (*any_shift stays 0)
    if (ox >= -1 && ox <= 0) ox = 0;
    if (oy >= -1 && oy <= 0) oy = 0;
    toff[2 * i] = ox;
    toff[2 * i + 1] = oy;
    if ((ox | oy) != 0 && *(volatile unsigned*)any_shift == 0u) *(volatile unsigned*)any_shift = 1u;
}

// corners of pixel (h, w) with corner base (x0, y0) that NO destination tile gathers: bit k = corner (x0 + (k & 1), y0 + (k >> 1))
__device__ __forceinline__ unsigned far_corner_mask(const int* __restrict__ toff, int tiles_x, int x0, int y0, int h, int w, int H,
                                                    int W, bool any_shift) {
    if (!any_shift) {           // every window is centred on its tile: all four corners share the round-2 test
        const int dx = x0 - w, dy = y0 - h;
        return (dx >= -GT_R && dx <= GT_R - 1 && dy >= -GT_R && dy <= GT_R - 1) ? 0u : 15u;
    }
    unsigned mask = 0u;
    int last_t = -1, lox = 0, loy = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
        if ((unsigned)cx >= (unsigned)W || (unsigned)cy >= (unsigned)H) continue;
        const int t = (cy / GT_H) * tiles_x + cx / GT_W;
        if (t != last_t) { last_t = t; lox = toff[2 * t]; loy = toff[2 * t + 1]; }
        const int dx = x0 - w - lox, dy = y0 - h - loy;
        if (!(dx >= -GT_R && dx <= GT_R - 1 && dy >= -GT_R && dy <= GT_R - 1)) mask |= 1u << k;
    }
    return mask;
}

// GG: the pass also computes d loss / d grid for the tile's own pixels (needs the 4-corner gathers of `in`); !GG: grid
// gradient left to grid_sample_bwd_kernel<MODE, false>, this pass only streams gsrc + gout and writes grad_input.
template <int MODE, bool GG, int GT_THREADS>
__global__ __launch_bounds__(GT_THREADS) void grid_sample_bwd_gather_kernel(const float* __restrict__ in,
                                                                    const float* __restrict__ gsrc,
                                                                    const float* __restrict__ gout,
                                                                    float* __restrict__ gin, int accum_gin,
                                                                    float* __restrict__ ggrid, int accum_ggrid, int C,
                                                                    int H, int W, GatherWs ws) {
    __shared__ int s_key[GT_NP];
    __shared__ float s_tx[GT_NP], s_ty[GT_NP];
    __shared__ float s_g[GT_CH][GT_NP];
    __shared__ float red[16];
    __shared__ unsigned s_far[GT_W * GT_H];
    __shared__ unsigned s_nfar;
    __shared__ unsigned s_rt;                          // round 6: the radius this tile's gathered pixels actually need (1 .. GT_R)
    constexpr int GT_TPT = GT_W * GT_H / GT_THREADS;   // vertically adjacent texels per thread in the gather phase
    const int n = blockIdx.z;
    const int tx0 = blockIdx.x * GT_W, ty0 = blockIdx.y * GT_H;
    const int tid = threadIdx.x;
    float th[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (MODE == GRID_AFFINE) {
#pragma unroll
        for (int i = 0; i < 6; ++i) th[i] = gsrc[n * 6 + i] + ((i == 0 || i == 4) ? 1.f : 0.f);
    }
    const size_t plane = (size_t)H * W;
    const float* inN = in + (size_t)n * C * plane;
    const float* goN = gout + (size_t)n * C * plane;
    if (tid == 0) { s_nfar = 0u; s_rt = 1u; }
    __syncthreads();
    const int tiles_x = gridDim.x, tiles_n = gridDim.x * gridDim.y;
    const int* const toffN = ws.toff + 2 * (size_t)n * tiles_n;                 // this image's offset table
    const bool any_shift = ws.count[2] != 0u;                                   // (uniform) some tile of the launch has o != 0
    const int ox = any_shift ? toffN[2 * (blockIdx.y * tiles_x + blockIdx.x)] : 0;
    const int oy = any_shift ? toffN[2 * (blockIdx.y * tiles_x + blockIdx.x) + 1] : 0;
    const bool shifted = (ox | oy) != 0;       // the staged region is (tile - o) +- GT_R: the tile's own pixels are not (all) in it
    const int rx0 = tx0 - ox - GT_R, ry0 = ty0 - oy - GT_R;                     // pixel at staged index (0, 0)
    float acc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gmax = 0.f;
    unsigned myrt = 1u;                                // the largest radius a pixel this thread staged needs (-> s_rt, one LDS atomic per wave)
    // Local arrays are only ever indexed by unrolled constants (a run-time channel index would put them in scratch memory).
    // ---- stage 1: the tile's own pixels (4 per thread, independent iterations: their loads overlap): geometry + gout -> LDS,
    //      d loss / d grid -> global, far pixels -> list ------------------------------------------------------------------------
    constexpr int OWN_PT = GT_W * GT_H / GT_THREADS;
#pragma unroll
    for (int i = 0; i < OWN_PT; ++i) {
        const int t = tid + i * GT_THREADS;
        const int ly = t >> 6, lx = t & 63;
        const int h = ty0 + ly, w = tx0 + lx;
        const int idx = (ly + GT_R) * GT_RW + lx + GT_R;
        int key = GT_KEY_NONE;
        float ftx = 0.f, fty = 0.f, g[GT_CH] = {0.f, 0.f, 0.f, 0.f};
        if (h < H && w < W) {
            float gx, gy;
            make_grid<MODE>(gsrc, n, h, w, H, W, th, gx, gy);
            const Sample s = locate(gx, gy, W, H);
            ftx = s.tx; fty = s.ty;
            const bool x0 = (unsigned)s.x0 < (unsigned)W, x1 = (unsigned)(s.x0 + 1) < (unsigned)W;
            const bool y0 = (unsigned)s.y0 < (unsigned)H, y1 = (unsigned)(s.y0 + 1) < (unsigned)H;
            const bool any = (x0 || x1) && (y0 || y1);
            const size_t it = (size_t)h * W + w;
            // corners no tile's window reaches -> this tile's far list (the scatter pass handles exactly those corners)
            if (any) {
                const unsigned fm = far_corner_mask(toffN, tiles_x, s.x0, s.y0, h, w, H, W, any_shift);
                if (fm) s_far[atomicAdd(&s_nfar, 1u)] = (unsigned)(((size_t)n * H + h) * W + w) | (fm << 28);
            }
            const bool near = s.x0 - w >= -GT_R && s.x0 - w <= GT_R - 1 && s.y0 - h >= -GT_R && s.y0 - h <= GT_R - 1;   // (o = 0 form)
            if (any && near) {
                key = ((s.y0 - ty0 + 2 * GT_R) << 16) | ((s.x0 - tx0 + 2 * GT_R) & 0xffff);
                // displacement (dx, dy) in [-r, r - 1]^2 needs radius r: identity, the linspace zoom and sub-pixel fields need 1
                const int ddx = s.x0 - w, ddy = s.y0 - h;
                const int rn = max(ddx >= 0 ? ddx + 1 : -ddx, ddy >= 0 ? ddy + 1 : -ddy);
                if (!shifted) myrt = max(myrt, (unsigned)rn);
            }
            // d loss / d grid (same arithmetic as grid_sample_bwd_kernel)
            const int o = s.y0 * W + s.x0;
            const float ex = 1.f - s.tx, ey = 1.f - s.ty;
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int c = 0; c < GT_CH; ++c) {
                if (c < C) {
                    g[c] = goN[(size_t)c * plane + it];
                    gmax = fmaxf(gmax, fabsf(g[c]));
                    if (GG) {
                        const float* pch = inN + (size_t)c * plane;
                        const float a = (x0 && y0) ? pch[o] : 0.f;
                        const float b = (x1 && y0) ? pch[o + 1] : 0.f;
                        const float cc = (x0 && y1) ? pch[o + W] : 0.f;
                        const float d = (x1 && y1) ? pch[o + W + 1] : 0.f;
                        gix += g[c] * ((b - a) * ey + (d - cc) * s.ty);
                        giy += g[c] * ((cc - a) * ex + (d - b) * s.tx);
                    }
                }
            }
            if (GG) {
                const float ggx = gix * (0.5f * (float)W), ggy = giy * (0.5f * (float)H);
                if (MODE == GRID_UNET) {
                    float* q = ggrid + (size_t)n * 2 * plane + it;
                    if (accum_ggrid) { q[0] += ggx; q[plane] += ggy; } else { q[0] = ggx; q[plane] = ggy; }
                } else if (MODE == GRID_EXPLICIT) {
                    float2* q = reinterpret_cast<float2*>(ggrid + ((size_t)n * plane + it) * 2);
                    if (accum_ggrid) { float2 t2 = *q; t2.x += ggx; t2.y += ggy; *q = t2; } else { *q = make_float2(ggx, ggy); }
                } else {
                    const float xb = affine_base(w, W), yb = affine_base(h, H);
                    acc6[0] += ggx * xb; acc6[1] += ggx * yb; acc6[2] += ggx;
                    acc6[3] += ggy * xb; acc6[4] += ggy * yb; acc6[5] += ggy;
                }
            }
        }
        if (GG && !shifted) {              // (an unshifted tile's own pixels ARE the centre of its staged region)
            s_key[idx] = key;
            s_tx[idx] = ftx;
            s_ty[idx] = fty;
#pragma unroll
            for (int c = 0; c < GT_CH; ++c) s_g[c][idx] = g[c];
        }
    }
    // ---- stage 2: the rest of the staged region: geometry + gout only.  Unshifted tile with the fused grid gradient: the halo ring
    //      (GT_R rows above / below, GT_R columns left / right of the tile); otherwise every pixel of the region ---------------------
    const bool whole = !GG || shifted;
    const int halo_n = whole ? GT_NP : GT_NP - GT_W * GT_H;
#pragma unroll 2
    for (int e = tid; e < halo_n; e += GT_THREADS) {
        int ry, rx;
        if (whole) {
            ry = e / GT_RW;
            rx = e - ry * GT_RW;
        } else if (e < 2 * GT_R * GT_RW) {
            const int r = e / GT_RW;
            rx = e - r * GT_RW;
            ry = r < GT_R ? r : GT_H + r;
        } else {
            const int q = e - 2 * GT_R * GT_RW;
            const int r = q / (2 * GT_R), c2 = q - r * (2 * GT_R);
            ry = GT_R + r;
            rx = c2 < GT_R ? c2 : GT_W + c2;
        }
        const int idx = ry * GT_RW + rx;
        const int h = ry0 + ry, w = rx0 + rx;
        int key = GT_KEY_NONE;
        float ftx = 0.f, fty = 0.f, g[GT_CH] = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
            float gx, gy;
            make_grid<MODE>(gsrc, n, h, w, H, W, th, gx, gy);
            const Sample s = locate(gx, gy, W, H);
            ftx = s.tx; fty = s.ty;
            const bool anyx = (unsigned)s.x0 < (unsigned)W || (unsigned)(s.x0 + 1) < (unsigned)W;
            const bool anyy = (unsigned)s.y0 < (unsigned)H || (unsigned)(s.y0 + 1) < (unsigned)H;
            const int dx = s.x0 - w - ox, dy = s.y0 - h - oy;                     // against THIS tile's window
            const bool near = dx >= -GT_R && dx <= GT_R - 1 && dy >= -GT_R && dy <= GT_R - 1;
            if (anyx && anyy && near) {
                const size_t it = (size_t)h * W + w;
#pragma unroll
                for (int c = 0; c < GT_CH; ++c)
                    if (c < C) g[c] = goN[(size_t)c * plane + it];
                key = ((s.y0 - ty0 + 2 * GT_R) << 16) | ((s.x0 - tx0 + 2 * GT_R) & 0xffff);
                const int rn = max(dx >= 0 ? dx + 1 : -dx, dy >= 0 ? dy + 1 : -dy);
                myrt = max(myrt, (unsigned)rn);
            }
        }
        s_key[idx] = key;
        s_tx[idx] = ftx;
        s_ty[idx] = fty;
#pragma unroll
        for (int c = 0; c < GT_CH; ++c) s_g[c][idx] = g[c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) myrt = max(myrt, (unsigned)__shfl_xor((int)myrt, o, 64));
    if ((tid & 63) == 0 && myrt > 1u) atomicMax(&s_rt, myrt);
    __syncthreads();
    // far pixels of this tile -> the tile's own list segment (no global counter), and the tile's max |gout| -> the scale of the
    // fixed-point scatter.  Only tiles that HAVE far pixels touch the two global words, and only when that would change them
    // (one word takes ~90 atomics per microsecond: 8 waves x 8192 tiles on it cost more than the whole pass).
    const size_t wgid = ((size_t)n * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tid == 0) ws.wg_count[wgid] = s_nfar;
    if (s_nfar) {
        gmax = wave_max(gmax);
        if ((tid & 63) == 0) red[tid >> 6] = gmax;
        __syncthreads();
        if (tid == 0) {
            float m = red[0];
            for (int i = 1; i < GT_THREADS / 64; ++i) m = fmaxf(m, red[i]);
            const unsigned bits = __float_as_uint(m);                   // positive floats order like their bit patterns
            if (bits > *(volatile unsigned*)ws.maxbits) atomicMax(ws.maxbits, bits);
            if (*(volatile unsigned*)ws.count == 0u) *(volatile unsigned*)ws.count = 1u;
        }
    }
    if (MODE == GRID_AFFINE && GG) {
        const int wg = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float t = block_sum(acc6[i], red);
            if (tid == 0) ws.gpart[((size_t)n * nwg + wg) * 6 + i] = t;
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < s_nfar; i += GT_THREADS) ws.far_list[wgid * (GT_W * GT_H) + i] = s_far[i];
    // ---- gather: a thread owns GT_TPT vertically adjacent texels (column lx, rows TPT*rg ..) and walks the (TPT + 2R) x (2R + 1)
    //      staged pixels around them once, in a fixed order; a pixel whose corner base is (ky, kx) touches texel rows ky, ky+1
    //      and columns kx, kx+1 with the bilinear weights of grid_sample ---------------------------------------------------------
    const int lx = tid & 63, rg = tid >> 6;
    const int xt = lx + 2 * GT_R, yt0 = GT_TPT * rg + 2 * GT_R;
    float sum[GT_TPT][GT_CH];
#pragma unroll
    for (int i = 0; i < GT_TPT; ++i)
#pragma unroll
        for (int c = 0; c < GT_CH; ++c) sum[i][c] = 0.f;
    // Round 6: only the staged pixels that CAN touch these texels are visited.  Every gathered pixel of the tile has its corner base within
    // [-rt, rt - 1] of its own position (relative to the window offset), rt = s_rt <= GT_R found while staging; the others — the outer
    // GT_R - rt rows / columns of the 7 x (TPT + 6) walk — would fail the test below anyway.  Same pixels, same order, same sums: bitwise
    // what the full walk gives; identity / the reference's linspace zoom / sub-pixel fields have rt = 1 (12 instead of 56 visits per thread).
    const int rt = (int)s_rt, skip = GT_R - rt;
#define GT_VISIT                                                                                                        \
            const int k = s_key[rowbase + dx]; \
            const int ex = xt - (int)(short)(k & 0xffff); \
            const int a = (k >> 16) - yt0; \
            if ((unsigned)ex <= 1u && a >= -1 && a <= GT_TPT - 1) { \
                const float ftx = s_tx[rowbase + dx], fty = s_ty[rowbase + dx]; \
                const float wx = ex ? ftx : 1.f - ftx; \
                const float wtop = wx * (1.f - fty), wbot = wx * fty; \
                float g[GT_CH]; \
_Pragma("unroll") \
                for (int c = 0; c < GT_CH; ++c) g[c] = s_g[c][rowbase + dx]; \
_Pragma("unroll") \
                for (int i = 0; i < GT_TPT; ++i) { \
                    const float wgt = (i == a) ? wtop : wbot; \
                    if (i == a || i == a + 1) { \
_Pragma("unroll") \
                        for (int c = 0; c < GT_CH; ++c) sum[i][c] += g[c] * wgt; \
                    } \
                } \
            }
    if (rt == GT_R) {                                  // rough fields: the full walk, fully unrolled as before
        for (int rr = 0; rr < GT_TPT + 2 * GT_R; ++rr) {
            const int rowbase = (GT_TPT * rg + rr) * GT_RW + lx;
#pragma unroll
            for (int dx = 0; dx <= 2 * GT_R; ++dx) {
                GT_VISIT
            }
        }
    } else if (rt == 1) {                              // identity, the reference's linspace zoom, sub-pixel fields: 3 x (TPT + 2) visits
#pragma unroll
        for (int rr = GT_R - 1; rr < GT_TPT + GT_R + 1; ++rr) {
            const int rowbase = (GT_TPT * rg + rr) * GT_RW + lx;
#pragma unroll
            for (int dx = GT_R - 1; dx <= GT_R + 1; ++dx) {
                GT_VISIT
            }
        }
    } else {
        for (int rr = skip; rr < GT_TPT + 2 * GT_R - skip; ++rr) {
            const int rowbase = (GT_TPT * rg + rr) * GT_RW + lx;
            for (int dx = skip; dx <= 2 * GT_R - skip; ++dx) {
                GT_VISIT
            }
        }
    }
#undef GT_VISIT
    float* ginN = gin + (size_t)n * C * plane;
    const int x = tx0 + lx;
#pragma unroll
    for (int i = 0; i < GT_TPT; ++i) {
        const int y = ty0 + GT_TPT * rg + i;
        if (y < H && x < W) {
#pragma unroll
            for (int c = 0; c < GT_CH; ++c) {
                if (c < C) {
                    float* q = ginN + (size_t)c * plane + (size_t)y * W + x;
                    *q = accum_gin ? *q + sum[i][c] : sum[i][c];
                }
            }
        }
    }
}

// one workgroup per tile, one thread per far pixel of it: corners scattered with 64-bit fixed-point atomics (associative =>
// reproducible)
template <int MODE>
__global__ __launch_bounds__(256) void far_scatter_kernel(const float* __restrict__ gsrc, const float* __restrict__ gout, int C,
                                                          int H, int W, GatherWs ws) {
    if (*ws.count == 0u) return;
    const size_t wgid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned count = ws.wg_count[wgid];
    if (count == 0u) return;
    const unsigned mb = *ws.maxbits;
    const int e = max((int)((mb >> 23) & 255u) - 126, -80);       // max |g| < 2^e (clamped: 2^(FIX_BITS - e) must be a float)
    const float scale = __uint_as_float((unsigned)(127 + FIX_BITS - e) << 23);   // 2^(FIX_BITS - e): products stay exact
    const size_t plane = (size_t)H * W;
    const int tiles_x = (W + GT_W - 1) / GT_W, tiles_y = (H + GT_H - 1) / GT_H;
    for (unsigned i = threadIdx.x; i < count; i += blockDim.x) {
        const unsigned ent = ws.far_list[wgid * (GT_W * GT_H) + i];
        const unsigned pid = ent & 0x0fffffffu, fmask = ent >> 28;          // pixel id | the corners no gather window reached
        const int w = (int)(pid % (unsigned)W);
        const unsigned t = pid / (unsigned)W;
        const int h = (int)(t % (unsigned)H), n = (int)(t / (unsigned)H);
        float th[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (MODE == GRID_AFFINE) {
#pragma unroll
            for (int k = 0; k < 6; ++k) th[k] = gsrc[n * 6 + k] + ((k == 0 || k == 4) ? 1.f : 0.f);
        }
        float gx, gy;
        make_grid<MODE>(gsrc, n, h, w, H, W, th, gx, gy);
        const Sample s = locate(gx, gy, W, H);
        const float ex = 1.f - s.tx, ey = 1.f - s.ty;
        const float wk[4] = {ex * ey, s.tx * ey, ex * s.ty, s.tx * s.ty};
        for (int k = 0; k < 4; ++k) {
            const int cx = s.x0 + (k & 1), cy = s.y0 + (k >> 1);
            if (!((fmask >> k) & 1u)) continue;
            if ((unsigned)cx >= (unsigned)W || (unsigned)cy >= (unsigned)H) continue;
            ws.dirty[((size_t)n * tiles_y + cy / GT_H) * tiles_x + cx / GT_W] = 1u;
            for (int c = 0; c < C; ++c) {
                const float v = gout[((size_t)n * C + c) * plane + (size_t)h * W + w] * wk[k];
                const long long q = __float2ll_rn(v * scale);
                atomicAdd(reinterpret_cast<unsigned long long*>(ws.acc + ((size_t)n * C + c) * plane + (size_t)cy * W + cx),
                          (unsigned long long)q);
            }
        }
    }
}

// tiles touched by the far scatter: grad_input += accumulator / scale; accumulator and flag back to zero
__global__ __launch_bounds__(256) void far_fold_kernel(float* __restrict__ gin, int C, int H, int W, GatherWs ws) {
    if (*ws.count == 0u) return;
    const int n = blockIdx.z;
    const size_t tile = ((size_t)n * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (ws.dirty[tile] == 0u) return;
    __syncthreads();                                   // every thread has read the flag before it is cleared
    const int e = max((int)((*ws.maxbits >> 23) & 255u) - 126, -80);
    const size_t plane = (size_t)H * W;
    for (int t = threadIdx.x; t < GT_W * GT_H; t += 256) {
        const int y = blockIdx.y * GT_H + (t >> 6), x = blockIdx.x * GT_W + (t & 63);
        if (y >= H || x >= W) continue;
        for (int c = 0; c < C; ++c) {
            const size_t o = ((size_t)n * C + c) * plane + (size_t)y * W + x;
            const long long a = ws.acc[o];
            if (a != 0) {
                gin[o] += (float)ldexp((double)a, e - FIX_BITS);
                ws.acc[o] = 0;
            }
        }
    }
    if (threadIdx.x == 0) ws.dirty[tile] = 0u;
}

// gtheta[n][i] (+)= sum over workgroups of gpart[n][wg][i], in workgroup order
__global__ __launch_bounds__(64) void affine_ggrid_fold_kernel(const float* __restrict__ gpart, float* __restrict__ ggrid, int nwg,
                                                               int accum) {
    const int n = blockIdx.x, i = threadIdx.x;
    if (i >= 6) return;
    float t = 0.f;
    for (int wg = 0; wg < nwg; ++wg) t += gpart[((size_t)n * nwg + wg) * 6 + i];
    ggrid[n * 6 + i] = accum ? ggrid[n * 6 + i] + t : t;
}

// nemar_grid_sample_tune (A/B build, common.h): 0 (default) = destination-tiled gather + fixed-point far path (needs the workspace; bitwise
// reproducible), 1 = LDS-tile fp32-atomic variant, 2 = global fp32 atomics (the round-1 default), bits 1..2 of values >= 4:
// ablations of the LDS-tile variant.  Round-1 measurements of the atomic variants (8x3x256^2, global vs LDS tile): zero /
// near-identity field 26 vs 43 us, smooth 3-pixel field 64 vs 45 us, white 1-pixel field 139 vs 44 us.
NEMAR_SWITCH(int, g_tiled_scatter, 0);
NEMAR_SWITCH(int, g_gather_512, 1);       // 512-thread workgroups in the gather pass (default: 170 vs 249 us at 8x3x1024^2); nemar_grid_sample_tune(16): 256
NEMAR_SWITCH(int, g_gather_follow, 1);    // gather windows follow the field (tile_offset_kernel); nemar_grid_sample_tune(32): centred on the tiles (round 2)
NEMAR_SWITCH(int, g_fwd_vec1, 1);         // forward: one pixel per lane (default: every gather / store instruction of a wave covers whole cache lines;
                            // 8x3x1024^2 identity 92 -> 71 us, smooth 3-px field 104 -> 74 us, profiles/r4_gs_fwd_vec.txt); nemar_grid_sample_tune(64):
                            // the round-1 form, 4 pixels per lane with 16-byte stores
NEMAR_SWITCH(int, g_gather_fused, 1);     // grid gradient fused into the gather pass (default; measured 5-10 % faster); nemar_grid_sample_tune(8): two passes

template <int MODE>
int launch_fwd(const float* in, const float* gsrc, float* out, int N, int C, int H, int W, int Ho, int Wo,
               hipStream_t st) {
    const bool vec4 = (Wo % 4 == 0) && (((uintptr_t)out & 15) == 0) && (((uintptr_t)gsrc & 15) == 0) &&
                      (MODE != GRID_EXPLICIT) && !g_fwd_vec1;
    const long long items = (long long)Ho * (vec4 ? Wo / 4 : Wo);
    int gx = nemar_cdiv(items, 256);
    const int cap = nemar_cdiv(256 * 8, N);
    if (gx > cap) gx = cap;
    dim3 grid(gx, N), block(256);
    NEMAR_AB_ONLY(if (vec4)
        hipLaunchKernelGGL((grid_sample_fwd_kernel<MODE, 4>), grid, block, 0, st, in, gsrc, out, C, H, W, Ho, Wo);
    else)
        hipLaunchKernelGGL((grid_sample_fwd_kernel<MODE, 1>), grid, block, 0, st, in, gsrc, out, C, H, W, Ho, Wo);
    return 0;
}

struct GatherLayout { size_t acc_off, dirty_off, zero_bytes, misc_off, wgc_off, list_off, gpart_off, toff_off, total; int tiles_x, tiles_y; };
GatherLayout gather_layout(int N, int C, int H, int W) {
    GatherLayout L;
    L.tiles_x = nemar_cdiv(W, GT_W); L.tiles_y = nemar_cdiv(H, GT_H);
    size_t o = 0;
    L.acc_off = o; o += sizeof(long long) * (size_t)N * C * H * W;
    L.dirty_off = o; o += sizeof(unsigned) * (size_t)N * L.tiles_x * L.tiles_y;
    o = (o + 15) & ~(size_t)15;
    L.zero_bytes = o;
    L.misc_off = o; o += 16;
    L.wgc_off = o; o += sizeof(unsigned) * (size_t)N * L.tiles_x * L.tiles_y;
    o = (o + 15) & ~(size_t)15;
    L.list_off = o; o += sizeof(unsigned) * (size_t)N * L.tiles_x * L.tiles_y * GT_W * GT_H;
    o = (o + 15) & ~(size_t)15;
    const int nogin_blocks = nemar_cdiv((long long)H * W, 256);     // per-block sums of the affine kernels (either of them)
    const int nwg = L.tiles_x * L.tiles_y > nogin_blocks ? L.tiles_x * L.tiles_y : nogin_blocks;
    L.gpart_off = o; o += sizeof(float) * (size_t)N * nwg * 6;
    o = (o + 15) & ~(size_t)15;
    L.toff_off = o; o += sizeof(int) * 2 * (size_t)N * L.tiles_x * L.tiles_y;
    L.total = o;
    return L;
}

template <int MODE>
int launch_bwd(const float* in, const float* gsrc, const float* gout, float* gin, int accum_gin, float* ggrid,
               int accum_ggrid, int N, int C, int H, int W, int Ho, int Wo, void* workspace, hipStream_t st) {
    const long long items = (long long)Ho * Wo;
    int gx = nemar_cdiv(items, 256);
    const int cap = nemar_cdiv(256 * 8, N);
    if (gx > cap) gx = cap;
    dim3 grid(gx, N), block(256);
    char* wsb = (char*)workspace;
    const GatherLayout L = gather_layout(N, C, H, W);
    float* gpart = (workspace && MODE == GRID_AFFINE) ? (float*)(wsb + L.gpart_off) : nullptr;
    const bool gather = gin && workspace && H == Ho && W == Wo && C <= GT_CH && g_tiled_scatter == 0 && N <= 65535 &&
                        (long long)N * H * W < (1ll << 28);        // (pixel id + 4-bit corner mask in one far-list word)
    if (gather) {
        GatherWs ws;
        ws.acc = (long long*)(wsb + L.acc_off); ws.dirty = (unsigned*)(wsb + L.dirty_off);
        ws.count = (unsigned*)(wsb + L.misc_off); ws.maxbits = ws.count + 1;
        ws.far_list = (unsigned*)(wsb + L.list_off); ws.gpart = (float*)(wsb + L.gpart_off);
        ws.wg_count = (unsigned*)(wsb + L.wgc_off);
        ws.toff = (const int*)(wsb + L.toff_off);
        (void)hipMemsetAsync(ws.count, 0, 12, st);                // count, maxbits, any_shift
        const dim3 tg(L.tiles_x, L.tiles_y, N);
        {
            const int nt = N * L.tiles_x * L.tiles_y;
            if (g_gather_follow)
                hipLaunchKernelGGL((tile_offset_kernel<MODE>), dim3(nemar_cdiv(nt, 256)), dim3(256), 0, st, gsrc, (int*)(wsb + L.toff_off), H, W,
                                   L.tiles_x, L.tiles_y, N, ws.count + 2);
            // (else: any_shift stays 0 = windows centred on the tiles, the round-2 scheme, for A/B)
        }
        if (g_gather_fused) {
            NEMAR_AB_ONLY(if (!g_gather_512)
                hipLaunchKernelGGL((grid_sample_bwd_gather_kernel<MODE, true, 256>), tg, dim3(256), 0, st, in, gsrc, gout, gin,
                                   accum_gin, ggrid, accum_ggrid, C, H, W, ws);
            else)
                hipLaunchKernelGGL((grid_sample_bwd_gather_kernel<MODE, true, 512>), tg, dim3(512), 0, st, in, gsrc, gout, gin,
                                   accum_gin, ggrid, accum_ggrid, C, H, W, ws);
            if (MODE == GRID_AFFINE)
                hipLaunchKernelGGL(affine_ggrid_fold_kernel, dim3(N), dim3(64), 0, st, (const float*)ws.gpart, ggrid,
                                   L.tiles_x * L.tiles_y, accum_ggrid);
        }
#ifdef NEMAR_AB
        else {
            // A/B variant: two streaming passes — d loss / d grid (grid_sample_bwd_kernel without grad_input), then the gather
            // pass over gsrc + gout only.  Measured 5-10 % SLOWER than the fused pass (profiles/r2_microbench.jsonl).
            hipLaunchKernelGGL((grid_sample_bwd_kernel<MODE, false>), grid, block, 0, st, in, gsrc, gout, nullptr, ggrid,
                               accum_ggrid, C, H, W, Ho, Wo, gpart);
            if (MODE == GRID_AFFINE)
                hipLaunchKernelGGL(affine_ggrid_fold_kernel, dim3(N), dim3(64), 0, st, (const float*)gpart, ggrid, gx, accum_ggrid);
            hipLaunchKernelGGL((grid_sample_bwd_gather_kernel<MODE, false, 256>), tg, dim3(256), 0, st, in, gsrc, gout, gin,
                               accum_gin, ggrid, accum_ggrid, C, H, W, ws);
        }
#endif
        hipLaunchKernelGGL((far_scatter_kernel<MODE>), tg, dim3(256), 0, st, gsrc, gout, C, H, W, ws);
        hipLaunchKernelGGL(far_fold_kernel, tg, dim3(256), 0, st, gin, C, H, W, ws);
        return 0;
    }
    // legacy scatter kernels: grad_input through fp32 atomics into the zero-filled (or accumulated) buffer
    if (gin && !accum_gin) (void)hipMemsetAsync(gin, 0, sizeof(float) * (size_t)N * C * H * W, st);
    if (MODE == GRID_AFFINE && !accum_ggrid && !gpart) (void)hipMemsetAsync(ggrid, 0, sizeof(float) * (size_t)N * 6, st);
#ifdef NEMAR_AB
    if (gin && H == Ho && W == Wo && (g_tiled_scatter & 1) && N <= 65535) {
        if (MODE == GRID_AFFINE && !accum_ggrid && gpart) (void)hipMemsetAsync(ggrid, 0, sizeof(float) * (size_t)N * 6, st);
        hipLaunchKernelGGL((grid_sample_bwd_tiled_kernel<MODE>), dim3(nemar_cdiv(W, TL_W), nemar_cdiv(H, TL_H), N),
                           dim3(TL_THREADS), 0, st, in, gsrc, gout, gin, ggrid, accum_ggrid, C, H, W, g_tiled_scatter >> 2);
        return 0;
    }
#endif
    if (gin)
        hipLaunchKernelGGL((grid_sample_bwd_kernel<MODE, true>), grid, block, 0, st, in, gsrc, gout, gin, ggrid,
                           accum_ggrid, C, H, W, Ho, Wo, gpart);
    else
        hipLaunchKernelGGL((grid_sample_bwd_kernel<MODE, false>), grid, block, 0, st, in, gsrc, gout, gin, ggrid,
                           accum_ggrid, C, H, W, Ho, Wo, gpart);
    if (gpart)
        hipLaunchKernelGGL(affine_ggrid_fold_kernel, dim3(N), dim3(64), 0, st, (const float*)gpart, ggrid, gx, accum_ggrid);
    return 0;
}

}  // namespace

#ifdef NEMAR_AB
NEMAR_API int nemar_grid_sample_tune(int variant) {
    g_fwd_vec1 = (variant & 64) ? 0 : 1;
    g_gather_fused = (variant & 8) ? 0 : 1;
    g_gather_512 = (variant & 16) ? 0 : 1;
    g_gather_follow = (variant & 32) ? 0 : 1;
    g_tiled_scatter = variant & ~56;
    return NEMAR_OK;
}
#endif

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

NEMAR_API size_t nemar_grid_sample_bwd_workspace(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return gather_layout(N, C, H, W).total;
}
NEMAR_API size_t nemar_grid_sample_bwd_zeroed_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return gather_layout(N, C, H, W).zero_bytes;
}

NEMAR_API int nemar_grid_sample_bwd(const float* in, const float* grid_src, int grid_mode, const float* gout,
                                    float* gin, int accum_gin, float* ggrid, int accum_ggrid, int N, int C, int H,
                                    int W, int Ho, int Wo, void* workspace, size_t ws_bytes, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(in && grid_src && gout && ggrid, "grid_sample_bwd: null pointer");
    NEMAR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "grid_sample_bwd: bad shape");
    NEMAR_REQUIRE((long long)H * W < (1ll << 31) && (long long)Ho * Wo < (1ll << 31) && N <= 65535,
                  "grid_sample_bwd: plane too large");
    if (workspace && ws_bytes < nemar_grid_sample_bwd_workspace(N, C, H, W)) {
        nemar_set_error("grid_sample_bwd: workspace %zu < %zu", ws_bytes, nemar_grid_sample_bwd_workspace(N, C, H, W));
        return NEMAR_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (grid_mode) {
        case GRID_EXPLICIT:
            launch_bwd<GRID_EXPLICIT>(in, grid_src, gout, gin, accum_gin, ggrid, accum_ggrid, N, C, H, W, Ho, Wo, workspace, st); break;
        case GRID_UNET:
            launch_bwd<GRID_UNET>(in, grid_src, gout, gin, accum_gin, ggrid, accum_ggrid, N, C, H, W, Ho, Wo, workspace, st); break;
        case GRID_AFFINE:
            launch_bwd<GRID_AFFINE>(in, grid_src, gout, gin, accum_gin, ggrid, accum_ggrid, N, C, H, W, Ho, Wo, workspace, st); break;
        default: NEMAR_REQUIRE(false, "grid_sample_bwd: unknown grid_mode %d", grid_mode);
    }
    NEMAR_CHECK_LAUNCH("grid_sample_bwd");
    return NEMAR_OK;
}
