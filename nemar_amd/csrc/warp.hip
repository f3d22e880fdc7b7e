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

template <int MODE>
int launch_bwd(const float* in, const float* gsrc, const float* gout, float* gin, float* ggrid, int accum_ggrid,
               int N, int C, int H, int W, int Ho, int Wo, hipStream_t st) {
    const long long items = (long long)Ho * Wo;
    int gx = nemar_cdiv(items, 256);
    const int cap = nemar_cdiv(256 * 8, N);
    if (gx > cap) gx = cap;
    dim3 grid(gx, N), block(256);
    if (gin)
        hipLaunchKernelGGL((grid_sample_bwd_kernel<MODE, true>), grid, block, 0, st, in, gsrc, gout, gin, ggrid,
                           accum_ggrid, C, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL((grid_sample_bwd_kernel<MODE, false>), grid, block, 0, st, in, gsrc, gout, gin, ggrid,
                           accum_ggrid, C, H, W, Ho, Wo);
    return 0;
}

}  // namespace

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
