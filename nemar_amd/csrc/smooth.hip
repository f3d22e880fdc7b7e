// K12: deformation-smoothness / bilateral regulariser, forward and backward.
//
// Replaces reference models/stn/stn_losses.py:4-30 (smoothness_loss), called from
// models/stn/unet_stn.py:179-201:
//   loss = mean(w1*|d[h+1,w]-d[h,w]|) + mean(w2*|d[h,w+1]-d[h,w]|)
//        + mean(w3*|d[h,w]-d[h+1,w+1]|) + mean(w4*|d[h,w+1]-d[h+1,w]|)
//   w_k  = mean_c exp(-alpha*|I[a]-I[b]|) on the same index pairs (alpha > 0 and an image given), else 1;
//          the image carries no gradient (detached, unet_stn.py:183).
// Four separate means: counts N*2*(H-1)*W, N*2*H*(W-1), N*2*(H-1)*(W-1) twice.
//
// HBM-bound stencil: 8 B/px forward (+4*Ci with the bilateral image), 16 B/px backward (+4*Ci).
// Forward is a deterministic two-stage reduction (per-workgroup partials in the caller's workspace,
// then one workgroup folds them); backward is a pure gather (no atomics).
#include "common.h"

namespace {

constexpr int MAX_CI = 8;

template <bool BILATERAL>
__device__ __forceinline__ float pair_weight(const float* __restrict__ img, size_t a, size_t b, size_t plane, int Ci,
                                             float alpha) {
    if (!BILATERAL) return 1.f;
    float s = 0.f;
    for (int c = 0; c < Ci; ++c) s += __expf(-alpha * fabsf(img[a + c * plane] - img[b + c * plane]));
    return s / (float)Ci;
}

template <bool BILATERAL>
__global__ __launch_bounds__(256) void smooth_fwd_kernel(const float* __restrict__ d, const float* __restrict__ img,
                                                         float alpha, float* __restrict__ partial, int H, int W,
                                                         int Ci, float inv1, float inv2, float inv3) {
    __shared__ float red[16];
    const int n = blockIdx.y;
    const size_t plane = (size_t)H * W;
    const float* d0 = d + (size_t)n * 2 * plane;
    const float* d1 = d0 + plane;
    const float* im = BILATERAL ? img + (size_t)n * Ci * plane : nullptr;
    float acc = 0.f;
    const int items = H * W;
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int h = it / W, w = it - h * W;
        const bool hv = h + 1 < H, wv = w + 1 < W;
        const size_t p = (size_t)it;
        if (hv) {  // vertical: (h+1,w) - (h,w)
            const float wt = pair_weight<BILATERAL>(im, p + W, p, plane, Ci, alpha);
            acc += inv1 * wt * (fabsf(d0[p + W] - d0[p]) + fabsf(d1[p + W] - d1[p]));
        }
        if (wv) {  // horizontal: (h,w+1) - (h,w)
            const float wt = pair_weight<BILATERAL>(im, p + 1, p, plane, Ci, alpha);
            acc += inv2 * wt * (fabsf(d0[p + 1] - d0[p]) + fabsf(d1[p + 1] - d1[p]));
        }
        if (hv && wv) {
            // main diagonal: (h,w) - (h+1,w+1)
            float wt = pair_weight<BILATERAL>(im, p, p + W + 1, plane, Ci, alpha);
            acc += inv3 * wt * (fabsf(d0[p] - d0[p + W + 1]) + fabsf(d1[p] - d1[p + W + 1]));
            // anti diagonal: (h,w+1) - (h+1,w)
            wt = pair_weight<BILATERAL>(im, p + 1, p + W, plane, Ci, alpha);
            acc += inv3 * wt * (fabsf(d0[p + 1] - d0[p + W]) + fabsf(d1[p + 1] - d1[p + W]));
        }
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void smooth_finish_kernel(const float* __restrict__ partial, int n_partial,
                                                            float factor, int accumulate, float* __restrict__ loss) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += blockDim.x) acc += partial[i];
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) loss[0] = (accumulate ? loss[0] : 0.f) + factor * t;
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// gd[n,c,h,w] (+)= gscale*factor * d loss / d d[n,c,h,w]; pixel p is the "a" end of four pairs and the "b" end
// of four others (see header); d|a-b|/da = sign(a-b), d/db = -sign(a-b), sign(0) = 0 like torch.abs.
template <bool BILATERAL>
__global__ __launch_bounds__(256) void smooth_bwd_kernel(const float* __restrict__ d, const float* __restrict__ img,
                                                         float alpha, const float* __restrict__ gscale, float factor,
                                                         float* __restrict__ gd, int accumulate, int H, int W, int Ci,
                                                         float inv1, float inv2, float inv3) {
    const int n = blockIdx.y;
    const size_t plane = (size_t)H * W;
    const float* dn = d + (size_t)n * 2 * plane;
    const float* im = BILATERAL ? img + (size_t)n * Ci * plane : nullptr;
    float* gn = gd + (size_t)n * 2 * plane;
    const float gs = gscale[0] * factor;
    const int items = H * W;
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int h = it / W, w = it - h * W;
        const bool up = h > 0, dn_ = h + 1 < H, lf = w > 0, rt = w + 1 < W;
        const size_t p = (size_t)it;
        float g0 = 0.f, g1 = 0.f;
        // as "b" of vertical pair with a = (h+1,w); as "a" of vertical pair with b = (h-1,w)
        if (dn_) {
            const float wt = inv1 * pair_weight<BILATERAL>(im, p + W, p, plane, Ci, alpha);
            g0 -= wt * sgn(dn[p + W] - dn[p]);
            g1 -= wt * sgn(dn[plane + p + W] - dn[plane + p]);
        }
        if (up) {
            const float wt = inv1 * pair_weight<BILATERAL>(im, p, p - W, plane, Ci, alpha);
            g0 += wt * sgn(dn[p] - dn[p - W]);
            g1 += wt * sgn(dn[plane + p] - dn[plane + p - W]);
        }
        // horizontal
        if (rt) {
            const float wt = inv2 * pair_weight<BILATERAL>(im, p + 1, p, plane, Ci, alpha);
            g0 -= wt * sgn(dn[p + 1] - dn[p]);
            g1 -= wt * sgn(dn[plane + p + 1] - dn[plane + p]);
        }
        if (lf) {
            const float wt = inv2 * pair_weight<BILATERAL>(im, p, p - 1, plane, Ci, alpha);
            g0 += wt * sgn(dn[p] - dn[p - 1]);
            g1 += wt * sgn(dn[plane + p] - dn[plane + p - 1]);
        }
        // main diagonal: a = (h,w), b = (h+1,w+1)
        if (dn_ && rt) {
            const float wt = inv3 * pair_weight<BILATERAL>(im, p, p + W + 1, plane, Ci, alpha);
            g0 += wt * sgn(dn[p] - dn[p + W + 1]);
            g1 += wt * sgn(dn[plane + p] - dn[plane + p + W + 1]);
        }
        if (up && lf) {  // p is b of the pair anchored at (h-1,w-1)
            const float wt = inv3 * pair_weight<BILATERAL>(im, p - W - 1, p, plane, Ci, alpha);
            g0 -= wt * sgn(dn[p - W - 1] - dn[p]);
            g1 -= wt * sgn(dn[plane + p - W - 1] - dn[plane + p]);
        }
        // anti diagonal: a = (h',w'+1), b = (h'+1,w')
        if (lf && dn_) {  // p = a of the pair anchored at (h,w-1): b = (h+1,w-1)
            const float wt = inv3 * pair_weight<BILATERAL>(im, p, p + W - 1, plane, Ci, alpha);
            g0 += wt * sgn(dn[p] - dn[p + W - 1]);
            g1 += wt * sgn(dn[plane + p] - dn[plane + p + W - 1]);
        }
        if (up && rt) {  // p = b of the pair anchored at (h-1,w): a = (h-1,w+1)
            const float wt = inv3 * pair_weight<BILATERAL>(im, p - W + 1, p, plane, Ci, alpha);
            g0 -= wt * sgn(dn[p - W + 1] - dn[p]);
            g1 -= wt * sgn(dn[plane + p - W + 1] - dn[plane + p]);
        }
        g0 *= gs;
        g1 *= gs;
        if (accumulate) { gn[p] += g0; gn[plane + p] += g1; } else { gn[p] = g0; gn[plane + p] = g1; }
    }
}

struct SmoothGeom {
    int gx;
    float inv1, inv2, inv3;
};
SmoothGeom geom(int N, int H, int W) {
    SmoothGeom g;
    g.gx = nemar_cdiv((long long)H * W, 256);
    const int cap = nemar_cdiv(256 * 8, N);
    if (g.gx > cap) g.gx = cap;
    const double c1 = (double)N * 2 * (H - 1) * W, c2 = (double)N * 2 * H * (W - 1), c3 = (double)N * 2 * (H - 1) * (W - 1);
    g.inv1 = c1 > 0 ? (float)(1.0 / c1) : 0.f;
    g.inv2 = c2 > 0 ? (float)(1.0 / c2) : 0.f;
    g.inv3 = c3 > 0 ? (float)(1.0 / c3) : 0.f;
    return g;
}

}  // namespace

NEMAR_API size_t nemar_smoothness_workspace(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const SmoothGeom g = geom(N, H, W);
    return sizeof(float) * (size_t)g.gx * N;
}

NEMAR_API int nemar_smoothness_fwd(const float* d, const float* img, int Ci, float alpha, float factor,
                                   float* loss, int accumulate, void* workspace, size_t ws_bytes, int N, int H, int W,
                                   void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(d && loss && workspace, "smoothness_fwd: null pointer");
    NEMAR_REQUIRE(N > 0 && N <= 65535 && H > 1 && W > 1, "smoothness_fwd: bad shape N=%d H=%d W=%d", N, H, W);
    const bool bil = img != nullptr && alpha > 0.f;
    NEMAR_REQUIRE(!bil || (Ci > 0 && Ci <= MAX_CI), "smoothness_fwd: image channels %d unsupported", Ci);
    const SmoothGeom g = geom(N, H, W);
    if (ws_bytes < sizeof(float) * (size_t)g.gx * N) {
        nemar_set_error("smoothness_fwd: workspace %zu < %zu", ws_bytes, sizeof(float) * (size_t)g.gx * N);
        return NEMAR_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    dim3 grid(g.gx, N), block(256);
    if (bil)
        hipLaunchKernelGGL((smooth_fwd_kernel<true>), grid, block, 0, st, d, img, alpha, partial, H, W, Ci, g.inv1,
                           g.inv2, g.inv3);
    else
        hipLaunchKernelGGL((smooth_fwd_kernel<false>), grid, block, 0, st, d, img, alpha, partial, H, W, Ci, g.inv1,
                           g.inv2, g.inv3);
    hipLaunchKernelGGL(smooth_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)partial, g.gx * N, factor,
                       accumulate, loss);
    NEMAR_CHECK_LAUNCH("smoothness_fwd");
    return NEMAR_OK;
}

NEMAR_API int nemar_smoothness_bwd(const float* d, const float* img, int Ci, float alpha, const float* gscale,
                                   float factor, float* gd, int accumulate, int N, int H, int W, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(d && gscale && gd, "smoothness_bwd: null pointer");
    NEMAR_REQUIRE(N > 0 && N <= 65535 && H > 1 && W > 1, "smoothness_bwd: bad shape N=%d H=%d W=%d", N, H, W);
    const bool bil = img != nullptr && alpha > 0.f;
    NEMAR_REQUIRE(!bil || (Ci > 0 && Ci <= MAX_CI), "smoothness_bwd: image channels %d unsupported", Ci);
    const SmoothGeom g = geom(N, H, W);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(g.gx, N), block(256);
    if (bil)
        hipLaunchKernelGGL((smooth_bwd_kernel<true>), grid, block, 0, st, d, img, alpha, gscale, factor, gd,
                           accumulate, H, W, Ci, g.inv1, g.inv2, g.inv3);
    else
        hipLaunchKernelGGL((smooth_bwd_kernel<false>), grid, block, 0, st, d, img, alpha, gscale, factor, gd,
                           accumulate, H, W, Ci, g.inv1, g.inv2, g.inv3);
    NEMAR_CHECK_LAUNCH("smoothness_bwd");
    return NEMAR_OK;
}
