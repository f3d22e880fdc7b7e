/* ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement (fp32 storage, double accumulation where the reference's ATen kernels accumulate in a wider
 * type is NOT assumed: arithmetic is float, like torch CPU fp32) of the HBM-bound operators of the NeMAR hot path:
 *   - F.grid_sample(bilinear, zeros, align_corners=False) forward/backward
 *         reference models/stn/unet_stn.py:173-174, models/stn/affine_stn.py:129-130
 *     with the UnetSTN grid (torch.linspace identity + planar offsets, models/stn/unet_stn.py:121-129,167)
 *   - smoothness_loss forward/backward, reference models/stn/stn_losses.py:4-30
 * The algorithm itself lives in PyTorch ATen (third party, not under /root/reference, unpinned — SURVEY.md §8c);
 * this file restates its published formulas.  Pinned by tests/test_oracle_c.py against
 * oracle/ops_np.py (itself pinned against torch and the reference's golden fixtures).
 * Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may load the library built from this file.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

static float linspace_m1_p1(int i, int n) {
    if (n <= 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return (i < n / 2) ? fmaf(step, (float)i, -1.f) : fmaf(-step, (float)(n - 1 - i), 1.f);
}

static float texel(const float* p, int H, int W, int y, int x) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? p[(size_t)y * W + x] : 0.f;
}

/* out[n,c,h,w] = bilinear sample of in[n,c] at the UnetSTN grid built from offsets[n,2,H,W] (H,W = Ho,Wo) */
void ref_unet_warp_fwd(const float* in, const float* off, float* out, int N, int C, int H, int W) {
    const size_t plane = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                const size_t o = (size_t)n * 2 * plane + (size_t)h * W + w;
                const float gx = linspace_m1_p1(w, W) + off[o], gy = linspace_m1_p1(h, H) + off[o + plane];
                const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
                const float fx = floorf(ix), fy = floorf(iy);
                const int x0 = (int)fx, y0 = (int)fy;
                const float tx = ix - fx, ty = iy - fy, ex = 1.f - tx, ey = 1.f - ty;
                for (int c = 0; c < C; ++c) {
                    const float* p = in + ((size_t)n * C + c) * plane;
                    out[((size_t)n * C + c) * plane + (size_t)h * W + w] =
                        texel(p, H, W, y0, x0) * (ex * ey) + texel(p, H, W, y0, x0 + 1) * (tx * ey) +
                        texel(p, H, W, y0 + 1, x0) * (ex * ty) + texel(p, H, W, y0 + 1, x0 + 1) * (tx * ty);
                }
            }
}

/* gin (zero-filled here, may be NULL) += scatter; goff[n,2,H,W] = d loss / d offsets */
void ref_unet_warp_bwd(const float* in, const float* off, const float* gout, float* gin, float* goff, int N, int C,
                       int H, int W) {
    const size_t plane = (size_t)H * W;
    if (gin) memset(gin, 0, sizeof(float) * (size_t)N * C * plane);
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                const size_t o = (size_t)n * 2 * plane + (size_t)h * W + w;
                const float gx = linspace_m1_p1(w, W) + off[o], gy = linspace_m1_p1(h, H) + off[o + plane];
                const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
                const float fx = floorf(ix), fy = floorf(iy);
                const int x0 = (int)fx, y0 = (int)fy;
                const float tx = ix - fx, ty = iy - fy, ex = 1.f - tx, ey = 1.f - ty;
                float gix = 0.f, giy = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float* p = in + ((size_t)n * C + c) * plane;
                    const float g = gout[((size_t)n * C + c) * plane + (size_t)h * W + w];
                    const float a = texel(p, H, W, y0, x0), b = texel(p, H, W, y0, x0 + 1);
                    const float cc = texel(p, H, W, y0 + 1, x0), d = texel(p, H, W, y0 + 1, x0 + 1);
                    gix += g * ((b - a) * ey + (d - cc) * ty);
                    giy += g * ((cc - a) * ex + (d - b) * tx);
                    if (gin) {
                        float* q = gin + ((size_t)n * C + c) * plane;
                        if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) q[(size_t)y0 * W + x0] += g * (ex * ey);
                        if (x0 + 1 >= 0 && x0 + 1 < W && y0 >= 0 && y0 < H) q[(size_t)y0 * W + x0 + 1] += g * (tx * ey);
                        if (x0 >= 0 && x0 < W && y0 + 1 >= 0 && y0 + 1 < H) q[(size_t)(y0 + 1) * W + x0] += g * (ex * ty);
                        if (x0 + 1 >= 0 && x0 + 1 < W && y0 + 1 >= 0 && y0 + 1 < H)
                            q[(size_t)(y0 + 1) * W + x0 + 1] += g * (tx * ty);
                    }
                }
                goff[o] = gix * (0.5f * (float)W);
                goff[o + plane] = giy * (0.5f * (float)H);
            }
}

static float pair_w(const float* img, size_t a, size_t b, size_t plane, int Ci, float alpha) {
    if (!img || alpha <= 0.f) return 1.f;
    float s = 0.f;
    for (int c = 0; c < Ci; ++c) s += expf(-alpha * fabsf(img[a + c * plane] - img[b + c * plane]));
    return s / (float)Ci;
}
static float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

/* smoothness_loss(d [N,2,H,W], img [N,Ci,H,W] or NULL, alpha); gd (may be NULL) receives d loss / d d */
double ref_smoothness(const float* d, const float* img, int Ci, float alpha, float* gd, int N, int H, int W) {
    const size_t plane = (size_t)H * W;
    const double c1 = (double)N * 2 * (H - 1) * W, c2 = (double)N * 2 * H * (W - 1), c3 = (double)N * 2 * (H - 1) * (W - 1);
    double l1 = 0, l2 = 0, l3 = 0, l4 = 0;
    if (gd) memset(gd, 0, sizeof(float) * (size_t)N * 2 * plane);
    for (int n = 0; n < N; ++n) {
        const float* im = img ? img + (size_t)n * Ci * plane : NULL;
        for (int ch = 0; ch < 2; ++ch) {
            const float* q = d + ((size_t)n * 2 + ch) * plane;
            float* g = gd ? gd + ((size_t)n * 2 + ch) * plane : NULL;
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    const size_t p = (size_t)h * W + w;
                    if (h + 1 < H) {
                        const float wt = pair_w(im, p + W, p, plane, Ci, alpha), df = q[p + W] - q[p];
                        l1 += wt * fabsf(df);
                        if (g) { g[p + W] += (float)(wt * sgnf(df) / c1); g[p] -= (float)(wt * sgnf(df) / c1); }
                    }
                    if (w + 1 < W) {
                        const float wt = pair_w(im, p + 1, p, plane, Ci, alpha), df = q[p + 1] - q[p];
                        l2 += wt * fabsf(df);
                        if (g) { g[p + 1] += (float)(wt * sgnf(df) / c2); g[p] -= (float)(wt * sgnf(df) / c2); }
                    }
                    if (h + 1 < H && w + 1 < W) {
                        float wt = pair_w(im, p, p + W + 1, plane, Ci, alpha), df = q[p] - q[p + W + 1];
                        l3 += wt * fabsf(df);
                        if (g) { g[p] += (float)(wt * sgnf(df) / c3); g[p + W + 1] -= (float)(wt * sgnf(df) / c3); }
                        wt = pair_w(im, p + 1, p + W, plane, Ci, alpha);
                        df = q[p + 1] - q[p + W];
                        l4 += wt * fabsf(df);
                        if (g) { g[p + 1] += (float)(wt * sgnf(df) / c3); g[p + W] -= (float)(wt * sgnf(df) / c3); }
                    }
                }
        }
    }
    return l1 / c1 + l2 / c2 + l3 / c3 + l4 / c3;
}
