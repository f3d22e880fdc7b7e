// K13 + K15: loss reductions with fused gradients, and the fused Adam update.
//   l1      torch.nn.L1Loss()                      — reference models/nemar_model.py:68,179,195; also
//           mean|dtheta| of the affine STN          — reference models/stn/affine_stn.py:136-138 (b = NULL)
//   gan     GANLoss.__call__ vs a constant target   — reference models/networks.py:263-281
//           vanilla: BCEWithLogits -> mean softplus(-x) (real) / softplus(x) (fake); lsgan: mean (x-t)^2;
//           wgangp: -mean(x) (real) / mean(x) (fake)
//   adam    torch.optim.Adam(lr, betas=(beta1, 0.999)).step() — reference models/nemar_model.py:128-137,274,282-283
// Losses reduce deterministically in two stages (per-workgroup partials in the caller's workspace, one
// workgroup folds them); every loss is produced already multiplied by its lambda (`weight`) and may be accumulated
// into an existing device scalar, so the step never needs scalar arithmetic on the host or in other kernels.
#include "common.h"

namespace {

constexpr int GAN_VANILLA = 0, GAN_LSGAN = 1, GAN_WGANGP = 2;
constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }

__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         long long n, float* __restrict__ partial) {
    __shared__ float red[16];
    float acc = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        acc += fabsf(b ? a[i] - b[i] : a[i]);
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void gan_partial_kernel(const float* __restrict__ x, long long n, int mode, int real,
                                                          float* __restrict__ partial) {
    __shared__ float red[16];
    float acc = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = x[i];
        if (mode == GAN_VANILLA) acc += softplusf(real ? -v : v);
        else if (mode == GAN_LSGAN) { const float d = v - (real ? 1.f : 0.f); acc += d * d; }
        else acc += real ? -v : v;
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void finish_kernel(const float* __restrict__ partial, int n_partial, float scale,
                                                     int accumulate, float* __restrict__ loss) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += blockDim.x) acc += partial[i];
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) loss[0] = (accumulate ? loss[0] : 0.f) + scale * t;
}

__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ gscale, float scale, float* __restrict__ ga,
                                                     long long n, int accumulate) {
    const float s = gscale[0] * scale;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float d = b ? a[i] - b[i] : a[i];
        const float g = d > 0.f ? s : (d < 0.f ? -s : 0.f);
        ga[i] = accumulate ? ga[i] + g : g;
    }
}

__global__ __launch_bounds__(256) void gan_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gscale,
                                                      float scale, int mode, int real, float* __restrict__ gx, long long n) {
    const float s = gscale[0] * scale;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = x[i];
        float g;
        if (mode == GAN_VANILLA) g = 1.f / (1.f + __expf(-v)) - (real ? 1.f : 0.f);   // sigmoid(x) - target
        else if (mode == GAN_LSGAN) g = 2.f * (v - (real ? 1.f : 0.f));
        else g = real ? -1.f : 1.f;
        gx[i] = s * g;
    }
}

// torch.optim.Adam (single-tensor, non-capturable path), fp32:
//   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, value=1-b2); denom = v.sqrt()/sqrt(bc2) + eps; p.addcdiv_(m, denom, -lr/bc1)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float w1, float beta2,
                                                   float one_m_beta2, float step_size, float bc2_sqrt, float eps) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        float mi = m[i], vi = v[i];
        // at::lerp: weight < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
        const float diff = gi - mi;
        mi = (w1 < 0.5f) ? mi + w1 * diff : gi - diff * (1.f - w1);
        vi = vi * beta2 + (one_m_beta2 * gi) * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] + (-step_size) * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

// the same update with the two step-dependent scalars (lr / bias-correction-1, sqrt(bias-correction-2)) read from device memory: a captured
// hipGraph replays this launch every step, the host refreshes the two floats before each replay
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long long n, float w1, float beta2,
                                                       float one_m_beta2, const float* __restrict__ hyper, float eps) {
    const float step_size = hyper[0], bc2_sqrt = hyper[1];
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        float mi = m[i], vi = v[i];
        const float diff = gi - mi;
        mi = (w1 < 0.5f) ? mi + w1 * diff : gi - diff * (1.f - w1);
        vi = vi * beta2 + (one_m_beta2 * gi) * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] + (-step_size) * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

int red_blocks(long long n) {
    int b = nemar_cdiv(n, 256 * 4);
    if (b < 1) b = 1;
    if (b > RED_BLOCKS) b = RED_BLOCKS;
    return b;
}

}  // namespace

NEMAR_API size_t nemar_loss_workspace(void) { return sizeof(float) * RED_BLOCKS; }

// loss[0] = (accumulate ? loss[0] : 0) + weight * mean|a - b|      (b NULL => mean|a|)
NEMAR_API int nemar_l1_loss_fwd(const float* a, const float* b, long long n, float weight, float* loss, int accumulate,
                                void* workspace, size_t ws_bytes, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(a && loss && workspace && n > 0, "l1_loss_fwd: bad arguments");
    if (ws_bytes < nemar_loss_workspace()) { nemar_set_error("l1_loss_fwd: workspace too small"); return NEMAR_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, st, a, b, n, (float*)workspace);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, nb, (float)(weight / (double)n),
                       accumulate, loss);
    NEMAR_CHECK_LAUNCH("l1_loss_fwd");
    return NEMAR_OK;
}

// ga (+)= gscale[0] * weight * sign(a - b) / n
NEMAR_API int nemar_l1_loss_bwd(const float* a, const float* b, long long n, const float* gscale, float weight, float* ga,
                                int accumulate, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(a && gscale && ga && n > 0, "l1_loss_bwd: bad arguments");
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(nemar_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, gscale,
                       (float)(weight / (double)n), ga, n, accumulate);
    NEMAR_CHECK_LAUNCH("l1_loss_bwd");
    return NEMAR_OK;
}

// loss[0] = (accumulate ? loss[0] : 0) + weight * GANLoss(mode)(x, target_is_real)
NEMAR_API int nemar_gan_loss_fwd(const float* x, long long n, int mode, int target_is_real, float weight, float* loss,
                                 int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && loss && workspace && n > 0, "gan_loss_fwd: bad arguments");
    NEMAR_REQUIRE(mode >= 0 && mode <= 2, "gan_loss_fwd: unknown gan mode %d", mode);
    if (ws_bytes < nemar_loss_workspace()) { nemar_set_error("gan_loss_fwd: workspace too small"); return NEMAR_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(gan_partial_kernel, dim3(nb), dim3(256), 0, st, x, n, mode, target_is_real, (float*)workspace);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, nb, (float)(weight / (double)n),
                       accumulate, loss);
    NEMAR_CHECK_LAUNCH("gan_loss_fwd");
    return NEMAR_OK;
}

NEMAR_API int nemar_gan_loss_bwd(const float* x, long long n, int mode, int target_is_real, const float* gscale,
                                 float weight, float* gx, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && gscale && gx && n > 0, "gan_loss_bwd: bad arguments");
    NEMAR_REQUIRE(mode >= 0 && mode <= 2, "gan_loss_bwd: unknown gan mode %d", mode);
    hipLaunchKernelGGL(gan_bwd_kernel, dim3(nemar_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, gscale,
                       (float)(weight / (double)n), mode, target_is_real, gx, n);
    NEMAR_CHECK_LAUNCH("gan_loss_bwd");
    return NEMAR_OK;
}

// One Adam step over a flat parameter buffer (p, g, m, v all length n); `step` is 1-based.
NEMAR_API int nemar_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1,
                              double beta2, double eps, int step, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
    // hyper-parameters arrive as doubles (python floats) and are rounded to fp32 exactly where torch rounds them
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(nemar_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1), (float)sqrt(bc2),
                       (float)eps);
    NEMAR_CHECK_LAUNCH("adam_step");
    return NEMAR_OK;
}

// nemar_adam_step with its step-dependent scalars in device memory: hyper[0] = lr / (1 - beta1^step), hyper[1] = sqrt(1 - beta2^step),
// both computed by the caller in double and rounded to float (what nemar_adam_step does internally) — the form a captured hipGraph
// can replay (the launch arguments are frozen at capture, the two floats are refreshed before every replay).
NEMAR_API int nemar_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper, double beta1,
                                  double beta2, double eps, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(p && g && m && v && hyper && n > 0, "adam_step_dev: bad arguments");
    hipLaunchKernelGGL(adam_dev_kernel, dim3(nemar_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), hyper, (float)eps);
    NEMAR_CHECK_LAUNCH("adam_step_dev");
    return NEMAR_OK;
}
