// K5 + K6 + K7: pointwise activation gradients, 2x2 max-pool, bilinear resize, dropout.
// All HBM-bound streaming kernels: grid-stride, coalesced, 16 B/lane where the layout allows.
//   act_bwd        gradient through an activation fused into a conv epilogue (LeakyReLU / ReLU / Tanh) —
//                  reference models/networks.py:377,576,585,593 ; models/stn/layers.py:61-64
//   maxpool2       nn.MaxPool2d(2) — reference models/stn/layers.py:174
//   bilinear       F.interpolate(mode='bilinear', align_corners=False) — reference models/stn/unet_stn.py:96,
//                  188-195 ; models/nemar_model.py:187-188,204-205,226-227,240-241,254-255
//   dropout        nn.Dropout(0.5) — reference models/networks.py:427-428 (ON by default, nemar_model.py:102)
#include "common.h"
#include "max_words.h"

namespace {

constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3;

__device__ __forceinline__ float dact_from_out(float y, int act, float slope) {
    if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == ACT_LRELU) return y > 0.f ? 1.f : slope;
    if (act == ACT_TANH) return 1.f - y * y;
    return 1.f;
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                      float* __restrict__ gx, long long n, int act, float slope) {
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 g = reinterpret_cast<const float4*>(gy)[i];
        const float4 o = reinterpret_cast<const float4*>(y)[i];
        float4 r;
        r.x = g.x * dact_from_out(o.x, act, slope);
        r.y = g.y * dact_from_out(o.y, act, slope);
        r.z = g.z * dact_from_out(o.z, act, slope);
        r.w = g.w * dact_from_out(o.w, act, slope);
        reinterpret_cast<float4*>(gx)[i] = r;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        gx[i] = gy[i] * dact_from_out(y[i], act, slope);
}

// stand-alone activation (the U-Net generator's skip path needs relu() of a tensor whose producer already fused another
// activation; everywhere else activations ride in a conv / InstanceNorm epilogue)
__device__ __forceinline__ float act_apply_f(float v, int act, float slope) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == ACT_TANH) return tanhf(v);
    return v;
}
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, int act,
                                                      float slope) {
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 r;
        r.x = act_apply_f(v.x, act, slope);
        r.y = act_apply_f(v.y, act, slope);
        r.z = act_apply_f(v.z, act, slope);
        r.w = act_apply_f(v.w, act, slope);
        reinterpret_cast<float4*>(y)[i] = r;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = act_apply_f(x[i], act, slope);
}

// ---- 2x2 max pool ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                           int Ho, int Wo, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(idx % Wo);
        const long long t = idx / Wo;
        const int ho = (int)(t % Ho);
        const long long nc = t / Ho;
        const float* p = x + (nc * H + 2 * ho) * (long long)W + 2 * wo;
        const float2 a = *reinterpret_cast<const float2*>(p);   // W even or wo in range => 8-byte aligned pairs
        const float2 b = *reinterpret_cast<const float2*>(p + W);
        y[idx] = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
    }
}
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel_unaligned(const float* __restrict__ x, float* __restrict__ y,
                                                                     int H, int W, int Ho, int Wo, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(idx % Wo);
        const long long t = idx / Wo;
        const int ho = (int)(t % Ho);
        const long long nc = t / Ho;
        const float* p = x + (nc * H + 2 * ho) * (long long)W + 2 * wo;
        y[idx] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
    }
}

// gx = [addend +] scatter of gy to the first maximum of each window (torch's tie rule: row-major first)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           const float* __restrict__ addend, float* __restrict__ gx,
                                                           int H, int W, int Ho, int Wo, long long total_in) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_in;
         idx += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(idx % W);
        const long long t = idx / W;
        const int h = (int)(t % H);
        const long long nc = t / H;
        const int ho = h >> 1, wo = w >> 1;
        float g = 0.f;
        if (ho < Ho && wo < Wo) {
            const float* p = x + (nc * H + 2 * ho) * (long long)W + 2 * wo;
            const float v0 = p[0], v1 = p[1], v2 = p[W], v3 = p[W + 1];
            int am = 0;
            float m = v0;
            if (v1 > m) { m = v1; am = 1; }
            if (v2 > m) { m = v2; am = 2; }
            if (v3 > m) { m = v3; am = 3; }
            if (am == ((h & 1) << 1 | (w & 1))) g = gy[(nc * Ho + ho) * (long long)Wo + wo];
        }
        gx[idx] = addend ? addend[idx] + g : g;
    }
}

// ---- bilinear resize, align_corners=False ---------------------------------------------------------------------------
struct Tap1D { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap1D tap1d(int d, int n_in, float scale) {
    // s = max((d + 0.5) * in/out - 0.5, 0); i0 = floor(s); i1 = min(i0 + 1, in - 1)
    float s = ((float)d + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    Tap1D t;
    t.i0 = min((int)s, n_in - 1);
    t.i1 = min(t.i0 + 1, n_in - 1);
    t.l1 = s - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                           int Ho, int Wo, float sh, float sw, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(idx % Wo);
        const long long t = idx / Wo;
        const int ho = (int)(t % Ho);
        const long long nc = t / Ho;
        const Tap1D th = tap1d(ho, H, sh), tw = tap1d(wo, W, sw);
        const float* p = x + nc * (long long)H * W;
        const float top = p[th.i0 * W + tw.i0] * tw.l0 + p[th.i0 * W + tw.i1] * tw.l1;
        const float bot = p[th.i1 * W + tw.i0] * tw.l0 + p[th.i1 * W + tw.i1] * tw.l1;
        y[idx] = top * th.l0 + bot * th.l1;
    }
}

// scatter form (any size ratio): gx zero-filled by the entry point, fp32 atomics
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int H,
                                                           int W, int Ho, int Wo, float sh, float sw, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(idx % Wo);
        const long long t = idx / Wo;
        const int ho = (int)(t % Ho);
        const long long nc = t / Ho;
        const Tap1D th = tap1d(ho, H, sh), tw = tap1d(wo, W, sw);
        float* p = gx + nc * (long long)H * W;
        const float g = gy[idx];
        atomicAdd(p + th.i0 * W + tw.i0, g * th.l0 * tw.l0);
        atomicAdd(p + th.i0 * W + tw.i1, g * th.l0 * tw.l1);
        atomicAdd(p + th.i1 * W + tw.i0, g * th.l1 * tw.l0);
        atomicAdd(p + th.i1 * W + tw.i1, g * th.l1 * tw.l1);
    }
}

// Integer-factor DOWNsample backward as a gather (no atomics, no zero-fill): with H = f * Ho the source coordinate of output d is
// s = d*f + (f-1)/2 — for even f the two taps f*d + f/2 - 1 and f*d + f/2 with weights 1/2, 1/2; for odd f the single
// tap f*d + (f-1)/2 with weight 1 (its neighbour has weight 0).  So every input texel receives from at most ONE output pixel
// per axis: gx[h][w] = wy(h) * wx(w) * gy[h / fy][w / fx].  (the multi-resolution discriminators and regulariser: /2, /4)
__device__ __forceinline__ float down_weight(int i, int f) {
    const int r = i % f;
    if (f & 1) return r == (f - 1) / 2 ? 1.f : 0.f;
    return (r == f / 2 - 1 || r == f / 2) ? 0.5f : 0.f;
}
__global__ __launch_bounds__(256) void bilinear_down_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int H,
                                                                int W, int fy, int fx, long long total_in) {
    const int Ho = H / fy, Wo = W / fx;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_in;
         idx += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(idx % W);
        const long long t = idx / W;
        const int h = (int)(t % H);
        const long long nc = t / H;
        const float wy = down_weight(h, fy), wx = down_weight(w, fx);
        float v = 0.f;
        if (wy != 0.f && wx != 0.f) {
            // same products as the scatter form: g * l_y * l_x
            v = gy[nc * (long long)Ho * Wo + (long long)(h / fy) * Wo + w / fx] * wy * wx;
        }
        gx[idx] = v;
    }
}

// exact 2x upsample backward as a gather (no atomics): out[2i] = .25 in[i-1] + .75 in[i], out[2i+1] = .75 in[i] + .25 in[i+1]
// with the border taps clamped.  gin[i] collects from out rows 2i-1 .. 2i+2.
__device__ __forceinline__ void up2_taps(int i, int n, int* o, float* wgt, int& cnt) {
    // contributions of output index o to input i, for Ho = 2n
    cnt = 0;
    const int No = 2 * n;
    // out[2i]   : .75*in[i] + .25*in[max(i-1,0)]      out[2i+1] : .75*in[i] + .25*in[min(i+1,n-1)]
    o[cnt] = 2 * i; wgt[cnt++] = (i == 0) ? 1.f : 0.75f;
    o[cnt] = 2 * i + 1; wgt[cnt++] = (i == n - 1) ? 1.f : 0.75f;
    if (i + 1 < n) { o[cnt] = 2 * i + 2; wgt[cnt++] = 0.25f; }   // out[2(i+1)] takes .25*in[i]
    if (i >= 1) { o[cnt] = 2 * i - 1; wgt[cnt++] = 0.25f; }      // out[2(i-1)+1] takes .25*in[i]
    (void)No;
}
__global__ __launch_bounds__(256) void bilinear_up2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int H,
                                                               int W, long long total_in) {
    const int Wo = 2 * W, Ho = 2 * H;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_in;
         idx += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(idx % W);
        const long long t = idx / W;
        const int h = (int)(t % H);
        const long long nc = t / H;
        int oy[4], ox[4], ny, nx;
        float wy[4], wx[4];
        up2_taps(h, H, oy, wy, ny);
        up2_taps(w, W, ox, wx, nx);
        const float* p = gy + nc * (long long)Ho * Wo;
        float s = 0.f;
        for (int a = 0; a < ny; ++a) {
            float r = 0.f;
            for (int b = 0; b < nx; ++b) r += wx[b] * p[oy[a] * Wo + ox[b]];
            s += wy[a] * r;
        }
        gx[idx] = s;
    }
}

// ---- dropout: counter-based Philox4x32-10, 4 elements per counter; the mask is regenerated in backward ----------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// grid.y = sample (maxw != null: `n` = elements per sample, a multiple of 4; the Philox counter stays the GLOBAL float4 index, so the
// masks are those of the one-sample launch over the whole tensor)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long long n,
                                                      unsigned thresh, float scale, unsigned seed_lo, unsigned seed_hi,
                                                      unsigned offset, unsigned* maxw, const unsigned* obase) {
    if (obase) offset += *obase;             // (nemar_set_dropout_base: the per-step part of the offset lives in device memory)
    __shared__ unsigned red[4];
    const long long n4 = (n + 3) >> 2;
    const long long q0 = (long long)blockIdx.y * n4;
    x += (size_t)blockIdx.y * n;
    y += (size_t)blockIdx.y * n;
    unsigned omax = 0;
    for (long long ql = (long long)blockIdx.x * blockDim.x + threadIdx.x; ql < n4; ql += (long long)gridDim.x * blockDim.x) {
        const long long q = q0 + ql;
        unsigned r[4];
        philox4x32_10((unsigned)q, (unsigned)(q >> 32), offset, 0u, seed_lo, seed_hi, r);
        const long long i = ql << 2;
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            float4 o;
            o.x = r[0] >= thresh ? v.x * scale : 0.f;
            o.y = r[1] >= thresh ? v.y * scale : 0.f;
            o.z = r[2] >= thresh ? v.z * scale : 0.f;
            o.w = r[3] >= thresh ? v.w * scale : 0.f;
            *reinterpret_cast<float4*>(y + i) = o;
            if (maxw) {
                const unsigned a = __builtin_bit_cast(unsigned, o.x) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, o.y) & 0x7fffffffu,
                               c = __builtin_bit_cast(unsigned, o.z) & 0x7fffffffu, d = __builtin_bit_cast(unsigned, o.w) & 0x7fffffffu;
                omax = max(omax, max(max(a < 0x7f800000u ? a : 0u, b < 0x7f800000u ? b : 0u), max(c < 0x7f800000u ? c : 0u, d < 0x7f800000u ? d : 0u)));
            }
        } else {
            for (int k = 0; k < 4 && i + k < n; ++k) y[i + k] = r[k] >= thresh ? x[i + k] * scale : 0.f;
        }
    }
    if (maxw) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = omax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned m = max(max(red[0], red[1]), max(red[2], red[3]));
            maxw[gridDim.y + blockIdx.y * gridDim.x + blockIdx.x] = m;          // this workgroup's partial (max_words.h)
        }
    }
}

// ---- input pipeline: crop + horizontal flip + Normalize(0.5, 0.5) of a resident image pool, one launch per batch -------------
// y[b][c][h][w] = (pool[idx[b]][c][y0[b] + h][flip[b] ? x0[b] + Wc - 1 - w : x0[b] + w] * scale - 0.5) / 0.5
// params = int4 per sample (pool index, y0, x0, flip): the reference draws one crop position / flip per A-B pair
// (data/base_dataset.py get_params :63-78) and applies it to both images (get_transform :81-112, Normalize :111).
__global__ __launch_bounds__(256) void crop_flip_normalize_kernel(const float* __restrict__ pool, const int* __restrict__ params,
                                                                  float* __restrict__ y, int C, int H, int W, int Hc, int Wc,
                                                                  float scale, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(idx % Wc);
        long long t = idx / Wc;
        const int h = (int)(t % Hc); t /= Hc;
        const int c = (int)(t % C);
        const int b = (int)(t / C);
        const int4 pr = reinterpret_cast<const int4*>(params)[b];
        const int sx = pr.w ? pr.z + Wc - 1 - w : pr.z + w;
        const float v = pool[(((size_t)pr.x * C + c) * H + (pr.y + h)) * W + sx];
        y[idx] = (v * scale - 0.5f) / 0.5f;
    }
}

}  // namespace

NEMAR_API int nemar_act_bwd(const float* gy, const float* y, float* gx, long long n, int act, float slope, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(gy && y && gx && n > 0, "act_bwd: bad arguments");
    NEMAR_REQUIRE((((uintptr_t)gy | (uintptr_t)y | (uintptr_t)gx) & 15) == 0, "act_bwd: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(nemar_stream_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, gy, y,
                       gx, n, act, slope);
    NEMAR_CHECK_LAUNCH("act_bwd");
    return NEMAR_OK;
}

// ---- batch concatenation / gradient of batch slices, and the sum of two gradients (ABI 602) ----------------------------------------
// The batched passes of the model (T on [real_A ; R(real_A)], D on [real ; fake ; fake]: reference models/nemar_model.py:161-206 evaluates
// them as separate calls) concatenate along the batch and slice the result; autograd's own cat / zero-fill + copy + add kernels for that
// are the only ATen arithmetic a step would launch — and ATen's elementwise kernels use the packed-FP32 instruction forms this library is
// built without.  dst = [piece 0 | piece 1 | ...]; a NULL piece contributes zeros (a slice no loss term reads).
namespace {
constexpr int CONCAT_MAX = 8;
struct ConcatArgs { const float* src[CONCAT_MAX]; long long end[CONCAT_MAX]; int k; };
template <bool VEC>
__global__ __launch_bounds__(256) void concat_pieces_kernel(ConcatArgs a, float* __restrict__ dst, long long total) {
    const long long units = VEC ? (total >> 2) : total;
    for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long long)gridDim.x * 256) {
        const long long idx = VEC ? (u << 2) : u;
        int j = 0;
        long long beg = 0;
#pragma unroll
        for (int i = 0; i < CONCAT_MAX - 1; ++i)
            if (i + 1 < a.k && idx >= a.end[i]) { j = i + 1; beg = a.end[i]; }
        // (no dynamically indexed private array: select the pointer with a chain of compares)
        const float* s = a.src[0];
#pragma unroll
        for (int i = 1; i < CONCAT_MAX; ++i)
            if (j == i) s = a.src[i];
        if (VEC) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s) v = *reinterpret_cast<const float4*>(s + (idx - beg));
            *reinterpret_cast<float4*>(dst + idx) = v;
        } else {
            dst[idx] = s ? s[idx - beg] : 0.f;
        }
    }
}
__global__ __launch_bounds__(256) void add2_kernel(const float* a, const float* b, float* out, long long n) {      // (out may alias a or b: same element)
    const long long n4 = n >> 2;
    for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < n4; u += (long long)gridDim.x * 256) {
        const float4 x = reinterpret_cast<const float4*>(a)[u], y = reinterpret_cast<const float4*>(b)[u];
        reinterpret_cast<float4*>(out)[u] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = a[i] + b[i];
}
}  // namespace

NEMAR_API int nemar_concat_pieces(const float* const* pieces, const long long* counts, int k, float* dst, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(pieces && counts && dst && k >= 1 && k <= CONCAT_MAX, "concat_pieces: 1..8 pieces");
    ConcatArgs a;
    long long total = 0;
    bool vec = ((uintptr_t)dst & 15) == 0;
    for (int i = 0; i < CONCAT_MAX; ++i) {
        a.src[i] = i < k ? pieces[i] : nullptr;
        if (i < k) {
            NEMAR_REQUIRE(counts[i] > 0, "concat_pieces: empty piece");
            vec = vec && (counts[i] & 3) == 0 && ((uintptr_t)pieces[i] & 15) == 0;
            total += counts[i];
        }
        a.end[i] = total;
    }
    a.k = k;
    const long long units = vec ? total / 4 : total;
    if (vec) hipLaunchKernelGGL((concat_pieces_kernel<true>), dim3(nemar_stream_grid(units, 256)), dim3(256), 0, (hipStream_t)stream, a, dst, total);
    else hipLaunchKernelGGL((concat_pieces_kernel<false>), dim3(nemar_stream_grid(units, 256)), dim3(256), 0, (hipStream_t)stream, a, dst, total);
    NEMAR_CHECK_LAUNCH("concat_pieces");
    return NEMAR_OK;
}

NEMAR_API int nemar_add2(const float* a, const float* b, float* out, long long n, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(a && b && out && n > 0, "add2: bad arguments");
    NEMAR_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "add2: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(add2_kernel, dim3(nemar_stream_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    NEMAR_CHECK_LAUNCH("add2");
    return NEMAR_OK;
}

NEMAR_API int nemar_act_fwd(const float* x, float* y, long long n, int act, float slope, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && y && n > 0, "act_fwd: bad arguments");
    NEMAR_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "act_fwd: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(nemar_stream_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n,
                       act, slope);
    NEMAR_CHECK_LAUNCH("act_fwd");
    return NEMAR_OK;
}

NEMAR_API int nemar_maxpool2_fwd(const float* x, float* y, int planes, int H, int W, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && y && planes > 0 && H >= 2 && W >= 2, "maxpool2_fwd: bad arguments");
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)planes * Ho * Wo;
    dim3 grid(nemar_stream_grid(total, 256)), block(256);
    if ((W & 1) == 0 && (((uintptr_t)x) & 7) == 0)
        hipLaunchKernelGGL(maxpool2_fwd_kernel, grid, block, 0, (hipStream_t)stream, x, y, H, W, Ho, Wo, total);
    else
        hipLaunchKernelGGL(maxpool2_fwd_kernel_unaligned, grid, block, 0, (hipStream_t)stream, x, y, H, W, Ho, Wo, total);
    NEMAR_CHECK_LAUNCH("maxpool2_fwd");
    return NEMAR_OK;
}

// gx [planes,H,W] = (addend ? addend : 0) + unpool(gy [planes,H/2,W/2]) using x to recompute the argmax
NEMAR_API int nemar_maxpool2_bwd(const float* x, const float* gy, const float* addend, float* gx, int planes, int H, int W,
                                 void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && gy && gx && planes > 0 && H >= 2 && W >= 2, "maxpool2_bwd: bad arguments");
    const long long total = (long long)planes * H * W;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, gy,
                       addend, gx, H, W, H / 2, W / 2, total);
    NEMAR_CHECK_LAUNCH("maxpool2_bwd");
    return NEMAR_OK;
}

NEMAR_API int nemar_bilinear_fwd(const float* x, float* y, int planes, int H, int W, int Ho, int Wo, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "bilinear_fwd: bad arguments");
    const long long total = (long long)planes * Ho * Wo;
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H,
                       W, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo, total);
    NEMAR_CHECK_LAUNCH("bilinear_fwd");
    return NEMAR_OK;
}

// gx [planes,H,W] <- gy [planes,Ho,Wo]   (written, not accumulated)
NEMAR_API int nemar_bilinear_bwd(const float* gy, float* gx, int planes, int H, int W, int Ho, int Wo, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(gy && gx && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "bilinear_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (Ho == 2 * H && Wo == 2 * W) {
        const long long total = (long long)planes * H * W;
        hipLaunchKernelGGL(bilinear_up2_bwd_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, gy, gx, H, W,
                           total);
    } else if (Ho > 0 && Wo > 0 && H % Ho == 0 && W % Wo == 0 && H / Ho >= 2 && W / Wo >= 2) {
        const long long total = (long long)planes * H * W;
        hipLaunchKernelGGL(bilinear_down_bwd_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, gy, gx, H, W, H / Ho,
                           W / Wo, total);
    } else {
        // arbitrary ratios (not on the training path): scatter with fp32 atomics
        NEMAR_HIP_CALL(hipMemsetAsync(gx, 0, sizeof(float) * (size_t)planes * H * W, st));
        const long long total = (long long)planes * Ho * Wo;
        hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, gy, gx, H, W, Ho, Wo,
                           (float)H / (float)Ho, (float)W / (float)Wo, total);
    }
    NEMAR_CHECK_LAUNCH("bilinear_bwd");
    return NEMAR_OK;
}

// A device word added to the `offset` of every dropout-type launch (nemar_dropout, nemar_dropout_max, nemar_instnorm_fwd_planes) at run
// time; NULL (default) = none.  A captured hipGraph freezes launch arguments: with the per-step part of the Philox offset in this word the
// replayed step still draws fresh masks (the caller rewrites the word before every replay).
const unsigned* g_dropout_base = nullptr;
namespace {
struct Words8 { unsigned w[8]; };
__global__ void store_words_kernel(unsigned* dst, Words8 v, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = v.w[threadIdx.x];
}
}  // namespace

NEMAR_API int nemar_store_words(void* dst, const void* host_words, int n, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(dst && host_words && n >= 1 && n <= 8, "store_words: bad arguments");
    Words8 v;
    for (int i = 0; i < 8; ++i) v.w[i] = i < n ? ((const unsigned*)host_words)[i] : 0u;
    hipLaunchKernelGGL(store_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned*)dst, v, n);
    NEMAR_CHECK_LAUNCH("store_words");
    return NEMAR_OK;
}

NEMAR_API int nemar_set_dropout_base(const void* device_word) {
    g_dropout_base = (const unsigned*)device_word;
    return NEMAR_OK;
}

// y = x * mask / (1 - p) with mask ~ Bernoulli(1 - p) drawn from Philox(seed, offset); call it again with the same
// (seed, offset) on the upstream gradient for the backward pass.
NEMAR_API int nemar_dropout(const float* x, float* y, long long n, float p, unsigned long long seed, unsigned offset,
                            void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "dropout: bad arguments");
    NEMAR_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "dropout: pointers must be 16-byte aligned");
    const double t = (double)p * 4294967296.0;
    const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    hipLaunchKernelGGL(dropout_kernel, dim3(nemar_stream_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n,
                       thresh, 1.f / (1.f - p), (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), offset, (unsigned*)nullptr, g_dropout_base);
    NEMAR_CHECK_LAUNCH("dropout");
    return NEMAR_OK;
}

// ... over `samples` samples of `per_sample` elements (a multiple of 4), and max |y| of sample i into max_words[i] (a
// NEMAR_MAX_WORDS(samples) buffer, max_words.h).
// Same masks as nemar_dropout over the samples * per_sample elements.
NEMAR_API int nemar_dropout_max(const float* x, float* y, int samples, long long per_sample, float p, unsigned long long seed,
                                unsigned offset, void* max_words, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(x && y && max_words && samples > 0 && samples <= 65535 && per_sample > 0 && per_sample % 4 == 0 && p >= 0.f && p < 1.f,
                  "dropout_max: bad arguments");
    NEMAR_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "dropout_max: pointers must be 16-byte aligned");
    const double t = (double)p * 4294967296.0;
    const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    int gx = nemar_stream_grid(per_sample / 4, 256);
    if (gx * samples > 8192) gx = (8192 + samples - 1) / samples;
    hipLaunchKernelGGL(dropout_kernel, dim3(gx, samples), dim3(256), 0, (hipStream_t)stream, x, y, per_sample, thresh, 1.f / (1.f - p),
                       (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), offset, (unsigned*)max_words, g_dropout_base);
    max_words_finalize((unsigned*)max_words, samples, gx, (hipStream_t)stream);
    NEMAR_CHECK_LAUNCH("dropout_max");
    return NEMAR_OK;
}

// On-GPU augmentation of the input pipeline (reference data/base_dataset.py:63-112: crop, flip, ToTensor, Normalize).
// pool [M,C,H,W] (values in [0, 1/scale]), params [B,4] int32 (pool index, y0, x0, flip) -> y [B,C,Hc,Wc] in [-1,1].
NEMAR_API int nemar_crop_flip_normalize(const float* pool, const int* params, float* y, int M, int B, int C, int H, int W,
                                        int Hc, int Wc, float scale, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    NEMAR_REQUIRE(pool && params && y && M > 0 && B > 0 && C > 0 && Hc > 0 && Wc > 0 && Hc <= H && Wc <= W,
                  "crop_flip_normalize: bad arguments");
    NEMAR_REQUIRE((((uintptr_t)params) & 15) == 0, "crop_flip_normalize: params must be 16-byte aligned");
    const long long total = (long long)B * C * Hc * Wc;
    hipLaunchKernelGGL(crop_flip_normalize_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, pool,
                       params, y, C, H, W, Hc, Wc, scale, total);
    NEMAR_CHECK_LAUNCH("crop_flip_normalize");
    return NEMAR_OK;
}
