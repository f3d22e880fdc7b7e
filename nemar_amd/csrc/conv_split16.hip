// nemar_amd — stride-1 / pad-1 convolutions of the wide layers (the translation net's 256-channel 3x3 residual blocks: reference
// models/networks.py:418-439, 18 convolutions per pass, 36.5 of the 63.8 ms of convolution time in a round-1 step; and the
// discriminator's 256 -> 512 4x4 layer, networks.py:576-597) on the 16-BIT matrix pipe, at fp32 accuracy.  Forward and data
// gradient here, weight gradient in conv_split16_wgrad.hip.
//
// gfx950 has no TF32-like mode and its fp32 MFMA runs at the vector rate (157 TFLOP/s); the 16-bit MFMAs run 16x faster.  Every
// fp32 operand is split into 16-bit terms whose partial products, accumulated in fp32 by the MFMA, reproduce the fp32 product to
// (better than) the rounding error of an fp32 multiply-add chain:
//   * fp16 x 3 (default):  v 2^k = h + l + e,  h = RN16(v 2^k), l = RN16(v 2^k - h), |e| <= 2^-22 |v 2^k|; products h h' + h l' + l h'
//     (the dropped l l' and the two e terms are ~3 * 2^-22 relative: an fp32 accumulation of the layer's 2304-term dot products
//     contributes ~2^-19 either way).  fp16's exponent range is narrow, so each tensor is scaled by the power of two that puts its
//     largest magnitude at 2^11..2^12 (absmax_kernel -> one word; folded border sums stay below 2^14) and the epilogue takes the
//     two scales out again, exactly.  3/16 of the fp32-MFMA issue time, 4 operand bytes per element.
//   * bf16 x 6 (nemar_tune(21, 3)):  v = b0 + b1 + b2 EXACTLY (3 x 8 significant bits), products of weight 2^-16 and above:
//     a0c0 + (a0c1 + a1c0) + (a0c2 + a1c1 + a2c0); no scaling (bf16 has fp32's exponent range).  6/16 of the issue time, 6 bytes.
//   Measured against float64 on the bench shape (tests/test_conv_real_shapes_gpu.py::test_split16_error_is_fp32_class): max error
//   fwd 3.4e-6 (fp16 x 3) / 4.4e-6 (bf16 x 6) / 5.2e-6 (exact-fp32 MFMA kernel), rms 3.1e-7 / 4.3e-7 / 5.0e-7 — the 16-bit MFMA
//   adds its 16 products per instruction before rounding once, the fp32 MFMA rounds after every product.
//
// Data flow of one convolution (all inside nemar_conv2d_fwd / nemar_conv2d_bwd_data):
//   1. absmax_kernel (fp16 form) + split_planes_kernel: source [N,C,H,W] fp32 -> NPL 16-bit planes, CHANNEL-BLOCKED and PADDED:
//         plane[t][n][c/8][row 0..H+3][slot 0..W+3][8 channels]      (16 bytes per (pixel, channel group))
//      rows 1..H / slots 1..W hold the image, row 0 / H+1 and slot 0 / W+1 the padding ALREADY MATERIALISED (zeros, or the
//      mirrored texels of nn.ReflectionPad2d(1)), rows H+2, H+3 and slots W+2, W+3 the pre-folded border sums the reflect DATA
//      gradient needs (below).  One lane's MFMA operand (8 consecutive reduction channels of one pixel) is one 16-byte word,
//      and a tile's halo (its rows + 2, full padded width) is ONE contiguous run per (plane, channel group).
//   2. igemm_split16_kernel: implicit GEMM, workgroup tile = 128 output channels x 256 pixels (whole image rows), four waves of
//      128 channels x 64 pixels (8 accumulators of 32x32), one per SIMD.  Per 16-channel chunk of the reduction the tile's halo is
//      copied once (global_load_lds, 16 bytes per lane, no VGPRs) into a double-buffered LDS region and all nine taps read it at
//      shifted addresses — the source is fetched once per chunk, not once per tap — plus one stage of packed weights per tap
//      into a 4-slot ring.  One workgroup barrier per tap.  How the copies are issued and why: see the kernel.
//      Few-tile layers (32x32 maps) cut the reduction into slab runs summed in order; layers below 2 GMAC stay on the exact-fp32
//      kernels (conv.hip decides); 4x4 layers run on the input-sized domain with the last output row / column masked.
//   3. the reflect data gradient  dx = Pad^T(Conv^T(gy))  folds the padded border back: output row 1 receives the gradient of
//      padded row 0, which only the first filter row produces, i.e. for that (row, tap) pair the source row is gy[0] + gy[2]
//      instead of gy[2]; same for row H-2, columns 1 and W-2, and the four corners.  The split kernel writes those sums once
//      (rows H+2 / H+3, slots W+2 / W+3) and the MFMA waves select them by address: no ring launch, no extra taps.
// Weights are split and re-ordered once per optimizer step (split16_pack_kernel) into [chunk][tap][128-row block][plane][k group]
// [128][8]: a stage is one contiguous run.
#include "common.h"
#include "conv_split16.h"
#include "pack_plan.h"
#include "max_words.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rn(float v) {
    const __bf16 h = (__bf16)v;                                   // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf16_val(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// v == b0 + b1 + b2 exactly (both subtractions are exact in fp32: Sterbenz-type cancellation of the leading bits)
__device__ __forceinline__ void split3(float v, unsigned short& b0, unsigned short& b1, unsigned short& b2) {
    b0 = bf16_rn(v);
    const float r1 = v - bf16_val(b0);
    b1 = bf16_rn(r1);
    const float r2 = r1 - bf16_val(b1);
    b2 = bf16_rn(r2);
}

__device__ __forceinline__ int mirror(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// value of plane position (row, slot) of one channel image xc [Hs, Ws_] seen through a (H + 4) x (W + 4) plane whose row / slot
// `pad` is source row / column 0 (pad = 1: the 3x3 and 4x4 pad-1 layers; pad = 2: the 4x4 data gradient, a full correlation)
__device__ __forceinline__ float plane_value(const float* xc, int row, int slot, int H, int W, int mode, int pad, int Hs, int Ws_) {
    if (mode != SPLIT16_DGRAD_REFLECT) {
        int y = row - pad, x = slot - pad;
        if (mode == SPLIT16_REFLECT) {
            if (row > H + 1 || slot > W + 1) return 0.f;
            y = mirror(y, H);
            x = mirror(x, W);
        } else if ((unsigned)y >= (unsigned)Hs || (unsigned)x >= (unsigned)Ws_) {
            return 0.f;
        }
        return xc[y * Ws_ + x];
    }
    int ya, yb = -1, xa, xb = -1;
    if (row >= 1 && row <= H) ya = row - 1;
    else if (row == H + 2) { ya = 0; yb = 2; }
    else if (row == H + 3) { ya = H - 3; yb = H - 1; }
    else return 0.f;
    if (slot >= 1 && slot <= W) xa = slot - 1;
    else if (slot == W + 2) { xa = 0; xb = 2; }
    else if (slot == W + 3) { xa = W - 3; xb = W - 1; }
    else return 0.f;
    float v = xc[ya * W + xa];
    if (yb >= 0) v += xc[yb * W + xa];
    if (xb >= 0) {
        float u = xc[ya * W + xb];
        if (yb >= 0) u += xc[yb * W + xb];
        v += u;
    }
    return v;
}

__device__ __forceinline__ u32x4 pack8(const unsigned short* b) {
    u32x4 o;
    o[0] = (unsigned)b[0] | ((unsigned)b[1] << 16);
    o[1] = (unsigned)b[2] | ((unsigned)b[3] << 16);
    o[2] = (unsigned)b[4] | ((unsigned)b[5] << 16);
    o[3] = (unsigned)b[6] | ((unsigned)b[7] << 16);
    return o;
}

// ---- fp16 x 3 variant of the same idea ---------------------------------------------------------------------------------
// fp16 carries 11 significant bits: v * 2^k = h + l + e with h = RN16(v 2^k), l = RN16(v 2^k - h), |e| <= 2^-22 |v 2^k|, and three
// products  h h' + h l' + l h'  leave a relative error of ~3 * 2^-22 per product — far below what the fp32 ACCUMULATION of a
// 2304-term dot product contributes either way (measured: tests/test_conv_real_shapes_gpu.py).  Half the MFMAs and two thirds of
// the operand bytes of the bf16 x 6 form.  fp16's exponent range is narrow, so every tensor is scaled by a power of two that puts
// its largest magnitude at 2^11..2^12 (folded border sums of the reflect data gradient stay below 2^14); the scale comes from one
// max-reduction pass (absmax_kernel -> a word behind the planes / behind the packed weights) and is taken out again, exactly, in
// the convolution's epilogue.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float pow2_scale(unsigned maxbits) {       // 2^(11 - floor(log2 max)), 1 for zero / non-finite
    const int e = (int)((maxbits >> 23) & 255u);
    if (e == 0 || e == 255) return 1.f;
    const int se = 127 + 11 - (e - 127);
    if (se < 1 || se > 254) return 1.f;
    return __builtin_bit_cast(float, (unsigned)se << 23);
}
__device__ __forceinline__ unsigned short f16_rn(float v) {
    const _Float16 h = (_Float16)v;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ void split2_f16(float v, unsigned short& h, unsigned short& l) {
    h = f16_rn(v);
    const float r = v - (float)__builtin_bit_cast(_Float16, h);
    l = f16_rn(r);
}

// max |x| over the FINITE elements as the bit pattern of a non-negative float (ordered like an unsigned integer); non-finite
// elements do not take part: they become fp16 infinities / NaN on their own and must not push the scale of everything else to 1.
// 16-byte loads; `n4` whole float4s + tail.
__device__ __forceinline__ unsigned finite_bits(unsigned u) {
    u &= 0x7fffffffu;
    return u < 0x7f800000u ? u : 0u;
}
__device__ __forceinline__ unsigned block_absmax(const float* __restrict__ x, long long n, unsigned* red) {
    unsigned m = 0;
    const long long n4 = ((size_t)x & 15) == 0 ? n >> 2 : 0;          // (a view at an odd offset: scalar loads throughout)
    const u32x4* x4 = reinterpret_cast<const u32x4*>(x);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const u32x4 v = x4[i];
        m = max(max(m, finite_bits(v[0])), max(max(finite_bits(v[1]), finite_bits(v[2])), finite_bits(v[3])));
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = max(m, finite_bits(__builtin_bit_cast(unsigned, x[i])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    return max(max(red[0], red[1]), max(red[2], red[3]));
}

// out[sample] zeroed by the caller; grid.y = sample (`per` elements each).  (A ticketed last-block reduction that needs no zeroing
// was measured at 53 us against 22 us for this form on the 33 MB bench tensors: its device-scope release fence per block writes back
// the XCD's L2.)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long per, unsigned* out) {
    __shared__ unsigned red[4];
    const unsigned m = block_absmax(x + (size_t)blockIdx.y * per, per, red);
    if (threadIdx.x == 0 && m) atomicMax(out + blockIdx.y, m);
}

// one thread = one (n, channel group, row, slot): 8 strided reads (coalesced across the slots of a row), NPL x 16-byte writes
template <int NPL>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, u32x4* __restrict__ out, int N, int C,
                                                           int H, int W, int mode, long long total, const unsigned* maxbits,
                                                           int mstride, int pad, int Hs, int Ws_) {
    const int Hp = H + 4, Ws = W + 4, CG = C >> 3;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int slot = (int)(t % Ws);
        long long q = t / Ws;
        const int row = (int)(q % Hp);
        q /= Hp;
        const int cg = (int)(q % CG), n = (int)(q / CG);
        const float scale = NPL == 2 ? pow2_scale(maxbits[n * mstride]) : 1.f;        // one scale per SAMPLE (mstride 0: per tensor)
        const float* xc = x + ((size_t)n * C + (size_t)cg * 8) * Hs * Ws_;
        unsigned short b0[8], b1[8], b2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = plane_value(xc + (size_t)j * Hs * Ws_, row, slot, H, W, mode, pad, Hs, Ws_);
            if (NPL == 3) split3(v, b0[j], b1[j], b2[j]);
            else split2_f16(v * scale, b0[j], b1[j]);
        }
        out[t] = pack8(b0);
        out[total + t] = pack8(b1);
        if (NPL == 3) out[2 * total + t] = pack8(b2);
    }
}

// packed weights: 16-byte word index (((chunk * NT + tap) * mblks + mblk) * NPL + plane) * 256 + kgroup * 128 + m, NT = KS * KS taps
template <int NPL>
__global__ __launch_bounds__(256) void split16_pack_kernel(const float* __restrict__ w, u32x4* __restrict__ out, int M, int Cred,
                                                       int dgrad, const unsigned* maxbits, int NT) {
    const int mblks = M >> 7;
    const long long total = (long long)(Cred >> 4) * NT * mblks * 256;
    const float scale = NPL == 2 ? pow2_scale(*maxbits) : 1.f;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(t & 127), kg = (int)((t >> 7) & 1);
        long long q = t >> 8;
        const int mblk = (int)(q % mblks);
        q /= mblks;
        const int tap = (int)(q % NT), chunk = (int)(q / NT);
        const int mg = mblk * 128 + m;
        unsigned short b0[8], b1[8], b2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cr = chunk * 16 + kg * 8 + j;
            // forward: w[K = M][C = Cred][3][3];  data gradient: w[K = Cred][C = M][3][3] with the taps flipped
            const float v = dgrad ? w[((size_t)cr * M + mg) * NT + (NT - 1 - tap)] : w[((size_t)mg * Cred + cr) * NT + tap];
            if (NPL == 3) split3(v, b0[j], b1[j], b2[j]);
            else split2_f16(v * scale, b0[j], b1[j]);
        }
        u32x4* o = out + (((size_t)(chunk * NT + tap) * mblks + mblk) * NPL) * 256 + kg * 128 + m;
        o[0] = pack8(b0);
        o[256] = pack8(b1);
        if (NPL == 3) o[512] = pack8(b2);
    }
}

// the fp16 x 3 pack as a job of a weight-pack plan (pack_plan.h): the scale comes from the NEMAR_PACK_MAX_PARTS partial maxima a stage-1
// job wrote (no zero fill, no atomics); block 0 also leaves the final max word where the convolution kernels read it
struct Split16PackArgs {
    const float* w; u32x4* out;
    int M, Cred, dgrad, NT;
    const unsigned* parts; unsigned* maxword;
    int gx, gy;
};
__device__ __forceinline__ void split16_pack_body(const Split16PackArgs& a, int bx, int, int gx) {
    unsigned mb = 0;
#pragma unroll
    for (int i = 0; i < NEMAR_PACK_MAX_PARTS; ++i) mb = max(mb, a.parts[i]);
    if (bx == 0 && threadIdx.x == 0) *a.maxword = mb;
    const int mblks = a.M >> 7, NT = a.NT, M = a.M, Cred = a.Cred;
    const long long total = (long long)(Cred >> 4) * NT * mblks * 256;
    const float scale = pow2_scale(mb);
    for (long long t = (long long)bx * 256 + threadIdx.x; t < total; t += (long long)gx * 256) {
        const int m = (int)(t & 127), kg = (int)((t >> 7) & 1);
        long long q = t >> 8;
        const int mblk = (int)(q % mblks);
        q /= mblks;
        const int tap = (int)(q % NT), chunk = (int)(q / NT);
        const int mg = mblk * 128 + m;
        unsigned short b0[8], b1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cr = chunk * 16 + kg * 8 + j;
            const float v = a.dgrad ? a.w[((size_t)cr * M + mg) * NT + (NT - 1 - tap)] : a.w[((size_t)mg * Cred + cr) * NT + tap];
            split2_f16(v * scale, b0[j], b1[j]);
        }
        u32x4* o = a.out + (((size_t)(chunk * NT + tap) * mblks + mblk) * 2) * 256 + kg * 128 + m;
        o[0] = pack8(b0);
        o[256] = pack8(b1);
    }
}
NEMAR_PACK_MULTI(split16_pack_multi_kernel, Split16PackArgs, split16_pack_body, 256)
void split16_pack_multi(const void* jobs, int njobs, int gx, int gy, hipStream_t st) {
    hipLaunchKernelGGL(split16_pack_multi_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, (const Split16PackArgs*)jobs);
}
struct RegSplit16Pack {
    RegSplit16Pack() { nemar_pack_register(PACK_FAM_SPLIT16, sizeof(Split16PackArgs), split16_pack_multi); }
} g_reg_split16_pack;

struct Split16Params {
    const u32x4* planes;       // split source, see split_planes_kernel
    const u32x4* wp;           // packed weights
    const float* bias;         // [M] or null
    int bias_al;               // bias is 16-byte aligned (float4 loads in the epilogue)
    float* dst;                // [N, M, H, W]
    int N, H, W, M, Cred;
    int Ws, HpWs;              // slots per plane row, 16-byte words per (plane, n, channel group) image
    int wshift, RT;            // log2 W, output rows per tile (256 / W)
    int tiles_per_img, mblks;
    int halo_instr, aux_instr; // 1 KiB DMA instructions per (plane, k group) for the halo rows / the two folded rows
    int halo16, aux16;         // exact 16-byte words of those two runs (second-generation kernel: exact-sized LDS regions)
    int fold;                  // reflect data gradient: select the folded rows / slots
    int xcd;                   // workgroup -> tile mapping keeps neighbouring tiles on one XCD
    long long plane16;         // 16-byte words per plane
    int ksplit;                // reduction runs per tile (1: none); slab z of a tile starts at dst + z * slab_stride
    long long slab_stride;
    int OH, OW;                // valid output extents (4x4 layers: H - 1, W - 1; stores beyond them are masked)
    const unsigned* xmax;      // fp16 x 3 form: max |source| per sample (word n * xstride) and max |weight| bit patterns (the
    const unsigned* wmax;      // power-of-two scales follow from them)
    int xstride;
    long long* tl;             // NEMAR_TIMELINE builds: cycle stamps of workgroup 0 (tools/timeline_split16.py)
    const float* addend;       // [N, M, OH, OW] added to the result in the epilogue (the ResnetBlock skip gradient), or null (ksplit == 1 only)
    unsigned* maxw;            // NEMAR_MAX_WORDS(N) buffer: this workgroup's max |result| into its partial word, or null (ksplit == 1 only)
    int maxw_lazy;             // the consumer reduces the partial words: result word n = the marker (max_words.h)
};

__device__ __forceinline__ void glds16(const u32x4* g, u32x4* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// scalar (wave-uniform) base + per-lane byte offset: hipcc selects the SGPR-base form of global_load_lds for it
__device__ __forceinline__ void glds16u(const u32x4* ubase, unsigned lane_bytes, u32x4* lds) {
    glds16(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(ubase) + lane_bytes), lds);
}

// ---- the product kernel --------------------------------------------------------------------------------------------------
// What the first generation's measurements said (tools/timeline_split16.py, tools/probes/dma_probe2.hip):
//   * a wave that streams MFMAs back to back monopolises the instruction issue of its SIMD: the loader waves, co-resident with MFMA
//     waves 0 and 1, got one instruction issued per ~8 MFMAs (a six-copy burst that takes 120 cycles on a quiet SIMD took 1750;
//     s_setprio makes no difference), so every tap ended with the MFMA waves parked at the barrier while the loaders finished:
//     tap = MFMA block + loader tail = 2900 cycles for 1536 cycles of matrix pipe.  => no loader waves: the four MFMA waves issue
//     the copies themselves, wave w a quarter of every weight stage and every fourth halo copy; it waits for ITS copies (counted
//     vmcnt) before the tap's barrier, and the barrier makes all four quarters visible.
//   * one wave per SIMD owns the whole 512-register file: every fragment of tap T + 1 is read during tap T into a second register
//     set, so the MFMAs of a tap never wait for LDS and rotate over the eight accumulators (no dependent chains).
//   * the tap body is branch-free: the halo copies of a chunk are described once per wave (source offset, LDS offset, lane limit)
//     in registers indexed by the unrolled tap position, the tail re-issues harmless copies instead of skipping, and the per-tap
//     copy count is a compile-time function of the tap, so the counted waits need no filler copies.
//   * the weight stage for tap T + 4 is issued during tap T, into the slot whose fragments went to registers a tap ago: two full
//     taps of lead on a 4-slot ring.
//   * hipcc clumps the LDS reads and copies of a tap (any clump longer than an MFMA's 32-cycle shadow idles the matrix pipe): the
//     body is cut into slots of [<= 1 memory instruction + its address arithmetic][its share of the MFMAs], pinned by sched_barrier.
//   Result (timeline, gpurun_out/timeline_bf6_v4.txt): 1680-1730 cycles per 48-MFMA tap = 90 % matrix-pipe issue, after which the
//   clock, not the schedule, is what is left: under this MFMA density the chip runs at ~1.45 GHz.  Hence fp16 x 3: half the MFMAs.
// Two chunks (18 taps) per loop iteration, so that the register-set parity is a compile-time constant of the tap position.
// NPL = operand planes: 3 = bf16 x 6 products, 2 = fp16 x 3 products (scaled, see split2_f16)
// KS = filter size (3x3, or the discriminator's 4x4 / pad 1 layers computed on the input-sized domain, last row / column masked)
// RING = weight-stage slots: 4 (stage T + 4 issued during tap T), or 3 — with 3 slots and no folded border rows the 3x3 fp16 x 3 kernel
// needs 76.8 KB of LDS and TWO workgroups share a CU (two waves per SIMD: while one sits at a barrier or waits for LDS the other issues
// MFMAs).  LDS is dynamic: [RING weight stages][two halo buffers].  (RING = 3 needs NT % 3 == 0: the slot of a tap is then a
// compile-time constant.)
// EPI = the fused epilogue of the ResnetBlock data gradient (p.addend, p.maxw): its own instantiation, so that the forward / plain
// data-gradient kernel keeps the register allocation it was tuned with (with the 128 addend loads in one epilogue the allocator ran
// out of architectural registers and moved loop-carried values to the accumulator file: 189 -> 218 us per launch)
template <int NBW, int NPL, int KS = 3, int RING = 4, int EPI = 0>      // EPI: 0 plain, 1 + max words, 2 + addend + max words   // NBW = halo copy slots per wave per tap on taps 3..NT-1 of a chunk (they carry the next chunk's halo)
__global__ __launch_bounds__(256) void igemm_split16_kernel(Split16Params p) {
    constexpr int NT = KS * KS;                          // taps
    constexpr int ASTAGE16 = 256 * NPL, KB = (NT - 3) * NBW, NREG = 2 * NPL, NP = NPL == 3 ? 6 : 3, NMFMA = 8 * NP;
    constexpr int ACOPY = NPL;                           // 1 KiB weight copies per wave per stage (a quarter of the stage)
    static_assert(RING == 4 || (RING == 3 && NT % 3 == 0), "ring slot of a tap must be a compile-time constant");
#define S16_SLOTOF(stage_, tap_) (RING == 4 ? ((stage_) & 3) : ((tap_) % 3))       /* tap_ == stage_ mod NT, up to a multiple of 3 */
#ifdef NEMAR_HOST_EMULATION
    __shared__ __attribute__((aligned(16))) u32x4 smem[9728];      // (the emulator has no dynamic LDS)
#else
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
#endif
    u32x4* const As = smem;
    u32x4* const Bs = smem + RING * ASTAGE16;
    const int region16 = p.halo16 + p.aux16, bbuf16 = NREG * region16;

    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int t = blockIdx.x;
    if (p.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);
    // few-tile layers (the discriminator's 32x32 maps: 64 / 128 tiles for 256 CUs): the reduction is cut into p.ksplit runs of
    // chunks, run z of a tile in its own workgroup writing slab z; the slabs are summed in order afterwards (host side)
    const int zsplit = t % p.ksplit;
    t /= p.ksplit;
    const int ptile = t / p.mblks, mblk = t - ptile * p.mblks;
    const int n = ptile / p.tiles_per_img, y0 = (ptile - n * p.tiles_per_img) * p.RT;
    const int nchunks = (p.Cred >> 4) / p.ksplit, nstage = nchunks * NT;       // of THIS workgroup
    const int chunk0 = zsplit * nchunks;
    const int CG = p.Cred >> 3;

    // ---- this wave's copies: weights = words [192 wid, 192 wid + 192) of every 768-word stage; halo = every fourth 1 KiB copy ----
    const int hi = (p.halo16 + 63) >> 6, ai = (p.aux16 + 63) >> 6, ipr = hi + ai, ncopies = NREG * ipr;
    const size_t wstage = (size_t)p.mblks * ASTAGE16;
    const u32x4* const wsrc0 = p.wp + (size_t)mblk * ASTAGE16 + wid * (64 * ACOPY) + (size_t)chunk0 * NT * wstage;
    const u32x4* const bsrc0 = p.planes + ((size_t)n * CG + 2 * chunk0) * p.HpWs;
    const int halo_off = y0 * p.Ws, aux_off = (p.H + 2) * p.Ws;
    unsigned goff[KB], blds[KB];      // per halo copy of this wave: source word offset from the chunk's images (+ lane), LDS word offset
    int blim[KB];                     // ... and the number of lanes that take part
    int share = 0;
    {
        int breg = 0, bin = wid;
        while (bin >= ipr) { bin -= ipr; ++breg; }
        unsigned g = 0, l = 0;
        int lim = 0;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            if (wid + 4 * k < ncopies) {
                const bool aux = bin >= hi;
                const int j = aux ? bin - hi : bin;
                const int pl = breg >> 1, kg = breg & 1;
                g = (unsigned)((size_t)pl * p.plane16 + (size_t)kg * p.HpWs + (aux ? aux_off : halo_off) + j * 64);
                l = (unsigned)(breg * region16 + (aux ? p.halo16 : 0) + j * 64);
                lim = (aux ? p.aux16 : p.halo16) - j * 64;
                share = k + 1;
                bin += 4;
                while (bin >= ipr) { bin -= ipr; ++breg; }
            }
            goff[k] = g + lane;       // (slots beyond the share repeat the last copy)
            blds[k] = l;
            blim[k] = lim;
        }
    }
    // copies of one stage: COUNT(ti) = ACOPY + (ti >= 3 ? NBW : 0) wave-instructions, in this order
#define S16_COPIES(stage_, ti_, ci_)                                                                                   \
    {                                                                                                                   \
        const int st_ = min((stage_), nstage - 1);                   /* tail: harmless re-copies of the last stage */   \
        const u32x4* const a_ = wsrc0 + (size_t)st_ * wstage + lane;                                                    \
        u32x4* const ad_ = As + S16_SLOTOF(stage_, ti_) * ASTAGE16 + wid * (64 * ACOPY);                                \
        _Pragma("unroll") for (int q = 0; q < ACOPY; ++q) glds16(a_ + 64 * q, ad_ + 64 * q);                            \
        if ((ti_) >= 3) {                                                                                               \
            const int hc_ = min((ci_) + 1, nchunks - 1);             /* chunk whose halo travels with this stage */     \
            const u32x4* const bb_ = bsrc0 + (size_t)(2 * hc_) * p.HpWs;                                                \
            u32x4* const bd_ = Bs + (((ci_) + 1) & 1) * bbuf16;                                                         \
            _Pragma("unroll") for (int q = 0; q < NBW; ++q) {                                                           \
                const int k_ = ((ti_) - 3) * NBW + q;                                                                   \
                if (lane < blim[k_]) glds16(bb_ + goff[k_], bd_ + blds[k_]);                                            \
            }                                                                                                           \
        }                                                                                                               \
    }
#define S16_COUNT(ti_) (ACOPY + ((ti_) >= 3 ? NBW : 0))
#define S16_VMCNT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14));

    // ---- MFMA side ----
    const int l31 = lane & 31, lhi = lane >> 5;
    int row[2], col[2];
    bool top[2], bot[2], lft[2], rgt[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int px = 64 * wid + 32 * nt + l31;
        row[nt] = px >> p.wshift;
        col[nt] = px & (p.W - 1);
        const int y = y0 + row[nt];
        top[nt] = p.fold && y == 1;
        bot[nt] = p.fold && y == p.H - 2;
        lft[nt] = p.fold && col[nt] == 1;
        rgt[nt] = p.fold && col[nt] == p.W - 2;
    }
    const int auxoff = p.halo16;
    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    u32x4 af[2][4][NPL], bf[2][2][NPL];
    // all fragments of tap (r_, sx_) of the chunk in halo buffer hb_, weights in ring slot slot_, into register set set_
#define S16_READ(set_, slot_, hb_, r_, sx_)                                                                            \
    {                                                                                                                   \
        const u32x4* const Bb_ = Bs + (hb_) * bbuf16 + lhi * region16;                                                  \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                              \
            int ra_ = (row[nt] + (r_) + z) * p.Ws;       /* z: see the loop head */                                     \
            if ((r_) == 2) ra_ = top[nt] ? auxoff : ra_;                                                                \
            if ((r_) == 0) ra_ = bot[nt] ? auxoff + p.Ws : ra_;                                                         \
            int sl_ = col[nt] + (sx_);                                                                                  \
            if ((sx_) == 2) sl_ = lft[nt] ? p.W + 2 : sl_;                                                              \
            if ((sx_) == 0) sl_ = rgt[nt] ? p.W + 3 : sl_;                                                              \
            _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) bf[set_][nt][pl] = Bb_[pl * 2 * region16 + ra_ + sl_];   \
        }                                                                                                               \
        const u32x4* const Ab_ = As + (slot_) * ASTAGE16 + lhi * 128 + l31;                                             \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                                \
            _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) af[set_][mt][pl] = Ab_[pl * 256 + mt * 32];              \
    }
    // partial products, smallest first: bf16 (a2 b0) (a1 b1) (a0 b2) (a1 b0) (a0 b1) (a0 b0); fp16 (l h') (h l') (h h')
    constexpr int PA[6] = {NPL == 3 ? 2 : 1, NPL == 3 ? 1 : 0, 0, 1, 0, 0}, PB[6] = {0, 1, NPL == 3 ? 2 : 0, 0, 1, 0};

    // ---- prologue: this wave's share of the first halo and of stages 0..3 ----
    {
        const u32x4* const bb = bsrc0;
#pragma unroll
        for (int k = 0; k < KB; ++k)
            if (k < share && lane < blim[k]) glds16(bb + goff[k], Bs + blds[k]);
    }
    S16_COPIES(0, 0, 0)
    S16_COPIES(1, 1, 0)
    S16_COPIES(2, 2, 0)
    if (RING == 4) {
        S16_COPIES(3, 3, 0)
        S16_VMCNT(2 * ACOPY + NBW)                   // (stages 2 and 3 may be in flight) the first halo and stages 0, 1 have landed
    } else {
        S16_VMCNT(ACOPY)                             // (stage 2 may be in flight)
    }
    __builtin_amdgcn_s_barrier();                     // ... for all four waves
    int z = 0;
    S16_READ(0, 0, 0, 0, 0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                     // every wave holds the fragments of tap 0: slot 0 may be refilled (tap 0 does)
    for (int chunk0 = 0; chunk0 < nchunks; chunk0 += 2) {
        // an opaque zero per iteration: keeps hipcc from hoisting the 18 taps' fragment addresses out of the loop (that costs
        // more registers than the file has: 512 + spills) — they are three VALU instructions each to recompute
#ifndef NEMAR_HOST_EMULATION
        asm volatile("s_mov_b32 %0, 0" : "=s"(z));
#endif
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int chunk = chunk0 + half;
            if (half == 1 && chunk >= nchunks) break;
            const int hb = half;                      // = chunk & 1
#pragma unroll
            for (int tap = 0; tap < NT; ++tap) {
                const int cur = (tap + half * NT) & 1, nxt = cur ^ 1;     // = T & 1 with T = NT chunk + tap
                const int T = chunk * NT + tap;
                // tap T + 1: every fragment into the other register set (after the last tap: a harmless read of stale LDS);
                // stage T + 4 into the slot of stage T (read during tap T - 1): two full taps ahead of its first use.
                // 23 + NBW slots of [<= 1 memory instruction + its scalar / vector arithmetic][2 MFMAs], pinned: hipcc otherwise
                // clumps the reads and copies, and every clump longer than an MFMA's 32-cycle shadow idles the matrix pipe
#ifdef NEMAR_TIMELINE
                const bool xprobe = p.tl != nullptr && blockIdx.x == 0 && lane == 0 && T >= 40 && T < 48;
#define S16_STAMP(i_) if (xprobe) p.tl[(wid * 8 + (T - 40)) * 8 + (i_)] = clock64();
#else
#define S16_STAMP(i_)
#endif
                S16_STAMP(0)
                const int ntap = tap == NT - 1 ? 0 : tap + 1;
                const int nhb = tap == NT - 1 ? hb ^ 1 : hb;
                const int nr = ntap / KS, nsx = ntap % KS;
                const int iti = tap + RING >= NT ? tap + RING - NT : tap + RING;        // tap position / chunk of stage T + RING
                const int ici = tap + RING >= NT ? chunk + 1 : chunk;
#define S16_MFMAS(beg_, end_)                 /* MFMAs [beg_, end_) of the tap's NMFMA: m = 8 q + 2 mt + nt */        \
                _Pragma("unroll") for (int m_ = (beg_); m_ < (end_) && m_ < NMFMA; ++m_) {                              \
                    const int q_ = m_ >> 3, mt_ = (m_ & 7) >> 1, nt_ = m_ & 1;                                          \
                    if constexpr (NPL == 3)                                                                             \
                        acc[mt_][nt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[cur][mt_][PA[q_]]), \
                                                                                __builtin_bit_cast(bf16x8, bf[cur][nt_][PB[q_]]), \
                                                                                acc[mt_][nt_], 0, 0, 0);                \
                    else                                                                                                \
                        acc[mt_][nt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[cur][mt_][PA[q_]]), \
                                                                               __builtin_bit_cast(f16x8, bf[cur][nt_][PB[q_]]), \
                                                                               acc[mt_][nt_], 0, 0, 0);                 \
                }                                                                                                       \
                __builtin_amdgcn_sched_barrier(0);
                // slot i of NSLOT gets MFMAs [i NMFMA / NSLOT, (i + 1) NMFMA / NSLOT)
                constexpr int NSLOT = 1 + 6 * NPL + ACOPY + NBW;
#define S16_SLOT(i_) S16_MFMAS((i_) * NMFMA / NSLOT, ((i_) + 1) * NMFMA / NSLOT)
                int baddr[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    int ra_ = (row[nt] + nr + z) * p.Ws;
                    if (KS == 3 && nr == 2) ra_ = top[nt] ? auxoff : ra_;
                    if (KS == 3 && nr == 0) ra_ = bot[nt] ? auxoff + p.Ws : ra_;
                    int sl_ = col[nt] + nsx;
                    if (KS == 3 && nsx == 2) sl_ = lft[nt] ? p.W + 2 : sl_;
                    if (KS == 3 && nsx == 0) sl_ = rgt[nt] ? p.W + 3 : sl_;
                    baddr[nt] = ra_ + sl_;
                }
                const u32x4* const Bn_ = Bs + nhb * bbuf16 + lhi * region16;
                const u32x4* const An_ = As + S16_SLOTOF(T + 1, tap + 1) * ASTAGE16 + lhi * 128 + l31;
                S16_SLOT(0)
#pragma unroll
                for (int i = 0; i < 2 * NPL; ++i) {           // the B fragments
                    bf[nxt][i / NPL][i % NPL] = Bn_[(i % NPL) * 2 * region16 + baddr[i / NPL]];
                    S16_SLOT(1 + i)
                }
#pragma unroll
                for (int i = 0; i < 4 * NPL; ++i) {           // the A fragments
                    af[nxt][i / NPL][i % NPL] = An_[(i % NPL) * 256 + (i / NPL) * 32];
                    S16_SLOT(1 + 2 * NPL + i)
                }
                {                                             // the copies of stage T + RING
                    const int st_ = min(T + RING, nstage - 1);   // (tail: harmless re-copies of the last stage)
                    const u32x4* const a_ = wsrc0 + (size_t)st_ * wstage + lane;
                    u32x4* const ad_ = As + S16_SLOTOF(T, tap) * ASTAGE16 + wid * (64 * ACOPY);
#pragma unroll
                    for (int q = 0; q < ACOPY; ++q) {
                        glds16(a_ + 64 * q, ad_ + 64 * q);
                        S16_SLOT(1 + 6 * NPL + q)
                    }
                    if (iti >= 3) {
                        const int hc_ = min(ici + 1, nchunks - 1);
                        const u32x4* const bb_ = bsrc0 + (size_t)(2 * hc_) * p.HpWs;
                        u32x4* const bd_ = Bs + ((ici + 1) & 1) * bbuf16;
#pragma unroll
                        for (int q = 0; q < NBW; ++q) {
                            const int k_ = (iti - 3) * NBW + q;
                            if (lane < blim[k_]) glds16(bb_ + goff[k_], bd_ + blds[k_]);
                            S16_SLOT(1 + 6 * NPL + ACOPY + q)
                        }
                    } else {
                        S16_MFMAS((1 + 6 * NPL + ACOPY) * NMFMA / NSLOT, NMFMA)
                    }
                }
#undef S16_SLOT
#undef S16_MFMAS
                // this wave's copies of stage T + 2 have landed; those of T + 3 and T + 4 may still be in flight
                const int t3 = tap + 3 >= NT ? tap + 3 - NT : tap + 3;
                // (a 3-slot ring has only stage T + 3 = T + RING in flight behind the one that must have landed)
                const int nfl = RING == 4 ? S16_COUNT(t3) + S16_COUNT(iti) : S16_COUNT(iti);     // compile-time after unrolling
                S16_STAMP(1)
                if (nfl == ACOPY) S16_VMCNT(ACOPY)
                else if (nfl == ACOPY + NBW) S16_VMCNT(ACOPY + NBW)
                else if (nfl == 2 * ACOPY) S16_VMCNT(2 * ACOPY)
                else if (nfl == 2 * ACOPY + NBW) S16_VMCNT(2 * ACOPY + NBW)
                else S16_VMCNT(2 * ACOPY + 2 * NBW)
                S16_STAMP(2)
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): every fragment of tap T + 1 is in registers
                S16_STAMP(3)
                __builtin_amdgcn_s_barrier();         // B_T: stage T + 2 complete for all waves; slot of stage T + 1 is free
                S16_STAMP(4)
#undef S16_STAMP
            }
        }
    }
    wait_vmem();                                      // (the tail's re-copies)
#undef S16_READ
#undef S16_VMCNT
#undef S16_COUNT
#undef S16_COPIES
#undef S16_SLOTOF

    const size_t HW = (size_t)p.OH * p.OW;
    // fp16 form: take the two power-of-two operand scales out again (exact)
    const float unscale = NPL == 2 ? 1.f / (pow2_scale(p.xmax[n * p.xstride]) * pow2_scale(*p.wmax)) : 1.f;
    if (!EPI) {
        // element (mt, r) of a lane lies a wave-uniform (32 mt + (r & 3) + 8 (r >> 2)) HW floats behind the lane's first one: one 64-bit
        // add per store; the 16 bias values of a group are four aligned 16-byte loads (round 6: the per-element form — a guarded scalar
        // load and a 64-bit multiply per store — cost 15 us of a 200 us launch)
        const bool with_bias = p.bias != nullptr && zsplit == 0;
        // the two 32-pixel halves of a lane pair (nt = 0, 1) are neighbouring 128-byte runs of the same channel row: stored back to back
        float* d0[2];
        bool ok[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            ok[nt] = KS == 3 || (y0 + row[nt] < p.OH && col[nt] < p.OW);                 // (4x4: the row / column beyond the valid output)
            d0[nt] = p.dst + (size_t)zsplit * p.slab_stride + (size_t)n * p.M * HW + (size_t)(y0 + row[nt]) * p.OW + col[nt] +
                     (size_t)(mblk * 128 + 4 * lhi) * HW;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float bv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (with_bias) {
                    const float* const bp = p.bias + mblk * 128 + 4 * lhi + mt * 32 + 8 * q;
                    b4 = p.bias_al ? *reinterpret_cast<const float4*>(bp) : make_float4(bp[0], bp[1], bp[2], bp[3]);
                }
                bv[4 * q] = b4.x; bv[4 * q + 1] = b4.y; bv[4 * q + 2] = b4.z; bv[4 * q + 3] = b4.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v = acc[mt][nt][r];
                    if (NPL == 2) v *= unscale;
                    v += bv[r];
                    if (ok[nt]) d0[nt][(size_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * HW] = v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    // ---- EPI: result [+ addend (the skip gradient)] -> dst, and the tile's max |result| into its partial word (max_words.h).  Four groups
    // of 32 elements (both 32-pixel halves of 16 channel rows), each [its addend loads][add, max, stores], fenced: at most 32 loads' worth of extra registers.  Element (mt, r) of a
    // lane lies a wave-uniform (32 mt + (r & 3) + 8 (r >> 2)) HW floats behind the lane's first one.
    unsigned omax = 0;
    size_t o0[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
        o0[nt] = (size_t)n * p.M * HW + (size_t)(y0 + row[nt]) * p.OW + col[nt] + (size_t)(mblk * 128 + 4 * lhi) * HW;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float ad[2][16];
        if (EPI == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) ad[nt][r] = p.addend[o0[nt] + (size_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * HW];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float v = acc[mt][nt][r] * unscale;
                if (EPI == 2) v += ad[nt][r];
                p.dst[o0[nt] + (size_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * HW] = v;
                omax = max(omax, finite_bits(__builtin_bit_cast(unsigned, v)));
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (p.maxw) {
        __shared__ unsigned mred[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
        if (lane == 0) mred[wid] = omax;
        __syncthreads();
        if (tid == 0) {
            const int parts = p.tiles_per_img * p.mblks;
            p.maxw[p.N + (size_t)n * parts + (ptile - n * p.tiles_per_img) * p.mblks + mblk] = max(max(mred[0], mred[1]), max(mred[2], mred[3]));
            if (p.maxw_lazy && ptile == n * p.tiles_per_img && mblk == 0) p.maxw[n] = NEMAR_MAX_LAZY_MARK | (unsigned)parts;
        }
    }
}

int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

}  // namespace

bool nemar_split16_eligible(int N, int H, int W, int M, int Cred, int R, int S, int stride, int pad, int mode, int variant) {
    if (R != S || (R != 3 && R != 4) || stride != 1 || pad != 1) return false;
    if (M % 128 != 0 || Cred % 16 != 0 || M <= 0 || Cred <= 0) return false;
    if (!(W == 32 || W == 64 || W == 128 || W == 256)) return false;
    const int RT = 256 / W;
    if (H % RT != 0 || H < 4) return false;
    if ((long long)N * Cred * (H + 4) * (W + 4) >= (1ll << 31) || N > 256) return false;
    if (R == 4 && (mode != SPLIT16_ZERO || variant == 0)) return false;       // the discriminator's 4x4 layers: zero padding only
    if (variant == 0) return false;       // (the first-generation kernel with loader waves is gone: tools/ history)
    // two halo buffers of 2 NPL regions + the 4-slot weight ring must fit the 152 KiB of LDS the kernel declares
    const int npl = variant == 3 ? 3 : 2;
    const int region16 = (RT + R - 1) * (W + 4) + (mode == SPLIT16_DGRAD_REFLECT ? 2 * (W + 4) : 0);
    return 2 * 2 * npl * region16 + 4 * 256 * npl <= 9728;
}

// reduction runs per tile: enough workgroups for the 256 CUs when the layer has few tiles (each run >= 4 chunks)
NEMAR_SWITCH(int, g_split16_ksplit_cap, 8);    // nemar_tune(39): most reduction runs per tile (1: never split — the tests' small shapes with the fused epilogue)
int nemar_split16_ksplit(int N, int H, int W, int M, int Cred) {
    const int tiles = N * (H / (256 / W)) * (M / 128), nchunks = Cred / 16;
    int ks = 1;
    while (tiles * ks * 2 <= 256 && nchunks % (ks * 2) == 0 && nchunks / (ks * 2) >= 4 && ks < g_split16_ksplit_cap) ks *= 2;
    return ks;
}

// planes + (ksplit > 1) the slabs of the split reduction
size_t nemar_split16_scratch_total(int N, int H, int W, int M, int Cred, int OH, int OW) {
    const int ks = nemar_split16_ksplit(N, H, W, M, Cred);
    return nemar_split16_scratch_bytes(N, Cred, H, W) + (ks > 1 ? (size_t)ks * N * M * OH * OW * sizeof(float) : 0);
}

size_t nemar_split16_scratch_bytes(int N, int Cred, int H, int W) {
    return (size_t)3 * N * (Cred / 8) * (H + 4) * (W + 4) * 16 + 16384;     // + slack for whole-KiB halo reads
}

size_t nemar_split16_pack_bytes(int M, int Cred, int KS) { return (size_t)(Cred / 16) * KS * KS * (M / 128) * 768 * 16 + 16384; }

namespace {
// the max words sit in the slack behind the planes / the packed weights
unsigned* scratch_max_word(void* scratch, int N, int Cred, int H, int W) {
    return (unsigned*)((char*)scratch + nemar_split16_scratch_bytes(N, Cred, H, W) - 1024);      // up to 256 per-sample words
}
unsigned* pack_max_word(void* packed, int M, int Cred, int KS) {
    return (unsigned*)((char*)packed + nemar_split16_pack_bytes(M, Cred, KS) - 64);
}

// nemar_absmax_hint: max |t| words the caller has already computed for tensors the next calls take as sources (count words: one per
// sample of the tensor, or 1 = one for the whole tensor)
thread_local const void* g_hint_tensor[4] = {nullptr, nullptr, nullptr, nullptr};      // (per calling thread, like the route note)
thread_local const unsigned* g_hint_word[4] = {nullptr, nullptr, nullptr, nullptr};
thread_local int g_hint_count[4] = {0, 0, 0, 0};
// nemar_planes_hint: the fp16 x 3 planes of a source already exist (a producer wrote them: norm_planes.hip) — reflect 3x3 layout only
// kind = the SPLIT16_* content the planes hold (SPLIT16_REFLECT: a forward producer's; SPLIT16_ZERO / SPLIT16_DGRAD_REFLECT: the
// data-gradient planes nemar_instnorm_bwd_planes writes)
struct PlanesHint { const void* tensor; const void* planes; int N, C, H, W, kind; };
thread_local PlanesHint g_planes_hint[2] = {{nullptr, nullptr, 0, 0, 0, 0, 0}, {nullptr, nullptr, 0, 0, 0, 0, 0}};
}  // namespace

void nemar_split16_set_planes_hint(const void* tensor, const void* planes, int N, int C, int H, int W, int kind) {
    int slot = -1;
    for (int i = 0; i < 2; ++i)
        if (g_planes_hint[i].tensor == tensor) slot = i;
    if (!planes) {
        if (slot >= 0) g_planes_hint[slot] = PlanesHint{nullptr, nullptr, 0, 0, 0, 0, 0};
        return;
    }
    if (slot < 0) slot = g_planes_hint[0].tensor ? 1 : 0;
    g_planes_hint[slot] = PlanesHint{tensor, planes, N, C, H, W, kind};
}

static const void* planes_hint_for(const void* tensor, int N, int C, int H, int W, int kind) {
    for (int i = 0; i < 2; ++i) {
        const PlanesHint& h = g_planes_hint[i];
        if (h.tensor == tensor && tensor && h.N == N && h.C == C && h.H == H && h.W == W && h.kind == kind) return h.planes;
    }
    return nullptr;
}

// the epilogue side inputs of the next nemar_split16_conv call on this thread (conv.hip sets them from nemar_conv_extras, one call)
thread_local const float* g_s16_addend = nullptr;
thread_local unsigned* g_s16_maxw = nullptr;
thread_local int g_s16_epilogue_done = 0;
void nemar_split16_set_epilogue(const float* addend, void* max_words) { g_s16_addend = addend; g_s16_maxw = (unsigned*)max_words; g_s16_epilogue_done = 0; }
int nemar_split16_epilogue_done() { return g_s16_epilogue_done; }

const unsigned* nemar_split16_hint(const void* tensor, int* count) {
    for (int i = 0; i < 4; ++i)
        if (g_hint_tensor[i] == tensor && tensor) {
            *count = g_hint_count[i];
            return g_hint_word[i];
        }
    return nullptr;
}

void nemar_split16_set_hint(const void* tensor, const void* word, int count) {
    int slot = -1;
    for (int i = 0; i < 4; ++i)
        if (g_hint_tensor[i] == tensor) slot = i;
    if (!word) {
        if (slot >= 0) { g_hint_tensor[slot] = nullptr; g_hint_word[slot] = nullptr; g_hint_count[slot] = 0; }
        return;
    }
    if (slot < 0)
        for (int i = 0; i < 4 && slot < 0; ++i)
            if (!g_hint_tensor[i]) slot = i;
    if (slot < 0) slot = 0;
    g_hint_tensor[slot] = tensor;
    g_hint_word[slot] = (const unsigned*)word;
    g_hint_count[slot] = count;
}

// out: `samples` words, zero on entry
void nemar_split16_absmax(const float* x, int samples, long long per, void* out, hipStream_t st) {
    int grid = nemar_stream_grid(per, 256 * 16);
    if (grid * samples > 2048) grid = (2048 + samples - 1) / samples;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid, samples), dim3(256), 0, st, x, per, (unsigned*)out);
}

// per-sample max |src| words for a split pass: the caller's hint (one word per sample, or one for all: *stride = 0), or computed
// here into `own` (N words)
const unsigned* nemar_split16_source_max(const float* src, int N, long long per, unsigned* own, int* stride, hipStream_t st) {
    int count = 0;
    const unsigned* h = nemar_split16_hint(src, &count);
    if (h && (count == N || count == 1)) {
        *stride = count == N && N > 1 ? 1 : 0;
        return h;
    }
    (void)hipMemsetAsync(own, 0, sizeof(unsigned) * N, st);
    nemar_split16_absmax(src, N, per, own, st);
    *stride = N > 1 ? 1 : 0;
    return own;
}

// ---- measurement hook (bench.py's roofline entry): HIP events around the main kernel of every 3x3 convolution call (the dominant
// instantiation: what rocprofv3 --stats lists as igemm_split16_kernel<2, 2, 3>) ----
namespace {
constexpr int MAX_TIMED = 1024;
hipEvent_t g_tev[MAX_TIMED][2];
int g_tev_made = 0, g_tev_used = 0;
double g_tev_flop = 0.0;          // algorithmic (fp32-equivalent) flop of the launches timed so far
bool g_timing = false;
}  // namespace

void nemar_split16_timer(int on) {
    g_timing = on != 0;
    if (on) { g_tev_used = 0; g_tev_flop = 0.0; }
}

int nemar_split16_timer_read(double* total_ms, double* total_flop) {
    double t = 0.0;
    for (int i = 0; i < g_tev_used; ++i) {
        float ms = 0.f;
        (void)hipEventSynchronize(g_tev[i][1]);
        (void)hipEventElapsedTime(&ms, g_tev[i][0], g_tev[i][1]);
        t += ms;
    }
    *total_ms = t;
    *total_flop = g_tev_flop;
    const int n = g_tev_used;
    g_tev_used = 0;
    g_tev_flop = 0.0;
    return n;
}

#define S16_TIMED_LAUNCH(launch_)                                              \
    {                                                                          \
        const bool tm_ = g_timing && KS == 3 && g_tev_used < MAX_TIMED;        \
        if (tm_) {                                                             \
            while (g_tev_made <= g_tev_used) {                                 \
                (void)hipEventCreate(&g_tev[g_tev_made][0]);                   \
                (void)hipEventCreate(&g_tev[g_tev_made][1]);                   \
                ++g_tev_made;                                                  \
            }                                                                  \
            (void)hipEventRecord(g_tev[g_tev_used][0], st);                    \
        }                                                                      \
        launch_;                                                               \
        if (tm_) {                                                             \
            (void)hipEventRecord(g_tev[g_tev_used++][1], st);                  \
            g_tev_flop += 2.0 * N * OH * OW * (double)M * Cred * KS * KS;      \
        }                                                                      \
    }

void nemar_split16_pack(const float* w, void* packed, int K, int C, int KS, int dgrad, int variant, hipStream_t st) {
    const int M = dgrad ? C : K, Cred = dgrad ? K : C, NT = KS * KS;
    const long long total = (long long)(Cred / 16) * NT * (M / 128) * 256;
    if (variant == 4) {
        unsigned* mw = pack_max_word(packed, M, Cred, KS);
        if (nemar_pack_recording()) {                        // (partial words: in the slack behind the image, 1 KiB before its end)
            unsigned* parts = (unsigned*)((char*)packed + nemar_split16_pack_bytes(M, Cred, KS) - 1024);
            NemarPackMaxArgs ma{w, (long long)K * C * NT, parts, NEMAR_PACK_MAX_PARTS, 1};
            nemar_pack_record_job(PACK_FAM_MAX, &ma, ma.gx, 1);
            Split16PackArgs a{w, (u32x4*)packed, M, Cred, dgrad, NT, parts, mw, nemar_stream_grid(total, 256), 1};
            nemar_pack_record_job(PACK_FAM_SPLIT16, &a, a.gx, 1);
        }
        (void)hipMemsetAsync(mw, 0, sizeof(unsigned), st);
        nemar_split16_absmax(w, 1, (long long)K * C * NT, mw, st);
        hipLaunchKernelGGL((split16_pack_kernel<2>), dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, w, (u32x4*)packed, M, Cred, dgrad, mw,
                           NT);
        return;
    }
    hipLaunchKernelGGL((split16_pack_kernel<3>), dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, w, (u32x4*)packed, M, Cred, dgrad,
                       (const unsigned*)nullptr, NT);
}

NEMAR_SWITCH(int, g_split16_ring3, 0);         // nemar_tune(32, 1): 3-slot weight ring (76.8 KB of LDS) for the unfolded 3x3 launches.  Measured: same time
                                 // (287.6 vs 292.3 us per call, bench 37.5 vs 37.2-37.5 ms) — the kernel needs 332 VGPRs (two fragment sets), so a
                                 // SIMD still holds ONE wave and the second workgroup never becomes resident; kept for the experiment only

bool nemar_split16_conv(const float* src, const void* packed, const float* bias, float* dst, int N, int H, int W, int M, int Cred,
                         int KS, int src_pad, int Hs, int Ws_src, int OH, int OW, int mode, void* scratch, int xcd_map, int variant,
                         long long* tl, void* dual_g_out, hipStream_t st) {
    bool dual_written = false;
    const long long total = (long long)N * (Cred / 8) * (H + 4) * (W + 4);
    unsigned* const xmw = scratch_max_word(scratch, N, Cred, H, W);
    const unsigned* xmax = xmw;
    int xstride = 0;
    const void* ready = nullptr;          // planes a producer already wrote (with the max-word hint they were scaled by)
    if (variant == 4 && KS == 3 && src_pad == 1 && Hs == H && Ws_src == W) {
        int count = 0;
        ready = planes_hint_for(src, N, Cred, H, W, mode);
        if (ready && !(nemar_split16_hint(src, &count) && count == N)) ready = nullptr;
    }
    if (variant == 4) {
        xmax = nemar_split16_source_max(src, N, (long long)Cred * Hs * Ws_src, xmw, &xstride, st);
        if (!ready && dual_g_out && KS == 3 && src_pad == 1 && Hs == H && Ws_src == W && Cred % 64 == 0 &&
            (mode == SPLIT16_DGRAD_REFLECT || mode == SPLIT16_ZERO))
            // data gradient of a 3x3 layer whose weight gradient follows: both operand layouts of gy from one read (conv_split16_wgrad.hip)
        {
            nemar_split16_dual_split(src, scratch, dual_g_out, N, Cred, H, W, mode, xmax, xstride, st);
            dual_written = true;
        } else if (!ready)
            hipLaunchKernelGGL((split_planes_kernel<2>), dim3(nemar_cdiv(total, 256)), dim3(256), 0, st, src, (u32x4*)scratch, N, Cred, H, W,
                               mode, total, xmax, xstride, src_pad, Hs, Ws_src);
    } else {
        hipLaunchKernelGGL((split_planes_kernel<3>), dim3(nemar_cdiv(total, 256)), dim3(256), 0, st, src, (u32x4*)scratch, N, Cred, H, W,
                           mode, total, (const unsigned*)nullptr, 0, src_pad, Hs, Ws_src);
    }
    Split16Params p;
    p.planes = ready ? (const u32x4*)ready : (const u32x4*)scratch;
    p.wp = (const u32x4*)packed;
    p.bias = bias;
    p.bias_al = (((uintptr_t)bias) & 15) == 0 ? 1 : 0;
    p.dst = dst;
    p.N = N; p.H = H; p.W = W; p.M = M; p.Cred = Cred;
    p.Ws = W + 4;
    p.HpWs = (H + 4) * (W + 4);
    p.wshift = ilog2(W);
    p.RT = 256 / W;
    p.tiles_per_img = H / p.RT;
    p.mblks = M / 128;
    p.halo_instr = nemar_cdiv((long long)(p.RT + KS - 1) * p.Ws * 16, 1024);
    p.fold = mode == SPLIT16_DGRAD_REFLECT;
    p.aux_instr = p.fold ? nemar_cdiv((long long)2 * p.Ws * 16, 1024) : 0;
    p.plane16 = total;
    p.tl = tl;
    p.xmax = xmax;
    p.xstride = xstride;
    p.wmax = pack_max_word(const_cast<void*>(packed), M, Cred, KS);
    p.OH = OH; p.OW = OW;
    p.halo16 = (p.RT + KS - 1) * p.Ws;
    p.aux16 = p.fold ? 2 * p.Ws : 0;
    const int tiles = N * p.tiles_per_img * p.mblks;
    p.ksplit = nemar_split16_ksplit(N, H, W, M, Cred);
    p.slab_stride = (long long)N * M * OH * OW;
    // fused epilogue (skip-gradient add, per-sample max of the result): only where a tile's whole reduction runs in one workgroup
    const bool fuse = p.ksplit == 1 && variant == 4 && KS == 3 && !bias && (g_s16_addend || g_s16_maxw);
    p.addend = fuse ? g_s16_addend : nullptr;
    p.maxw = fuse ? g_s16_maxw : nullptr;
    p.maxw_lazy = (p.maxw && nemar_max_words_lazy() && p.tiles_per_img * p.mblks <= 0xFFFF) ? 1 : 0;
    g_s16_epilogue_done = fuse ? 1 : 0;
    float* const final_dst = dst;
    if (p.ksplit > 1) p.dst = (float*)((char*)scratch + nemar_split16_scratch_bytes(N, Cred, H, W));       // slabs behind the planes
    const int grid = tiles * p.ksplit;
    p.xcd = (xcd_map && grid % 8 == 0 && (grid / 8) % (p.mblks * p.ksplit) == 0) ? 1 : 0;
    const int region = p.halo_instr + p.aux_instr;
    const dim3 g(grid);
    // dynamic LDS: [RING weight stages][two halo buffers]; above 64 KiB the attribute is needed (nemar_lds_bytes sets it once per instantiation)
#ifdef NEMAR_HOST_EMULATION
#define S16_GO_(NBW_, NPL_, KS_, RING_, EPI_) { hipLaunchKernelGGL((igemm_split16_kernel<NBW_, NPL_, KS_, RING_, EPI_>), g, dim3(256), 0, st, p); }
#else
#define S16_GO_(NBW_, NPL_, KS_, RING_, EPI_)                                                                           \
    {                                                                                                                   \
        const size_t need_ = ((size_t)(RING_) * 256 * (NPL_) + (size_t)2 * 2 * (NPL_) * (p.halo16 + p.aux16)) * 16;     \
        const size_t lds_ = nemar_lds_bytes(reinterpret_cast<const void*>(&igemm_split16_kernel<NBW_, NPL_, KS_, RING_, EPI_>), need_,      \
                                            (g_lds_claim & 2) != 0);                 /* (whole-CU claim: common.h) */   \
        hipLaunchKernelGGL((igemm_split16_kernel<NBW_, NPL_, KS_, RING_, EPI_>), g, dim3(256), lds_, st, p);            \
    }
#endif
#define S16_GO(NBW_, NPL_, KS_, RING_) S16_GO_(NBW_, NPL_, KS_, RING_, 0)
    if (variant == 4) {                 // fp16 x 3 (nemar_split16_eligible checked the LDS budget)
        const int ipr = nemar_cdiv(p.halo16, 64) + nemar_cdiv(p.aux16, 64);
        const int nbw = nemar_cdiv(nemar_cdiv(4 * ipr, 4), KS * KS - 3);
        // 3-slot weight ring where that lets two workgroups share a CU (<= 80 KiB each: 3x3 layers without the folded border rows)
        NEMAR_AB_ONLY(const bool ring3 = g_split16_ring3 && KS == 3 && ((size_t)3 * 512 + (size_t)8 * (p.halo16 + p.aux16)) * 16 <= 80 * 1024 - 512;)
        S16_TIMED_LAUNCH(
            if (KS == 4) {
                if (nbw <= 1) S16_GO(1, 2, 4, 4)
                else S16_GO(2, 2, 4, 4)
            }
            NEMAR_AB_ONLY(else if (ring3) {
                if (nbw <= 1) S16_GO(1, 2, 3, 3)
                else if (nbw == 2) S16_GO(2, 2, 3, 3)
                else S16_GO(3, 2, 3, 3)
            })
            else if (fuse && p.addend && nbw <= 1) S16_GO_(1, 2, 3, 4, 2)
            else if (fuse && p.addend && nbw == 2) S16_GO_(2, 2, 3, 4, 2)
            else if (fuse && p.addend) S16_GO_(3, 2, 3, 4, 2)
            else if (fuse && nbw <= 1) S16_GO_(1, 2, 3, 4, 1)
            else if (fuse && nbw == 2) S16_GO_(2, 2, 3, 4, 1)
            else if (fuse) S16_GO_(3, 2, 3, 4, 1)
            else if (nbw <= 1) S16_GO(1, 2, 3, 4)
            else if (nbw == 2) S16_GO(2, 2, 3, 4)
            else S16_GO(3, 2, 3, 4))
        if (p.ksplit > 1) nemar_sum_partials(p.dst, p.slab_stride, p.ksplit, final_dst, p.slab_stride, false, st);
        if (p.maxw && !p.maxw_lazy) max_words_finalize(p.maxw, N, p.tiles_per_img * p.mblks, st);
        return dual_written;
    }
#ifdef NEMAR_AB      // nemar_tune(21, 3): bf16 x 6 products
    if (variant == 3) {
        const int ipr = nemar_cdiv(p.halo16, 64) + nemar_cdiv(p.aux16, 64);
        const int nbw = nemar_cdiv(nemar_cdiv(6 * ipr, 4), KS * KS - 3);
        if (KS == 4) {
            if (nbw <= 1) S16_GO(1, 3, 4, 4)
            else S16_GO(2, 3, 4, 4)
        } else if (nbw <= 2) S16_GO(2, 3, 3, 4)
        else S16_GO(3, 3, 3, 4)
        if (p.ksplit > 1) nemar_sum_partials(p.dst, p.slab_stride, p.ksplit, final_dst, p.slab_stride, false, st);
        return dual_written;
    }
#endif
#undef S16_GO
#undef S16_GO_
    (void)region;
    return dual_written;
}
