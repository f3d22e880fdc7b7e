// nemar_amd — shared host/device helpers for the gfx950 kernels behind include/nemar_hip.h.
//
// Conventions: the caller owns every buffer, kernels never allocate or
// synchronise, every launch goes on the stream handed in, tensors are contiguous NCHW fp32.
// Entry points return 0 or a negative NEMAR_E* code and leave a message for nemar_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define NEMAR_OK 0
#define NEMAR_EINVAL (-1)   // bad shape / null pointer / unsupported combination
#define NEMAR_ELAUNCH (-2)  // HIP reported a launch error
#define NEMAR_EWORKSPACE (-3)  // workspace too small

#define NEMAR_API extern "C" __attribute__((visibility("default")))

// Measurement switches.  The library is built twice from the same sources (build.py):
//   * libnemar_hip.so      (product): every switch is a compile-time constant at its default; kernels and launch branches that only
//                          a non-default value reaches are not compiled; nemar_tune / nemar_tune_ptr / nemar_grid_sample_tune do
//                          not exist (include/nemar_hip.h has no such entry point);
//   * libnemar_hip_ab.so   (-DNEMAR_AB; include/nemar_hip_ab.h): the switches are variables behind nemar_tune, for tools/ and the
//                          A/B tests that drive small shapes through a chosen route.
#ifdef NEMAR_AB
#define NEMAR_SWITCH(type, name, def) type name = def
#define NEMAR_AB_ONLY(...) __VA_ARGS__
#else
#define NEMAR_SWITCH(type, name, def) __attribute__((unused)) constexpr type name = def
#define NEMAR_AB_ONLY(...)
#endif

// Whole-CU LDS claim.  wgrad_split16_kernel with its operands staged by LDS-DMA (one workgroup per CU) reads wrong X
// fragments — rarely: ~3e-3 of its launches on the worst box — while an LDS-ACTIVE workgroup of ANOTHER kernel shares its CU (a second
// HIP stream): measured beside split_dual_kernel and beside a 1 KiB ds_read / ds_write test kernel (tools/diag_wgrad_beside.py, 60 000
// launches per cell), with any wait / barrier protocol inside the kernel, __syncthreads() per step included; never alone on its CU,
// never beside workgroups without LDS traffic, and never when the same pieces travel global -> registers -> ds_write (0 of 240 000).
// Two remedies: the 3x3 form of the kernel stages through registers (conv_split16_wgrad.hip, XREG: the default); a form that keeps
// LDS-DMA (the 4x4 layer's, one launch per step) launches with as much dynamic LDS as it takes to fill the CU's 160 KiB, so that no
// other kernel's LDS-using workgroup can be placed beside it (0 of 240 000 launches, 0 of 2 166 training steps).  The other LDS-DMA
// staged kernels were measured the same way WITHOUT a claim — igemm_split16_kernel and s16g_kernel forward calls as the victim:
// 0 of 240 000 — and do not claim (a claim costs what it keeps off the CU: 0.7 - 0.9 ms of a 30 ms step for the 3x3 weight gradient).
// nemar_lds_bytes: the dynamic LDS size to launch `kernel` with — `need` bytes, or what fills the CU beside the kernel's static LDS when
// `claim` — and, once per kernel, the attribute that allows more than 64 KiB.
size_t nemar_lds_bytes(const void* kernel, size_t need, bool claim);
// which kernel families claim (bit mask, nemar_tune(37) in the measurement build): 1 wgrad_split16_kernel's LDS-DMA forms, 2 igemm_split16_kernel,
// 4 s16g_kernel
#define NEMAR_LDS_CLAIM_DEFAULT 1
#ifdef NEMAR_AB
extern int g_lds_claim;      // nemar_tune(37, mask)
#else
constexpr int g_lds_claim = NEMAR_LDS_CLAIM_DEFAULT;
#endif

// include/nemar_hip.h's nemar_conv_extras (the per-call side inputs of nemar_conv2d_*_ex), field for field
struct nemar_conv_extras {
    void* scratch;
    size_t scratch_bytes;
    const void* src_max_words;
    int src_max_count;
    const void* src2_max_words;
    int src2_max_count;
    const void* src_planes;
    void* gy_planes_out;
    size_t gy_planes_bytes;
    const void* src2_planes;
    const float* addend;
    void* out_max_words;
    const float* bias_partials;
};

// thread-local so the message survives being raised on autograd's backward thread
void nemar_set_error(const char* fmt, ...);

#define NEMAR_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            nemar_set_error(__VA_ARGS__);        \
            return NEMAR_EINVAL;                 \
        }                                        \
    } while (0)

// hipGetLastError() is per-thread sticky state shared with every other HIP user in the process (torch, rocBLAS...):
// discard whatever is pending before our launches so that NEMAR_CHECK_LAUNCH reports only our own failures.
#define NEMAR_CLEAR_HIP_ERROR() ((void)hipGetLastError())

#define NEMAR_CHECK_LAUNCH(what)                                              \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) {                                              \
            nemar_set_error("%s: %s", what, hipGetErrorString(e__));          \
            return NEMAR_ELAUNCH;                                             \
        }                                                                     \
    } while (0)

#define NEMAR_HIP_CALL(expr)                                                 \
    do {                                                                      \
        hipError_t e__ = (expr);                                              \
        if (e__ != hipSuccess) {                                              \
            nemar_set_error("%s: %s", #expr, hipGetErrorString(e__));         \
            return NEMAR_ELAUNCH;                                             \
        }                                                                     \
    } while (0)

static inline int nemar_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Memory-bound kernels: cap the grid at 256 CUs x 8 workgroups and grid-stride the rest.
static inline int nemar_stream_grid(long long work_items, int per_block) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;
    return (int)b;
}

// Exact unsigned division by a launch-constant divisor d (n*d < 2^40), without a VALU divide:
// q = (n * M) >> 40 with M = ceil(2^40 / d).  Host builds it, device applies it.
struct FastDiv {
    unsigned d;
    unsigned long long m;
};
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d ? d : 1;
    f.m = ((1ull << 40) + f.d - 1) / f.d;
    return f;
}
__device__ __forceinline__ unsigned fd_div(unsigned n, const FastDiv& f) {
    return (unsigned)(((unsigned long long)n * f.m) >> 40);
}

// Direct global -> LDS copies (no VGPR staging): each lane supplies its own global address, the LDS destination is the
// wave-uniform `lds` base + lane * size.  Asynchronous: complete after s_waitcnt vmcnt(0) (+ a barrier for other waves).
__device__ __forceinline__ void glds_b128(const float* g, float* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ void glds_b32(const float* g, float* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 4, 0, 0);
}
__device__ __forceinline__ void wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // s_waitcnt vmcnt(0)

// a value every lane of the wave holds identically -> a scalar register (frees the vector register; the emulator has no scalar file)
__device__ __forceinline__ float uniform_f(float v) {
#ifdef NEMAR_HOST_EMULATION
    return v;
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
#endif
}

// 64-lane wavefront reductions (gfx950 wave = 64; never 32)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<=1024).  `red` is >=16 floats of LDS.
// Result valid in every thread.  Deterministic (fixed tree).
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();  // protect `red` from a previous use
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
