// TEST INFRASTRUCTURE ONLY — host-side SIMT emulator for the gfx950 kernels.
//
// The kernel sources under nemar_amd/csrc/*.hip are pure HIP.  The CPU-only test
// tier (`pytest -m "not gpu"`) compiles those same, unmodified sources with the
// host clang++ and THIS header shadowing <hip/hip_runtime.h>, so that index
// arithmetic, LDS tiling, MFMA fragment layouts and wave collectives can be
// checked against the oracle without a GPU.  Every GPU thread is a ucontext
// fiber; __syncthreads and wave collectives are rendezvous points.  Nothing in
// the product path (nemar_amd/) includes, links or loads this.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <tuple>
#include <utility>

#define NEMAR_HOST_EMULATION 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...)   /* __attribute__((amdgpu_waves_per_eu(a, b))) -> __attribute__(()) */
#define __constant__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
extern emu_uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
typedef void* hipEvent_t;
static const unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)1; return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)1; return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)1; return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }      // the emulator runs launches synchronously
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static const int hipMemcpyDeviceToDevice = 3, hipMemcpyHostToDevice = 1;

// ---- vector types -------------------------------------------------------------------------
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

// ---- runtime hooks (tests/emu/emu_runtime.cpp) -----------------------------------------------
void emu_sync_block();
// deposit `bytes` (<=64) from every live lane of the wave, return all 64 slots
void emu_wave_exchange(const void* mine, unsigned bytes, void* all64);
int emu_lane_id();
void emu_run_grid(dim3 grid, dim3 block, void (*thunk)(void*), void* ctx);

static inline void __syncthreads() { emu_sync_block(); }

template <typename T>
static inline T emu_shfl_from(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl type too wide");
    T all[64];
    emu_wave_exchange(&v, sizeof(T), all);
    return all[src & 63];
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = emu_lane_id();
    int base = lane & ~(width - 1);
    return emu_shfl_from(v, base + (src & (width - 1)));
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    return emu_shfl_from(v, emu_lane_id() ^ mask);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = emu_lane_id();
    int src = lane + (int)d;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return emu_shfl_from(v, src);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = emu_lane_id();
    int src = lane - (int)d;
    if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return emu_shfl_from(v, src);
}
static inline unsigned long long __ballot(int pred) {
    int all[64]; int p = pred ? 1 : 0;
    emu_wave_exchange(&p, sizeof(int), all);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) if (all[i]) m |= (1ull << i);
    return m;
}
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p |= v; return o; }
static inline int __all(int pred) { return __ballot(pred) == ~0ull; }
static inline int __any(int pred) { return __ballot(pred) != 0ull; }

// ---- MFMA (f32 in / f32 acc), bit-exact k-ordered fmaf chain per the CDNA4 guide ----------------
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
// 32x32x2: lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D reg r of lane l: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
    float ab[2] = {a, b};
    float all[64][2];
    emu_wave_exchange(ab, sizeof(ab), all);
    int l = emu_lane_id();
    int col = l & 31;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(all[row + 32 * k][0], all[col + 32 * k][1], acc);
        d[r] = acc;
    }
    return d;
}
// 16x16x4: lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r: col=l&15, row=(l>>4)*4+r
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
    float ab[2] = {a, b};
    float all[64][2];
    emu_wave_exchange(ab, sizeof(ab), all);
    int l = emu_lane_id();
    int col = l & 15;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(all[row + 16 * k][0], all[col + 16 * k][1], acc);
        d[r] = acc;
    }
    return d;
}


// 32x32x16 bf16: lane l holds A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][j'=l&31], j = 0..7; D as the f32 32x32 form.
// Products of two bf16 are exact in fp32; the 16 of them are added to the accumulator in k order (the hardware's internal
// order is not documented: the tests that run through this compare with a tolerance, not bitwise).
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c, int, int, int) {
    struct AB { float a[8], b[8]; } mine, all[64];
    for (int j = 0; j < 8; ++j) { mine.a[j] = (float)a[j]; mine.b[j] = (float)b[j]; }
    emu_wave_exchange(&mine, sizeof(mine), all);
    int l = emu_lane_id();
    int col = l & 31;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 8; ++j) acc += all[row + 32 * g].a[j] * all[col + 32 * g].b[j];
        d[r] = acc;
    }
    return d;
}


// 32x32x16 f16: same operand / result maps as the bf16 form
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x16 c, int, int, int) {
    struct AB { float a[8], b[8]; } mine, all[64];
    for (int j = 0; j < 8; ++j) { mine.a[j] = (float)a[j]; mine.b[j] = (float)b[j]; }
    emu_wave_exchange(&mine, sizeof(mine), all);
    int l = emu_lane_id();
    int col = l & 31;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 8; ++j) acc += all[row + 32 * g].a[j] * all[col + 32 * g].b[j];
        d[r] = acc;
    }
    return d;
}

// ---- direct global->LDS copy (global_load_lds_*): LDS destination = wave-uniform base + lane*size ------------------
static inline void __builtin_amdgcn_global_load_lds(const __attribute__((address_space(1))) void* g,
                                                    __attribute__((address_space(3))) void* l, unsigned size,
                                                    int offset, unsigned /*aux*/) {
    memcpy((char*)(void*)l + (size_t)emu_lane_id() * size + offset, (const void*)g, size);
}
static inline void __builtin_amdgcn_s_waitcnt(int) {}

// ---- v_permlane32_swap_b32 (gfx950): lanes 32..63 of vdst <-> lanes 0..31 of src; returns {new vdst, new src} -------------
typedef unsigned emu_u32x2 __attribute__((ext_vector_type(2)));
static inline emu_u32x2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned src, bool, bool) {
    unsigned mine[2] = {vdst, src};
    unsigned all[64][2];
    emu_wave_exchange(mine, sizeof(mine), all);
    const int l = emu_lane_id();
    emu_u32x2 r;
    r[0] = l < 32 ? vdst : all[l - 32][1];
    r[1] = l < 32 ? all[l + 32][0] : src;
    return r;
}
// v_alignbit_b32: low 32 bits of ({hi, lo} >> shift)
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) {
    return (unsigned)((((unsigned long long)hi << 32) | lo) >> (sh & 31));
}

// ---- scalar/uniform builtins ------------------------------------------------------------------
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
static inline void __builtin_amdgcn_s_barrier() { emu_sync_block(); }

// ---- atomics (single OS thread => plain RMW) ------------------------------------------------------
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline long long __float2ll_rn(float f) { return (long long)llrintf(f); }
static inline float unsafeAtomicAdd(float* p, float v) { return atomicAdd(p, v); }
static inline unsigned atomicInc(unsigned* p, unsigned lim) { unsigned o = *p; *p = (o >= lim) ? 0 : o + 1; return o; }

// ---- math ---------------------------------------------------------------------------------------
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
using std::min;
using std::max;
static inline void __threadfence() {}
static inline long long clock64() { return 0; }

// ---- launch ---------------------------------------------------------------------------------------
template <typename K, typename Tup, size_t... I>
static inline void emu_apply(K k, Tup& t, std::index_sequence<I...>) { k(std::get<I>(t)...); }

template <typename... KArgs, typename... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*shmem*/,
                                      hipStream_t /*stream*/, Args... args) {
    std::tuple<KArgs...> packed(static_cast<KArgs>(args)...);
    struct Ctx { void (*k)(KArgs...); std::tuple<KArgs...>* t; } ctx{kernel, &packed};
    emu_run_grid(grid, block,
                 [](void* p) {
                     Ctx* c = (Ctx*)p;
                     emu_apply(c->k, *c->t, std::index_sequence_for<KArgs...>{});
                 },
                 &ctx);
}
