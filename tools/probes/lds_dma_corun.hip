// Stand-alone probe (no library, no torch): does a workgroup that stages operands by LDS-DMA (global_load_lds_dwordx4) read wrong words from
// LDS — after its waves' counted vmcnt waits and a workgroup barrier — while an LDS-active workgroup of ANOTHER kernel shares its CU?
// (wgrad_split16_kernel's LDS-DMA form did, ~3e-3 of its launches beside split_dual_kernel or a 1 KiB ds_read / ds_write kernel;
// tools/diag_wgrad_beside.py is the repro on the library.  This probe rebuilds that kernel's staging skeleton — 4 waves, a 4-slot ring of
// 16 KiB stages, every wave issuing four 1 KiB copies per step, eight ds_read_b128 fragment reads per wave per step, 27 MFMAs per step — around
// SELF-CHECKING data: every 32-bit word of the source is a function of its index, so each lane knows what every fragment must hold.)
//
//   hipcc --offload-arch=gfx950 -O2 -o lds_dma_corun lds_dma_corun.hip
//   ./lds_dma_corun [launches = 4000] [aggressor: 0 none, 1 the 1 KiB LDS kernel, 2 a streaming copy without LDS] [staging: 0 LDS-DMA, 1 registers + ds_write]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ inline unsigned word_of(unsigned long long i) { return (unsigned)(i * 2654435761ull) ^ (unsigned)(i >> 7) ^ 0x9E3779B9u; }

constexpr int RING = 4, NCOL = 16, NCP = 4, STAGE16 = NCOL * 64;      // u32x4 words per stage: 16 pieces of 64 lanes
constexpr int NSTEPS = 72;

// source layout: pieces 0..3 (the "G" pieces, wave 0): [workgroup][piece][stage][64 lanes] u32x4, every word fetched once.  Pieces 4..15 (the "X"
// pieces, waves 1..3 = tap rows r = 0..2): [workgroup][sub = plane, chunk][row = stage + ROWSTEP r][64 lanes] — as in the weight gradient, the row a
// wave fetches as tap row r at stage T is fetched again as row r - 1 ROWSTEP stages later by its neighbour: the same lines, three times, by three waves.
constexpr int ROWSTEP = 4;
constexpr unsigned long long XBASE = 256ull * 4 * NSTEPS * 64;      // behind the G pieces of all workgroups
__device__ __forceinline__ unsigned long long src_index(int wg, int piece, int stage, int lane) {
    if (piece < 4) return (((unsigned long long)wg * 4 + piece) * NSTEPS + stage) * 64 + lane;
    const int r = (piece - 4) >> 2, sub = (piece - 4) & 3;
    return XBASE + (((unsigned long long)wg * 4 + sub) * (NSTEPS + 2 * ROWSTEP) + stage + ROWSTEP * r) * 64 + lane;
}

__device__ __forceinline__ void glds16(const u32x4* g, u32x4* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

struct Report { unsigned bad, step, piece, lane, got, want, wg, wave; };

template <bool XREG>
__global__ __launch_bounds__(256) void victim_kernel(const u32x4* __restrict__ S, Report* rep, float* sink) {
    __shared__ __attribute__((aligned(16))) u32x4 smem[RING * STAGE16];
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, wg = blockIdx.x;
    const int l31 = lane & 31, lhi = lane >> 5, wk = wid >> 1, wc = wid & 1;
    const u32x4* csrc[NCP];
#pragma unroll
    for (int q = 0; q < NCP; ++q) csrc[q] = S + src_index(wg, NCP * wid + q, 0, lane);
#define VMCNT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14));
    f32x16 acc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    u32x4 fr[2][8];                                    // two register sets of the 8 fragments (2 A + 6 B)
    u32x4 xr[2][NCP];
    unsigned bad = 0, b_step = 0, b_piece = 0, b_got = 0, b_want = 0;
    if (XREG) {
#pragma unroll
        for (int stg = 0; stg < 2; ++stg)
#pragma unroll
            for (int q = 0; q < NCP; ++q) smem[stg * STAGE16 + (NCP * wid + q) * 64 + lane] = csrc[q][(size_t)stg * 64];
#pragma unroll
        for (int stg = 2; stg < 4; ++stg)
#pragma unroll
            for (int q = 0; q < NCP; ++q) xr[stg & 1][q] = csrc[q][(size_t)stg * 64];
    } else {
#pragma unroll
        for (int stg = 0; stg < 4; ++stg)
#pragma unroll
            for (int q = 0; q < NCP; ++q) glds16(csrc[q] + (size_t)stg * 64, smem + stg * STAGE16 + (NCP * wid + q) * 64);
        VMCNT(2 * NCP)
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    const int a_off = lhi * 64 + wk * 32 + l31, b_off = 4 * 64 + lhi * 64 + wc * 32 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i) fr[0][i] = smem[a_off + i * 128];
#pragma unroll
    for (int i = 0; i < 6; ++i) fr[0][2 + i] = smem[b_off + i * 128];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    for (int T0 = 0; T0 < NSTEPS; T0 += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int T = T0 + half, cur = half, nxt = half ^ 1;
            __builtin_amdgcn_sched_barrier(0);
            u32x4* const d2 = smem + ((T + 2) & (RING - 1)) * STAGE16 + wid * (NCP * 64);
            const u32x4* const Sl = smem + ((T + 1) & (RING - 1)) * STAGE16;
            // fragments of step T + 1 (stage T + 1), interleaved with this step's MFMAs on the fragments of step T
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (XREG && i < NCP) d2[i * 64 + lane] = xr[cur][i];
                fr[nxt][i] = i < 2 ? Sl[a_off + i * 128] : Sl[b_off + (i - 2) * 128];
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[cur][i & 1]), __builtin_bit_cast(f16x8, fr[cur][2 + (i + m) % 6]),
                                                                    acc[m], 0, 0, 0);
            }
            {
                const int st = min(T + 4, NSTEPS - 1);
                u32x4* const d = smem + (T & (RING - 1)) * STAGE16 + wid * (NCP * 64);
#pragma unroll
                for (int q = 0; q < NCP; ++q) {
                    if (XREG) xr[cur][q] = csrc[q][(size_t)st * 64];
                    else glds16(csrc[q] + (size_t)st * 64, d + q * 64);
                    acc[q % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[cur][q & 1]), __builtin_bit_cast(f16x8, fr[cur][2 + q]), acc[q % 3], 0, 0, 0);
                }
            }
            if (!XREG) VMCNT(2 * NCP)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            // the check: fragment i of step T + 1 = piece (i < 2 ? 2 i + lhi : 4 + 2 (i - 2) + lhi), lane (i < 2 ? wk : wc) * 32 + l31, stage T + 1
            if (T + 1 < NSTEPS) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int piece = i < 2 ? 2 * i + lhi : 4 + 2 * (i - 2) + lhi, ln = (i < 2 ? wk : wc) * 32 + l31;
                    const unsigned long long w = src_index(wg, piece, T + 1, ln) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned want = word_of(w + e), got = fr[nxt][i][e];
                        if (got != want && !bad) { bad = 1; b_step = T + 1; b_piece = piece; b_got = got; b_want = want; }
                    }
                }
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (bad) {
        const unsigned n = atomicAdd(&rep->bad, 1u);
        if (n == 0) { rep->step = b_step; rep->piece = b_piece; rep->lane = lane; rep->got = b_got; rep->want = b_want; rep->wg = wg; rep->wave = wid; }
    }
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[a][e];
    if (t == 1234.5678f) sink[tid] = t;
}

__global__ __launch_bounds__(256) void fill_kernel(u32x4* S, unsigned long long n) {
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        u32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = word_of(i * 4 + e);
        S[i] = v;
    }
}

template <int WORDS>
__global__ __launch_bounds__(256) void agg_lds_kernel(float* out, int iters) {
    __shared__ float tile[WORDS];
    const int t = threadIdx.x;
    for (int i = t; i < WORDS; i += 256) tile[i] = (float)(i + blockIdx.x);
    __syncthreads();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int j = 0; j < 8; ++j) {
            const int a = (t * 33 + j * 257 + it) % WORDS;
            acc += tile[a];
            tile[(a + 64) % WORDS] = acc;
        }
        __syncthreads();
    }
    if (acc == 12345.678f) out[blockIdx.x * 256 + t] = acc;
}

__global__ __launch_bounds__(256) void agg_copy_kernel(const float4* in, float4* out, long n) {
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += gridDim.x * 256l) out[i] = in[i];
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 4000, aggressor = argc > 2 ? atoi(argv[2]) : 1, xreg = argc > 3 ? atoi(argv[3]) : 0;
    const int WGS = 256;
    const unsigned long long n16 = XBASE + (unsigned long long)WGS * 4 * (NSTEPS + 2 * ROWSTEP) * 64;
    u32x4* S;
    Report* rep;
    float *sink, *a, *b;
    CK(hipMalloc(&S, n16 * 16)); CK(hipMalloc(&rep, sizeof(Report))); CK(hipMalloc(&sink, 4096));
    const long copy16 = 4l << 20;
    CK(hipMalloc(&a, copy16 * 16)); CK(hipMalloc(&b, copy16 * 16));
    CK(hipMemset(rep, 0, sizeof(Report)));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, S, n16);
    CK(hipDeviceSynchronize());
    hipStream_t s_main, s_side;
    CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));
    for (int it = 0; it < launches; ++it) {
        if (aggressor == 1) for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((agg_lds_kernel<256>), dim3(2048), dim3(256), 0, s_main, sink, 40);
        if (aggressor == 2) hipLaunchKernelGGL(agg_copy_kernel, dim3(2048), dim3(256), 0, s_main, (const float4*)a, (float4*)b, copy16);
        if (xreg) hipLaunchKernelGGL((victim_kernel<true>), dim3(WGS), dim3(256), 0, s_side, S, rep, sink);
        else hipLaunchKernelGGL((victim_kernel<false>), dim3(WGS), dim3(256), 0, s_side, S, rep, sink);
        if ((it & 63) == 63) CK(hipDeviceSynchronize());
    }
    CK(hipDeviceSynchronize());
    Report h;
    CK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
    printf("staging %s, aggressor %d, %d launches x %d workgroups x %d steps: %u lanes saw a wrong word after the waits", xreg ? "registers + ds_write" : "LDS-DMA", aggressor,
           launches, WGS, NSTEPS, h.bad);
    if (h.bad) printf("  (first: workgroup %u wave %u lane %u, stage %u piece %u, got %08x want %08x)", h.wg, h.wave, h.lane, h.step, h.piece, h.got, h.want);
    printf("\n");
    return 0;
}
