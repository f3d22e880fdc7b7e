// Minimal stand-alone repro (no library, no torch) of the defect behind the round-4 "lost store" anomaly:
// a wave executing packed-FP32 VOP3P instructions (v_pk_add_f32 / v_pk_mul_f32 with op_sel / neg modifiers: what hipcc's SLP
// vectoriser makes of two parallel scalar float chains) returns WRONG results in the low half of lanes 48..63 while waves of ANOTHER
// kernel, resident on the same SIMD through a second HIP stream, execute ... <which instructions: this probe finds out>.
//
// Victims (stream 1, short launches, results compared bit for bit with the same launch run alone):
//   pk      the instruction sequence of grid_sample_bwd_kernel<UNET,false>'s channel loop, inline asm (7 packed ops per channel)
//   pknop   the same with `s_nop 1` after every packed op
//   scalar  the same arithmetic with scalar v_sub / v_mul / v_add
// Triggers (stream 2, ~two workgroups per CU, long-running):
//   0 none   1 mfma: back-to-back v_mfma_f32_32x32x16_f16   2 pk: back-to-back v_pk_mul_f32   3 mfma + pk (accumulator rescale)
//   4 valu: scalar v_mul_f32 stream   5 lds traffic + barriers
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pk_f32_corun pk_f32_corun.hip
//   ./pk_f32_corun [iterations]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// in: planes a, b, c, d, g of channel 0..2 (15 planes), then ey, ex, ty, tx (4 planes); out: 2 planes (lo, hi accumulators)
// single-instruction victims: a dependent chain of ONE instruction form, 3 x 7 deep like the sequence above; out = both halves of the result
//   3 v_pk_mul_f32 (no modifiers)   4 v_pk_add_f32 (no modifiers)   5 v_pk_fma_f32   6 v_fma_mix_f32 (VOP3P encoding, scalar result)
//   7 v_pk_add_f16   8 v_pk_mul_f32 with op_sel_hi:[0,1]   9 v_mul_f32 x 2 (scalar pair)
template <int FORM>
__global__ __launch_bounds__(256) void victim1(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    f32x2 acc = {in[i], in[n + i]};
    for (int c = 0; c < 21; ++c) {
        const f32x2 m = {1.f + 0.01f * in[(2 + (c % 17)) * n + i], 1.f - 0.01f * in[(2 + ((c + 5) % 17)) * n + i]};
        if (FORM == 3) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc) : "v"(acc), "v"(m));
        if (FORM == 4) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc) : "v"(acc), "v"(m));
        if (FORM == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=v"(acc) : "v"(acc), "v"(m));
        if (FORM == 6) {
            asm volatile("v_fma_mix_f32 %0, %1, %2, %1" : "=v"(acc.x) : "v"(acc.x), "v"(m.x));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %1" : "=v"(acc.y) : "v"(acc.y), "v"(m.y));
        }
        if (FORM == 7) { asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(acc.x) : "v"(acc.x), "v"(m.x)); asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(acc.y) : "v"(acc.y), "v"(m.y)); }
        if (FORM == 8) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "v"(m), "v"(acc));
        if (FORM == 9) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(acc.x) : "v"(acc.x), "v"(m.x)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(acc.y) : "v"(acc.y), "v"(m.y)); }
        //   10 v_pk_add_f32 op_sel_hi:[1,0] (src1's low half for both results)   11 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[0,0] (src1's halves SWAPPED)
        //   12 v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1] only   13 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0]   14 v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] (src0 swapped)
        if (FORM == 10) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(acc) : "v"(acc), "v"(m));
        if (FORM == 11) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(acc) : "v"(m), "v"(acc));
        if (FORM == 12) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(acc) : "v"(acc), "v"(m));
        if (FORM == 13) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(acc) : "v"(m), "v"(acc));
        if (FORM == 14) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(acc) : "v"(acc), "v"(m));
        //   the other VOP3P forms the library's kernels contain (nemar_amd/csrc/isa_scan.py census): the fp32 -> 2 x fp16 split
        //   15 v_fma_mixlo_f16 d, a, b, -c op_sel_hi:[0,0,1]   16 v_fma_mixhi_f16 (same operands)   17 v_pk_mov_b32 op_sel:[1,0]   18 v_cvt_pk_f16_f32
        if (FORM == 15 || FORM == 16) {
            unsigned h = __builtin_bit_cast(unsigned, acc.y);
            if (FORM == 15) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(h) : "v"(acc.x), "v"(m.x), "v"(h));
            else asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(h) : "v"(acc.x), "v"(m.x), "v"(h));
            acc.y = __builtin_bit_cast(float, h ^ 0x00010001u);
            acc.x = acc.x * m.y;
        }
        if (FORM == 17) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(acc) : "v"(acc), "v"(m));
        if (FORM == 18) {
            unsigned h;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(acc.x), "v"(m.y));
            acc.y = __builtin_bit_cast(float, h);
            acc.x = acc.x * m.x;
        }
    }
    out[i] = acc.x;
    out[n + i] = acc.y;
}

template <int MODE>      // 0 packed asm, 1 packed asm + nops, 2 scalar
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f32x2 e = {in[15 * n + i], in[16 * n + i]}, t = {in[17 * n + i], in[18 * n + i]};
    f32x2 acc = {0.f, 0.f};
    for (int c = 0; c < 3; ++c) {
        const float a = in[(5 * c + 0) * n + i], b = in[(5 * c + 1) * n + i], cc = in[(5 * c + 2) * n + i], d = in[(5 * c + 3) * n + i],
                    g = in[(5 * c + 4) * n + i];
        if (MODE == 2) {
            acc.x += g * ((b - a) * e.x + (d - cc) * t.x);
            acc.y += g * ((cc - a) * e.y + (d - b) * t.y);
        } else {
            f32x2 bc = {b, cc}, av = {a, a}, dv = {d, d}, gv = {g, g}, t22, t20;
#define NOP_ if (MODE == 1) asm volatile("s_nop 1");
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t22) : "v"(bc), "v"(av)); NOP_      // (b - a, cc - a)
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t20) : "v"(dv), "v"(bc)); NOP_   // (d - cc, d - b)
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t22) : "v"(e), "v"(t22)); NOP_
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t20) : "v"(t), "v"(t20)); NOP_
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t20) : "v"(t22), "v"(t20)); NOP_
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t20) : "v"(gv), "v"(t20)); NOP_
            asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(acc) : "v"(acc), "v"(t20)); NOP_
#undef NOP_
        }
    }
    out[i] = acc.x;
    out[n + i] = acc.y;
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void trigger(float* __restrict__ sink, int loops) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 16; ++k) acc[i][k] = (float)(tid + i + k) * 1e-3f;
    f16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.01f * (float)((tid + k) & 15)); b[k] = (_Float16)(0.02f * (float)((tid * 3 + k) & 15)); }
    float f = 1.0001f;
    for (int it = 0; it < loops; ++it) {
        if (KIND == 1 || KIND == 3) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        if (KIND == 2 || (KIND == 3 && (it & 3) == 0)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    f32x2 v = {acc[i][k], acc[i][k + 1]};
                    const f32x2 ff = {f, f};
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(ff));
                    acc[i][k] = v.x; acc[i][k + 1] = v.y;
                }
        }
        // 6: MFMA + scalar v_mul_f32 of the accumulators   7: MFMA + v_pk_mul_f32 of registers the MFMAs do not touch
        // 8: wave-specialised: even waves MFMA only, odd waves v_pk_mul_f32 only   9: MFMA + v_pk_add_f32   10: fp32 MFMA (32x32x2) + v_pk_mul_f32
        if (KIND == 6 || KIND == 7 || KIND == 9 || (KIND == 8 && !((tid >> 6) & 1))) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < (KIND == 7 ? 2 : 4); ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        if (KIND == 10) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(f, f, acc[i], 0, 0, 0);
        }
        if ((it & 3) == 0) {
            if (KIND == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(acc[i][k]) : "v"(acc[i][k]), "v"(f));
            }
            if (KIND == 7 || KIND == 9 || KIND == 10 || (KIND == 8 && ((tid >> 6) & 1))) {
#pragma unroll
                for (int i = (KIND == 7 ? 2 : 0); i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 16; k += 2) {
                        f32x2 v = {acc[i][k], acc[i][k + 1]};
                        const f32x2 ff = {f, f};
                        if (KIND == 9) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(ff));
                        else asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(ff));
                        acc[i][k] = v.x; acc[i][k + 1] = v.y;
                    }
            }
        }
        if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(acc[i][k]) : "v"(acc[i][k]), "v"(f));
        }
        if (KIND == 5) {
            lds[(tid * 17 + it) & 4095] = acc[0][0];
            __syncthreads();
            acc[0][0] += lds[(tid * 5 + it) & 4095];
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 16; ++k) s += acc[i][k];
    sink[blockIdx.x * 256 + tid] = s;
}

static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
static float urand(unsigned& s) { return (float)(lcg(s) >> 8) * (1.f / 16777216.f) * 2.f - 1.f; }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    const int n = 65536, NP = 19, NV = 19, REP = 6, NT = 11;
    unsigned seed = 777u;
    std::vector<float> hin((size_t)NP * n);
    for (auto& v : hin) v = urand(seed);
    float *in, *sink, *out[NV][REP];
    CK(hipMalloc(&in, hin.size() * 4)); CK(hipMalloc(&sink, 1024 * 256 * 4));
    CK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
    for (int v = 0; v < NV; ++v)
        for (int r = 0; r < REP; ++r) CK(hipMalloc(&out[v][r], 2 * n * 4));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const char* vname[NV] = {"pk-seq", "pk-seq+nop", "scalar-seq", "pk_mul", "pk_add", "pk_fma", "fma_mix", "pk_add_f16", "pk_mul opsel", "v_mul pair", "pk_add hi<-lo", "pk_add src1 swapped", "pk_add neg", "pk_mul src1 swapped", "pk_add src0 swapped", "fma_mixlo_f16", "fma_mixhi_f16", "pk_mov op_sel", "cvt_pk_f16"};
    const char* tname[NT] = {"none", "mfma", "pk_mul", "mfma+pk_mul", "v_mul", "lds+barrier", "mfma+v_mul", "mfma+pk_mul(indep regs)", "mfma | pk_mul waves",
                             "mfma+pk_add", "mfma_f32+pk_mul"};
    auto launch_v = [&](int v, float* o) {
        const dim3 g(n / 256), b(256);
        switch (v) {
            case 0: hipLaunchKernelGGL(victim<0>, g, b, 0, s1, in, o, n); break;
            case 1: hipLaunchKernelGGL(victim<1>, g, b, 0, s1, in, o, n); break;
            case 2: hipLaunchKernelGGL(victim<2>, g, b, 0, s1, in, o, n); break;
            case 3: hipLaunchKernelGGL(victim1<3>, g, b, 0, s1, in, o, n); break;
            case 4: hipLaunchKernelGGL(victim1<4>, g, b, 0, s1, in, o, n); break;
            case 5: hipLaunchKernelGGL(victim1<5>, g, b, 0, s1, in, o, n); break;
            case 6: hipLaunchKernelGGL(victim1<6>, g, b, 0, s1, in, o, n); break;
            case 7: hipLaunchKernelGGL(victim1<7>, g, b, 0, s1, in, o, n); break;
            case 8: hipLaunchKernelGGL(victim1<8>, g, b, 0, s1, in, o, n); break;
            case 9: hipLaunchKernelGGL(victim1<9>, g, b, 0, s1, in, o, n); break;
            case 10: hipLaunchKernelGGL(victim1<10>, g, b, 0, s1, in, o, n); break;
            case 11: hipLaunchKernelGGL(victim1<11>, g, b, 0, s1, in, o, n); break;
            case 12: hipLaunchKernelGGL(victim1<12>, g, b, 0, s1, in, o, n); break;
            case 13: hipLaunchKernelGGL(victim1<13>, g, b, 0, s1, in, o, n); break;
            case 14: hipLaunchKernelGGL(victim1<14>, g, b, 0, s1, in, o, n); break;
            case 15: hipLaunchKernelGGL(victim1<15>, g, b, 0, s1, in, o, n); break;
            case 16: hipLaunchKernelGGL(victim1<16>, g, b, 0, s1, in, o, n); break;
            case 17: hipLaunchKernelGGL(victim1<17>, g, b, 0, s1, in, o, n); break;
            default: hipLaunchKernelGGL(victim1<18>, g, b, 0, s1, in, o, n); break;
        }
    };
    auto launch_t = [&](int t) {
        const dim3 g(512), b(256);
        const int loops = 4000;
        switch (t) {
            case 1: hipLaunchKernelGGL(trigger<1>, g, b, 0, s2, sink, loops); break;
            case 2: hipLaunchKernelGGL(trigger<2>, g, b, 0, s2, sink, loops); break;
            case 3: hipLaunchKernelGGL(trigger<3>, g, b, 0, s2, sink, loops); break;
            case 4: hipLaunchKernelGGL(trigger<4>, g, b, 0, s2, sink, loops); break;
            case 5: hipLaunchKernelGGL(trigger<5>, g, b, 0, s2, sink, loops); break;
            case 6: hipLaunchKernelGGL(trigger<6>, g, b, 0, s2, sink, loops); break;
            case 7: hipLaunchKernelGGL(trigger<7>, g, b, 0, s2, sink, loops); break;
            case 8: hipLaunchKernelGGL(trigger<8>, g, b, 0, s2, sink, loops); break;
            case 9: hipLaunchKernelGGL(trigger<9>, g, b, 0, s2, sink, loops); break;
            case 10: hipLaunchKernelGGL(trigger<10>, g, b, 0, s2, sink, loops); break;
            default: break;
        }
    };
    std::vector<float> ref[NV], cur(2 * n);
    for (int v = 0; v < NV; ++v) {
        launch_v(v, out[v][0]);
        CK(hipDeviceSynchronize());
        ref[v].resize(2 * n);
        CK(hipMemcpy(ref[v].data(), out[v][0], 2 * n * 4, hipMemcpyDeviceToHost));
    }
    printf("pk-seq vs scalar-seq reference equal: %d   (victims that never differ are listed for trigger none only)\n", memcmp(ref[0].data(), ref[2].data(), 2 * n * 4) == 0);
    for (int t = 0; t < NT; ++t) {
        long bad[NV], quarter[NV][4], half[NV][2];
        memset(bad, 0, sizeof bad); memset(quarter, 0, sizeof quarter); memset(half, 0, sizeof half);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float trig_ms = 0.f;
        for (int it = 0; it < iters; ++it) {
            for (int v = 0; v < NV; ++v)
                for (int r = 0; r < REP; ++r) CK(hipMemsetAsync(out[v][r], 0xFF, 2 * n * 4, s1));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s2));
            launch_t(t);
            CK(hipEventRecord(e1, s2));
            for (int r = 0; r < REP; ++r)
                for (int v = 0; v < NV; ++v) launch_v(v, out[v][r]);
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&trig_ms, e0, e1));
            for (int v = 0; v < NV; ++v)
                for (int r = 0; r < REP; ++r) {
                    CK(hipMemcpy(cur.data(), out[v][r], 2 * n * 4, hipMemcpyDeviceToHost));
                    if (memcmp(cur.data(), ref[v].data(), 2 * n * 4) == 0) continue;
                    ++bad[v];
                    for (int i = 0; i < 2 * n; ++i)
                        if (memcmp(&cur[i], &ref[v][i], 4)) { ++quarter[v][(i % 64) / 16]; ++half[v][i / n]; }
                }
        }
        printf("trigger %-24s (%.3f ms):\n", tname[t], trig_ms);
        for (int v = 0; v < NV; ++v)
            if (bad[v] || t == 0)
                printf("      victim %-12s %3ld/%d launches differ, elements by quarter [%ld %ld %ld %ld], lo / hi half [%ld %ld]\n", vname[v], bad[v], iters * REP,
                       quarter[v][0], quarter[v][1], quarter[v][2], quarter[v][3], half[v][0], half[v][1]);
    }
    return 0;
}
