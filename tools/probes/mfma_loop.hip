// Probe: what limits a "stage" loop of fp32 MFMAs?  Variants of {accumulators per wave, operand source, barrier}.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e)); exit(1);} } while (0)

// NACC accumulators, 16 MFMAs per stage; LDSOP: operands re-read from LDS each stage; BAR: barrier each stage
template <int NACC, bool LDSOP, bool BAR, int NT>
__global__ __launch_bounds__(NT) void k(const float* in, float* out, int stages) {
    __shared__ float lds[16 * 128 * 2];
    for (int i = threadIdx.x; i < 16 * 128 * 2; i += NT) lds[i] = in[i & 511];
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
    f32x16 c[NACC];
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
    float a[8][2], b[8];
    for (int k2 = 0; k2 < 8; ++k2) { a[k2][0] = in[lane]; a[k2][1] = in[64 + lane]; b[k2] = in[128 + lane]; }
    for (int s = 0; s < stages; ++s) {
        if (LDSOP) {
            const int buf = s & 1;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const int kr = 2 * k2 + lhi;
                a[k2][0] = lds[buf * 2048 + kr * 128 + l31];
                a[k2][1] = lds[buf * 2048 + kr * 128 + 32 + l31];
                b[k2] = lds[buf * 2048 + kr * 128 + 64 + l31];
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            c[(2 * k2) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2][0], b[k2], c[(2 * k2) % NACC], 0, 0, 0);
            c[(2 * k2 + 1) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2][1], b[k2], c[(2 * k2 + 1) % NACC], 0, 0, 0);
        }
        if (BAR) __syncthreads();
    }
    float t = 0;
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) t += c[q][r];
    out[blockIdx.x * NT + threadIdx.x] = t;
}

template <int NACC, bool LDSOP, bool BAR, int NT>
void run(const char* name, const float* in, float* out, int blocks) {
    const int stages = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NACC, LDSOP, BAR, NT>), dim3(blocks), dim3(NT), 0, 0, in, out, 100);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NACC, LDSOP, BAR, NT>), dim3(blocks), dim3(NT), 0, 0, in, out, stages);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double fl = (double)blocks * (NT / 64) * stages * 16 * 32 * 32 * 2 * 2;
    printf("%-44s blocks=%4d waves/SIMD=%d : %6.1f TFLOP/s\n", name, blocks, blocks * (NT / 64) / 1024, fl / ms / 1e9);
}

int main() {
    float *in, *out; float h[512];
    CK(hipMalloc(&in, 2048)); CK(hipMalloc(&out, 4096 * 512 * 4));
    for (int i = 0; i < 512; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
    CK(hipMemcpy(in, h, 2048, hipMemcpyHostToDevice));
    for (int blocks : {256, 512}) {
        run<4, false, false, 256>("4acc regs nobar NT256", in, out, blocks * 2);
        run<2, false, false, 256>("2acc regs nobar NT256", in, out, blocks * 2);
        run<2, false, false, 512>("2acc regs nobar NT512", in, out, blocks);
        run<2, true, false, 512>("2acc LDS  nobar NT512", in, out, blocks);
        run<2, true, true, 512>("2acc LDS  bar   NT512", in, out, blocks);
        run<4, true, true, 256>("4acc LDS  bar   NT256", in, out, blocks * 2);
        run<2, false, true, 512>("2acc regs bar   NT512", in, out, blocks);
    }
    return 0;
}
