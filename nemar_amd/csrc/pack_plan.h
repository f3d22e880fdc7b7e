// Weight-pack plans (pack_plan.hip): every convolution family keeps its weights in a packed operand layout that is rebuilt after
// each optimizer step — ~240 launches of 4-6 us per training step when every (weight, direction) packs itself at first use.  A plan
// RECORDS those pack jobs once (the launchers below call nemar_pack_record_job while a plan is recording on the calling thread) and
// then re-runs all of them in a handful of launches: one multi-job kernel per family, grid.z = job, the job arguments read from a
// caller-owned device buffer.  Internal interface between the family files and pack_plan.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

enum {
    PACK_FAM_PRE = 0,           // stage 0: re-arranged weight tensors that later stages read (conv.hip: the 7x7 many -> few layers)
    PACK_FAM_MAX = 1,           // stage 1: NEMAR_PACK_MAX_PARTS partial maxima of |w| per tensor (pack_plan.hip)
    PACK_FAM_EXACT = 2,         // stage 2: the packs (conv.hip, conv_s16g.hip, conv_split16.hip, conv_k7.hip)
    PACK_FAM_S16G = 3,
    PACK_FAM_SPLIT16 = 4,
    PACK_FAM_K7 = 5,
    PACK_FAM_FLIPT = 6,         //          flipped + transposed weights of the narrow (<= 4 channel) data gradients (conv.hip)
    PACK_FAMS = 7
};
constexpr int NEMAR_PACK_MAX_PARTS = 64;
// stage-1 job: out[b] = largest finite |x[i]| (bit pattern) over block b's share of x[0, n), b < NEMAR_PACK_MAX_PARTS — plain stores
struct NemarPackMaxArgs {
    const float* x;
    long long n;
    unsigned* out;
    int gx, gy;
};

// launch the family's multi-job kernel: `jobs` = njobs argument structs in DEVICE memory, grid (gx_max, gy_max, njobs)
typedef void (*nemar_pack_multi_fn)(const void* jobs, int njobs, int gx_max, int gy_max, hipStream_t st);
void nemar_pack_register(int fam, size_t args_bytes, nemar_pack_multi_fn fn);
bool nemar_pack_recording();                                   // a plan is recording on this thread
// args: the family's argument struct (args_bytes of it are copied); gx, gy: the grid the single-job launch would use
void nemar_pack_record_job(int fam, const void* args, int gx, int gy);

// A multi-job kernel: job = blockIdx.z, blocks beyond the job's own grid leave at once
#define NEMAR_PACK_MULTI(KERNEL_, ARGS_, BODY_, THREADS_)                                                                   \
    __global__ __launch_bounds__(THREADS_) void KERNEL_(const ARGS_* jobs) {                                                \
        const ARGS_& a = jobs[blockIdx.z];                                                                                  \
        if ((int)blockIdx.x < a.gx && (int)blockIdx.y < a.gy) BODY_(a, blockIdx.x, blockIdx.y, a.gx);                       \
    }
