// Library-level entry points of the C-ABI (include/nemar_hip.h): version and last-error string.
#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <unordered_map>

#ifdef NEMAR_AB
int g_lds_claim = NEMAR_LDS_CLAIM_DEFAULT;
#endif

size_t nemar_lds_bytes(const void* kernel, size_t need, bool claim) {
#ifdef NEMAR_HOST_EMULATION
    (void)kernel; (void)claim;
    return need;                                       // (the emulator's kernels hold their LDS image as static arrays)
#else
    static std::mutex mu;
    // (device, kernel) -> dynamic bytes that fill the CU beside the kernel's static LDS (the attribute is per device)
    static std::unordered_map<unsigned long long, size_t> fill_of;
    size_t fill;
    {
        std::lock_guard<std::mutex> lock(mu);
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long key = (unsigned long long)(uintptr_t)kernel ^ ((unsigned long long)(dev + 1) << 56);
        auto it = fill_of.find(key);
        if (it == fill_of.end()) {
            hipFuncAttributes fa;
            size_t stat = 0;
            if (hipFuncGetAttributes(&fa, kernel) == hipSuccess) stat = fa.sharedSizeBytes;
            int cu = 0;
            // gfx950 has 160 KiB of LDS per CU; a runtime that reports less (64 KiB: the per-workgroup default of older ROCm) must not
            // put the attribute below what the kernels ask for
            if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || cu < 160 * 1024) cu = 160 * 1024;
            fill = (size_t)cu > stat ? (size_t)cu - stat : 0;
            if (fill < need) fill = need;
            const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fill);
            if (e != hipSuccess) nemar_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu): %s", fill, hipGetErrorString(e));
            (void)hipGetLastError();
            it = fill_of.emplace(key, fill).first;
        } else if (it->second < need) {
            const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
            if (e != hipSuccess) nemar_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu): %s", need, hipGetErrorString(e));
            (void)hipGetLastError();
            it->second = need;
        }
        fill = it->second;
    }
    return (claim && fill > need) ? fill : need;
#endif
}

#define NEMAR_HIP_VERSION 602  // major*10000 + minor*100 + patch  (0.6.2: nemar_concat_pieces, nemar_add2; 0.6.1: nemar_set_max_words_lazy; 0.6.0: round 6 — producer-written operand planes for all three calls of the
                               // wide layers, fused skip-gradient add / max words in the data gradient's epilogue; 0.5.0: round 5 — no nemar_tune* in the product library: the measurement
                               // switches are constants there and live in libnemar_hip_ab.so (-DNEMAR_AB, include/nemar_hip_ab.h);
                               // 0.4.0: side inputs per call only, weight-pack plans, nemar_store_words, 7x7 layers on the 16-bit pipe)

static thread_local char g_err[512] = "";
static thread_local int g_max_words_lazy = 0;
bool nemar_max_words_lazy() { return g_max_words_lazy != 0; }
// While on (per calling thread), producers that publish per-sample maxima (nemar_instnorm_fwd_planes max_words, nemar_conv_extras.out_max_words)
// leave the reduction of their partial words to the consumer: ONLY for words that go to nemar_instnorm_fwd_planes (residual_max_words) /
// nemar_instnorm_bwd_planes (gy_max_words) — no other entry point understands the marker (csrc/max_words.h).  Returns the previous setting.
NEMAR_API int nemar_set_max_words_lazy(int on) {
    const int prev = g_max_words_lazy;
    g_max_words_lazy = on ? 1 : 0;
    return prev;
}

void nemar_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

NEMAR_API int nemar_version(void) { return NEMAR_HIP_VERSION; }

// Message of the most recent failing call on the calling thread ("" if none).
NEMAR_API const char* nemar_last_error(void) { return g_err; }
