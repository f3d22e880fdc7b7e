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
    static std::unordered_map<const void*, size_t> fill_of;      // kernel -> dynamic bytes that fill the CU beside its static LDS
    size_t fill;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = fill_of.find(kernel);
        if (it == fill_of.end()) {
            hipFuncAttributes fa;
            size_t stat = 0;
            if (hipFuncGetAttributes(&fa, kernel) == hipSuccess) stat = fa.sharedSizeBytes;
            int dev = 0, cu = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || cu <= 0) cu = 160 * 1024;
            fill = (size_t)cu > stat ? (size_t)cu - stat : 0;
            (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fill);
            (void)hipGetLastError();
            it = fill_of.emplace(kernel, fill).first;
        }
        fill = it->second;
    }
    return (claim && fill > need) ? fill : need;
#endif
}

#define NEMAR_HIP_VERSION 500  // major*10000 + minor*100 + patch  (0.5.0: round 5 — no nemar_tune* in the product library: the measurement
                               // switches are constants there and live in libnemar_hip_ab.so (-DNEMAR_AB, include/nemar_hip_ab.h);
                               // 0.4.0: side inputs per call only, weight-pack plans, nemar_store_words, 7x7 layers on the 16-bit pipe)

static thread_local char g_err[512] = "";

void nemar_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

NEMAR_API int nemar_version(void) { return NEMAR_HIP_VERSION; }

// Message of the most recent failing call on the calling thread ("" if none).
NEMAR_API const char* nemar_last_error(void) { return g_err; }
