// Library-level entry points of the C-ABI (include/nemar_hip.h): version and last-error string.
#include "common.h"
#include <stdarg.h>

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
