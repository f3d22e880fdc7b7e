// Shader-clock trace: one wave samples s_memtime (shader-clock cycles) against s_memrealtime (constant 100 MHz) while other
// kernels run.  sclk over an interval = d(s_memtime) / d(s_memrealtime) * 100 MHz.  Loaded by tools/clock_trace.py (ctypes).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/clock_probe.hip -o tools/probes/_build/libclock_probe.so
#include <hip/hip_runtime.h>

__global__ void clock_probe_kernel(long long* out, int samples, int spin) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < samples; ++i) {
        out[2 * i] = clock64();
        out[2 * i + 1] = wall_clock64();
        for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(64);
    }
}

extern "C" __attribute__((visibility("default"))) int clock_probe_launch(long long* out, int samples, int spin, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, samples, spin);
    return (int)hipGetLastError();
}
