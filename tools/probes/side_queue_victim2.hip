// Which quantity of the victim goes wrong?  (follow-up of side_queue_victim.cpp.)  The victim here is a COPY of the
// arithmetic of grid_sample_bwd_kernel<UNET, false> that also stores its intermediates; the trigger is still the library's 7x7 stem
// weight-gradient call on a second stream.  A set of smaller victims (copy, integer division, floor / fract chain) runs beside it.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o side_queue_victim2 side_queue_victim2.hip -ldl
//   ./side_queue_victim2 <libnemar_hip.so> [iterations] [N of the stem call]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef size_t (*ws_fn)(int, int, int, int, int, int, int, int, int, int, int);
typedef int (*wgrad_fn)(const float*, int, const float*, int, const float*, float*, float*, int, int, int, int, int, int, int, int, int, int,
                        int, void*, size_t, void*);

__device__ __forceinline__ float linspace_m1_p1(int i, int n) {
    if (n <= 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return (i < n / 2) ? fmaf(step, (float)i, -1.f) : fmaf(-step, (float)(n - 1 - i), 1.f);
}

constexpr int NQ = 12;      // dumped quantities per pixel
// q: 0 h  1 w  2 gx  3 gy  4 ix  5 iy  6 x0  7 y0  8 tx  9 ty  10 ggx  11 ggy
__global__ __launch_bounds__(256) void victim_full(const float* __restrict__ in, const float* __restrict__ gsrc, const float* __restrict__ gout,
                                                   float* __restrict__ dump, int C, int H, int W) {
    const int items = H * W;
    const size_t plane = (size_t)H * W;
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int h = it / W;
        const int w = it - h * W;
        const float gx = linspace_m1_p1(w, W) + gsrc[it];
        const float gy = linspace_m1_p1(h, H) + gsrc[it + plane];
        const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
        const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0i = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f), y0i = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
        const float tx = ix - fx, ty = iy - fy;
        const float ex = 1.f - tx, ey = 1.f - ty;
        const bool x0 = (unsigned)x0i < (unsigned)W, x1 = (unsigned)(x0i + 1) < (unsigned)W;
        const bool y0 = (unsigned)y0i < (unsigned)H, y1 = (unsigned)(y0i + 1) < (unsigned)H;
        const int o = y0i * W + x0i;
        float gix = 0.f, giy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float g = gout[(size_t)c * plane + it];
            const float* p = in + (size_t)c * plane;
            const float a = (x0 && y0) ? p[o] : 0.f;
            const float b = (x1 && y0) ? p[o + 1] : 0.f;
            const float cc = (x0 && y1) ? p[o + W] : 0.f;
            const float d = (x1 && y1) ? p[o + W + 1] : 0.f;
            gix += g * ((b - a) * ey + (d - cc) * ty);
            giy += g * ((cc - a) * ex + (d - b) * tx);
        }
        float* q = dump + it;
        q[0 * plane] = (float)h; q[1 * plane] = (float)w; q[2 * plane] = gx; q[3 * plane] = gy; q[4 * plane] = ix; q[5 * plane] = iy;
        q[6 * plane] = (float)x0i; q[7 * plane] = (float)y0i; q[8 * plane] = tx; q[9 * plane] = ty;
        q[10 * plane] = gix * (0.5f * (float)W); q[11 * plane] = giy * (0.5f * (float)H);
    }
}

// small victims, one output plane each
__global__ __launch_bounds__(256) void victim_copy(const float* __restrict__ a, float* __restrict__ o, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = a[i];
}
__global__ __launch_bounds__(256) void victim_div(float* __restrict__ o, int n, int W) {      // run-time integer division + float division
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const int h = i / W; o[i] = (float)h * (2.f / (float)(W - 1)); }
}
__global__ __launch_bounds__(256) void victim_floor(const float* __restrict__ a, float* __restrict__ o, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float v = a[i] * 100.f; o[i] = v - floorf(v); }
}
__global__ __launch_bounds__(256) void victim_fma(const float* __restrict__ a, float* __restrict__ o, int n) {       // a longer dependent VALU chain
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        float v = a[i];
#pragma unroll
        for (int k = 0; k < 32; ++k) v = fmaf(v, 0.999f, 0.001f * (float)k);
        o[i] = v;
    }
}

static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
static float urand(unsigned& s) { return (float)(lcg(s) >> 8) * (1.f / 16777216.f) * 2.f - 1.f; }

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: %s lib.so [iters] [Nstem]\n", argv[0]); return 2; }
    const int iters = argc > 2 ? atoi(argv[2]) : 100, Ns = argc > 3 ? atoi(argv[3]) : 16;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
    ws_fn wsb = (ws_fn)dlsym(h, "nemar_conv2d_bwd_weight_workspace");
    wgrad_fn wgrad = (wgrad_fn)dlsym(h, "nemar_conv2d_bwd_weight");
    const int H = 256, W = 256, K = 64, C = 3;
    const size_t px = (size_t)H * W;
    unsigned seed = 12345u;
    std::vector<float> hx((size_t)Ns * C * px), hg((size_t)Ns * K * px), himg(3 * px), hgs(2 * px), hgo(3 * px);
    for (auto& v : hx) v = urand(seed);
    for (auto& v : hg) v = 0.01f * urand(seed);
    for (auto& v : himg) v = urand(seed);
    for (auto& v : hgo) v = 0.01f * urand(seed);
    for (size_t i = 0; i < px; ++i) {
        const float y = (float)(i / W), x = (float)(i % W);
        hgs[i] = 0.01f * sinf(x * 0.05f) * cosf(y * 0.03f);
        hgs[px + i] = 0.01f * cosf(x * 0.04f + 1.f) * sinf(y * 0.06f);
    }
    float *x, *g, *gw, *gb, *img, *gs, *go, *ws;
    const size_t wbytes = wsb(Ns, C, H, W, K, H, W, 7, 7, 1, 3);
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&g, hg.size() * 4)); CK(hipMalloc(&gw, (size_t)K * C * 49 * 4)); CK(hipMalloc(&gb, K * 4));
    CK(hipMalloc(&img, himg.size() * 4)); CK(hipMalloc(&gs, hgs.size() * 4)); CK(hipMalloc(&go, hgo.size() * 4)); CK(hipMalloc(&ws, wbytes + 256));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(img, himg.data(), himg.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(gs, hgs.data(), hgs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(go, hgo.data(), hgo.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(gw, 0, (size_t)K * C * 49 * 4)); CK(hipMemset(gb, 0, K * 4));
    constexpr int NV = 5;                                       // victims: full (NQ planes), copy, div, floor, fma (1 plane each)
    const char* vname[NV] = {"full", "copy", "div", "floor", "fma"};
    const size_t vplanes[NV] = {NQ, 1, 1, 1, 1};
    constexpr int REP = 6;                                      // launches of every victim per iteration
    float* out[NV][REP];
    std::vector<float> ref[NV];
    for (int v = 0; v < NV; ++v)
        for (int r = 0; r < REP; ++r) CK(hipMalloc(&out[v][r], vplanes[v] * px * 4));
    hipStream_t s_main, s_side;
    CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));
    const dim3 grid(256), block(256);
    auto launch = [&](int v, float* o) {
        switch (v) {
            case 0: hipLaunchKernelGGL(victim_full, grid, block, 0, s_main, img, gs, go, o, C, H, W); break;
            case 1: hipLaunchKernelGGL(victim_copy, grid, block, 0, s_main, img, o, (int)px); break;
            case 2: hipLaunchKernelGGL(victim_div, grid, block, 0, s_main, o, (int)px, W); break;
            case 3: hipLaunchKernelGGL(victim_floor, grid, block, 0, s_main, img, o, (int)px); break;
            case 4: hipLaunchKernelGGL(victim_fma, grid, block, 0, s_main, img, o, (int)px); break;
        }
    };
    for (int v = 0; v < NV; ++v) {
        launch(v, out[v][0]);
        CK(hipStreamSynchronize(s_main));
        ref[v].resize(vplanes[v] * px);
        CK(hipMemcpy(ref[v].data(), out[v][0], vplanes[v] * px * 4, hipMemcpyDeviceToHost));
    }
    long bad[NV] = {0, 0, 0, 0, 0}, badq[NQ], quarter[NV][4];
    memset(badq, 0, sizeof badq); memset(quarter, 0, sizeof quarter);
    std::vector<float> cur(NQ * px);
    int shown = 0;
    for (int it = 0; it < iters; ++it) {
        for (int v = 0; v < NV; ++v)
            for (int r = 0; r < REP; ++r) CK(hipMemsetAsync(out[v][r], 0xFF, vplanes[v] * px * 4, s_main));
        CK(hipDeviceSynchronize());
        if (wgrad(x, C, nullptr, 0, g, gw, gb, Ns, H, W, K, H, W, 7, 7, 1, 3, 1, ws, wbytes, s_side)) { printf("bwd_weight failed\n"); return 2; }
        for (int r = 0; r < REP; ++r)
            for (int v = 0; v < NV; ++v) launch(v, out[v][r]);
        CK(hipDeviceSynchronize());
        for (int v = 0; v < NV; ++v)
            for (int r = 0; r < REP; ++r) {
                CK(hipMemcpy(cur.data(), out[v][r], vplanes[v] * px * 4, hipMemcpyDeviceToHost));
                if (memcmp(cur.data(), ref[v].data(), vplanes[v] * px * 4) == 0) continue;
                ++bad[v];
                for (size_t i = 0; i < vplanes[v] * px; ++i)
                    if (memcmp(&cur[i], &ref[v][i], 4)) {
                        ++quarter[v][(i % 64) / 16];
                        if (v == 0) ++badq[i / px];
                        if (v == 0 && shown < 24 && (i / px == 3 || i / px == 5 || i / px == 9)) {
                            ++shown;
                            const size_t p = i % px;
                            printf("   pixel (y %zu, x %zu) quantity %zu: %.9g vs %.9g   [h %g w %g gx %.9g|%.9g gy %.9g|%.9g iy %.9g|%.9g ty %.9g|%.9g]\n", p / W, p % W, i / px,
                                   cur[i], ref[0][i], cur[p], cur[px + p], cur[2 * px + p], ref[0][2 * px + p], cur[3 * px + p], ref[0][3 * px + p],
                                   cur[5 * px + p], ref[0][5 * px + p], cur[9 * px + p], ref[0][9 * px + p]);
                        }
                    }
            }
    }
    for (int v = 0; v < NV; ++v)
        printf("victim %-5s: %ld of %d launches differ; differing elements by 16-lane quarter [%ld %ld %ld %ld]\n", vname[v], bad[v], iters * REP,
               quarter[v][0], quarter[v][1], quarter[v][2], quarter[v][3]);
    printf("victim full, differing elements by quantity (h w gx gy ix iy x0 y0 tx ty ggx ggy):");
    for (int q = 0; q < NQ; ++q) printf(" %ld", badq[q]);
    printf("\n");
    return 0;
}
