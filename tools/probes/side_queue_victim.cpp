// Stand-alone (torch-free) repro attempt of the round-4 "lost store" anomaly: while the 7x7 stem weight-gradient call
// (k7_wgrad_kernel, conv_k7.hip) runs on one HIP stream, grid_sample's grid-gradient kernel (grid_sample_bwd_kernel<UNET, false>,
// warp.hip) runs on another and its results are compared, bit for bit, with the ones it produced alone.
//
//   hipcc -O2 -o side_queue_victim side_queue_victim.cpp -ldl        (host code only: the kernels come from the library)
//   ./side_queue_victim <libnemar_hip.so> [iterations] [victims per iteration] [N of the stem call] [trigger: 1 stem wgrad, 0 none]
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
typedef int (*gsb_fn)(const float*, const float*, int, const float*, float*, int, float*, int, int, int, int, int, int, int, void*, size_t,
                      void*);

static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
static float urand(unsigned& s) { return (float)(lcg(s) >> 8) * (1.f / 16777216.f) * 2.f - 1.f; }

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: %s lib.so [iters] [victims] [Nstem] [trigger]\n", argv[0]); return 2; }
    const int iters = argc > 2 ? atoi(argv[2]) : 200, nvict = argc > 3 ? atoi(argv[3]) : 16, Ns = argc > 4 ? atoi(argv[4]) : 2;
    const int trigger = argc > 5 ? atoi(argv[5]) : 1;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
    ws_fn wsb = (ws_fn)dlsym(h, "nemar_conv2d_bwd_weight_workspace");
    wgrad_fn wgrad = (wgrad_fn)dlsym(h, "nemar_conv2d_bwd_weight");
    gsb_fn gsb = (gsb_fn)dlsym(h, "nemar_grid_sample_bwd");
    if (!wsb || !wgrad || !gsb) { printf("missing symbol\n"); return 2; }
    const int H = 256, W = 256, K = 64, C = 3;
    const size_t px = (size_t)H * W;
    unsigned seed = 12345u;
    // the stem's operands
    std::vector<float> hx((size_t)Ns * C * px), hg((size_t)Ns * K * px);
    for (auto& v : hx) v = urand(seed);
    for (auto& v : hg) v = 0.01f * urand(seed);
    // the victim's operands: image, smooth small offsets, upstream gradient
    std::vector<float> himg(3 * px), hgs(2 * px), hgo(3 * px);
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
    std::vector<float*> out(nvict);
    for (auto& p : out) CK(hipMalloc(&p, 2 * px * 4));
    hipStream_t s_main, s_side;
    CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));
    // reference: the victim alone
    std::vector<float> ref(2 * px), cur(2 * px);
    CK(hipMemset(out[0], 0xFF, 2 * px * 4));
    if (gsb(img, gs, 1, go, nullptr, 0, out[0], 0, 1, 3, H, W, H, W, nullptr, 0, s_main)) { printf("grid_sample_bwd failed\n"); return 2; }
    CK(hipStreamSynchronize(s_main));
    CK(hipMemcpy(ref.data(), out[0], 2 * px * 4, hipMemcpyDeviceToHost));
    long bad_launches = 0, bad_elems = 0, nan_elems = 0, quarter[4] = {0, 0, 0, 0}, plane[2] = {0, 0};
    for (int it = 0; it < iters; ++it) {
        for (auto& p : out) CK(hipMemsetAsync(p, 0xFF, 2 * px * 4, s_main));      // NaN sentinel, long before the victims
        CK(hipDeviceSynchronize());
        if (trigger)
            if (wgrad(x, C, nullptr, 0, g, gw, gb, Ns, H, W, K, H, W, 7, 7, 1, 3, 1, ws, wbytes, s_side)) { printf("bwd_weight failed\n"); return 2; }
        for (int v = 0; v < nvict; ++v)
            if (gsb(img, gs, 1, go, nullptr, 0, out[v], 0, 1, 3, H, W, H, W, nullptr, 0, s_main)) { printf("grid_sample_bwd failed\n"); return 2; }
        CK(hipDeviceSynchronize());
        for (int v = 0; v < nvict; ++v) {
            CK(hipMemcpy(cur.data(), out[v], 2 * px * 4, hipMemcpyDeviceToHost));
            if (memcmp(cur.data(), ref.data(), 2 * px * 4) == 0) continue;
            ++bad_launches;
            for (size_t i = 0; i < 2 * px; ++i)
                if (memcmp(&cur[i], &ref[i], 4)) {
                    ++bad_elems; nan_elems += std::isnan(cur[i]); ++quarter[(i % 64) / 16]; ++plane[i / px];
                    if (bad_elems <= 4) printf("   iteration %d victim %d element %zu (plane %zu, y %zu, x %zu): %g vs %g\n", it, v, i, i / px, (i % px) / W, i % W, cur[i], ref[i]);
                }
        }
    }
    printf("trigger %d, stem batch %d: %ld of %ld victim launches differ; %ld elements (%ld NaN); by 16-lane quarter [%ld %ld %ld %ld], by plane [%ld %ld]\n",
           trigger, Ns, bad_launches, (long)iters * nvict, bad_elems, nan_elems, quarter[0], quarter[1], quarter[2], quarter[3], plane[0], plane[1]);
    return 0;
}
