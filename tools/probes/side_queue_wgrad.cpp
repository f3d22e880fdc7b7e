// Is the wide-layer weight gradient (wgrad_split16_kernel + its split pass and slab sum) bitwise repeatable while ANOTHER stream keeps
// the chip busy?  (Round 5: the all-configuration side-stream test saw one 32 x 32 x (one tap row) tile of a residual-block weight
// gradient differ in ~1 of 200 steps at batch 2 — config c3_full — and never on one stream.)  Plain HIP, no torch, fixed buffers.
//
//   hipcc -O2 -o side_queue_wgrad side_queue_wgrad.cpp -ldl
//   ./side_queue_wgrad <libnemar_hip.so> [iterations] [N] [H] [trigger: 0 none, 1 wide forward calls, 2 device-to-device copies] [victims per iteration]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Extras {
    void* scratch; size_t scratch_bytes; const void* src_max_words; int src_max_count; const void* src2_max_words; int src2_max_count;
    const void* src_planes; void* gy_planes_out; size_t gy_planes_bytes; const void* src2_planes;
};
typedef size_t (*wws_fn)(int, int, int, int, int, int, int, int, int, int, int);
typedef size_t (*fws_fn)(int, int, int, int, int, int, int, int, int);
typedef size_t (*scr_fn)(int, int, int, int, int, int, int, int, int);
typedef int (*wgrad_fn)(const float*, int, const float*, int, const float*, float*, float*, int, int, int, int, int, int, int, int, int, int,
                        int, void*, size_t, void*, const Extras*);
typedef int (*fwd_fn)(const float*, int, const float*, int, const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int,
                      float, void*, size_t, int, void*, const Extras*);
typedef int (*amax_fn)(const float*, int, long long, void*, void*);
typedef size_t (*dws_fn)(int, int, int, int, int, int, int, int, int, int);
typedef size_t (*gpb_fn)(int, int, int, int, int, int, int, int, int, int);
typedef int (*dgrad_fn)(const float*, const float*, const float*, int, float, float*, int, float*, int, int, int, int, int, int, int, int, int, int, int,
                        int, void*, size_t, int, void*, const Extras*);
typedef int (*route_fn)(void);

static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
static float urand(unsigned& s) { return (float)(lcg(s) >> 8) * (1.f / 16777216.f) * 2.f - 1.f; }

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: %s lib.so [iters] [N] [H] [trigger] [victims]\n", argv[0]); return 2; }
    const int iters = argc > 2 ? atoi(argv[2]) : 100, N = argc > 3 ? atoi(argv[3]) : 4, H = argc > 4 ? atoi(argv[4]) : 64;
    const int trigger = argc > 5 ? atoi(argv[5]) : 1, nvict = argc > 6 ? atoi(argv[6]) : 6;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
    wws_fn wws = (wws_fn)dlsym(h, "nemar_conv2d_bwd_weight_workspace");
    fws_fn fws = (fws_fn)dlsym(h, "nemar_conv2d_fwd_workspace");
    scr_fn scr = (scr_fn)dlsym(h, "nemar_conv2d_scratch");
    wgrad_fn wgrad = (wgrad_fn)dlsym(h, "nemar_conv2d_bwd_weight_ex");
    fwd_fn fwd = (fwd_fn)dlsym(h, "nemar_conv2d_fwd_ex");
    amax_fn amax = (amax_fn)dlsym(h, "nemar_absmax_samples");
    route_fn route = (route_fn)dlsym(h, "nemar_last_route");
    dws_fn dws = (dws_fn)dlsym(h, "nemar_conv2d_bwd_data_workspace");
    gpb_fn gpb = (gpb_fn)dlsym(h, "nemar_conv2d_gy_planes_bytes");
    dgrad_fn dgrad = (dgrad_fn)dlsym(h, "nemar_conv2d_bwd_data_ex");
    if (!wws || !fws || !scr || !wgrad || !fwd || !amax || !route || !dws || !gpb || !dgrad) { printf("missing symbol\n"); return 2; }
    const int W = H, C = 256, K = 256;
    const size_t px = (size_t)H * W, nx = (size_t)N * C * px, nw = (size_t)K * C * 9;
    unsigned seed = 4242u;
    std::vector<float> hx(nx), hg(nx), hw(nw);
    for (auto& v : hx) v = urand(seed);
    for (auto& v : hg) v = 0.01f * urand(seed);
    for (auto& v : hw) v = 0.02f * urand(seed);
    float *x, *g, *w, *y, *xt, *yt;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&g, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, nx * 4));
    const int Nt = 16;                                           // the trigger: a batch-16 forward call of the same layer
    CK(hipMalloc(&xt, (size_t)Nt * C * px * 4)); CK(hipMalloc(&yt, (size_t)Nt * K * px * 4));
    CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hg.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < Nt; i += N) CK(hipMemcpy(xt + (size_t)i * C * px, hx.data(), (size_t)(Nt - i < N ? Nt - i : N) * C * px * 4, hipMemcpyHostToDevice));
    hipStream_t s_main, s_side;
    CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));
    // side inputs: arenas per stream, max words of x and gy
    const size_t sb = scr(N, H, W, K, C, 3, 3, 1, 1), sbt = scr(Nt, H, W, K, C, 3, 3, 1, 1);
    void *arena, *arena_t;
    unsigned *wx, *wg, *wxt;
    CK(hipMalloc(&arena, sb + 256)); CK(hipMalloc(&arena_t, sbt + 256));
    CK(hipMalloc(&wx, 4 * N)); CK(hipMalloc(&wg, 4 * N)); CK(hipMalloc(&wxt, 4 * Nt));
    CK(hipMemset(wx, 0, 4 * N)); CK(hipMemset(wg, 0, 4 * N)); CK(hipMemset(wxt, 0, 4 * Nt));
    amax(x, N, (long long)C * px, wx, s_side); amax(g, N, (long long)K * px, wg, s_side); amax(xt, Nt, (long long)C * px, wxt, s_side);
    CK(hipDeviceSynchronize());
    const size_t wsb = wws(N, C, H, W, K, H, W, 3, 3, 1, 1), fsb = fws(Nt, H, W, K, C, 3, 3, 1, 1);
    void *ws, *wst;
    CK(hipMalloc(&ws, wsb + 256)); CK(hipMalloc(&wst, fsb + 256));
    std::vector<float*> gw(nvict);
    for (auto& p : gw) CK(hipMalloc(&p, nw * 4));
    // triggers 3 / 4: the step's own partner — the data-gradient call of the same layer shape on the other stream (its gy split writes the planes
    // it hands to the weight gradient: 3; plain split: 4); victims then take their gy planes from such a call (hand-over), as in the step
    const size_t dwb = dws(N, C, H, W, K, 3, 3, 1, 1, 1), gbytes = gpb(N, C, H, W, K, 3, 3, 1, 1, 1);
    void *dwsb, *gplanes, *gplanes_t, *arena_d;
    float* gx;
    CK(hipMalloc(&dwsb, dwb + 256)); CK(hipMalloc(&gplanes, gbytes + 256)); CK(hipMalloc(&gplanes_t, gbytes + 256)); CK(hipMalloc(&arena_d, sb + 256));
    CK(hipMalloc(&gx, nx * 4));
    Extras ed = {arena_d, sb, wg, N, nullptr, 0, nullptr, gplanes, gbytes, nullptr};
    const bool handover = trigger >= 3 && gbytes > 0;
    if (handover) {
        if (dgrad(g, w, nullptr, 0, 0.f, gx, C, nullptr, 0, N, H, W, K, H, W, 3, 3, 1, 1, 1, dwsb, dwb, 0, s_side, &ed)) { printf("bwd_data failed\n"); return 2; }
        CK(hipDeviceSynchronize());
        printf("data-gradient route %d, gy planes %zu bytes handed to the victims\n", route(), gbytes);
    }
    Extras edt = {arena_d, sb, wg, N, nullptr, 0, nullptr, trigger == 3 ? gplanes_t : nullptr, trigger == 3 ? gbytes : 0, nullptr};
    Extras ev = {arena, sb, wx, N, wg, N, nullptr, nullptr, 0, handover ? gplanes : nullptr};
    Extras et = {arena_t, sbt, wxt, Nt, nullptr, 0, nullptr, nullptr, 0, nullptr};
    auto victim = [&](float* out) {
        CK(hipMemsetAsync(out, 0, nw * 4, s_side));
        if (wgrad(x, C, nullptr, 0, g, out, nullptr, N, H, W, K, H, W, 3, 3, 1, 1, 1, ws, wsb, s_side, &ev)) { printf("bwd_weight failed\n"); exit(2); }
    };
    // packed weights of the trigger once
    if (fwd(xt, C, nullptr, 0, w, nullptr, yt, Nt, H, W, K, 3, 3, 1, 1, 1, 0, 0.f, wst, fsb, 0, s_main, &et)) { printf("fwd failed\n"); return 2; }
    printf("routes: trigger forward %d, ", route());
    std::vector<float> ref(nw), cur(nw);
    victim(gw[0]);
    printf("victim weight gradient %d (2 = the wide fp16 x 3 route)\n", route());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), gw[0], nw * 4, hipMemcpyDeviceToHost));
    long bad = 0, elems = 0;
    for (int it = 0; it < iters; ++it) {
        if (trigger == 1)
            for (int r = 0; r < 3; ++r) fwd(xt, C, nullptr, 0, w, nullptr, yt, Nt, H, W, K, 3, 3, 1, 1, 1, 0, 0.f, wst, fsb, 1, s_main, &et);
        if (trigger == 2)
            for (int r = 0; r < 12; ++r) CK(hipMemcpyAsync(yt, xt, (size_t)Nt * C * px * 4, hipMemcpyDeviceToDevice, s_main));
        if (trigger >= 3)
            for (int r = 0; r < 2 * nvict; ++r) dgrad(g, w, nullptr, 0, 0.f, gx, C, nullptr, 0, N, H, W, K, H, W, 3, 3, 1, 1, 1, dwsb, dwb, 1, s_main, &edt);
        for (int v = 0; v < nvict; ++v) victim(gw[v]);
        CK(hipDeviceSynchronize());
        for (int v = 0; v < nvict; ++v) {
            CK(hipMemcpy(cur.data(), gw[v], nw * 4, hipMemcpyDeviceToHost));
            if (memcmp(cur.data(), ref.data(), nw * 4) == 0) continue;
            ++bad;
            long n = 0;
            int kmin = 1 << 30, kmax = -1, cmin = 1 << 30, cmax = -1;
            unsigned taps = 0;
            for (size_t i = 0; i < nw; ++i)
                if (memcmp(&cur[i], &ref[i], 4)) {
                    ++n;
                    const int k = (int)(i / (C * 9)), c = (int)(i / 9 % C), t = (int)(i % 9);
                    kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
                    taps |= 1u << t;
                }
            elems += n;
            if (bad <= 6) printf("   iteration %d victim %d: %ld elements, k %d..%d, c %d..%d, tap mask 0x%03x\n", it, v, n, kmin, kmax, cmin, cmax, taps);
        }
    }
    printf("N %d, %dx%d, trigger %d: %ld of %ld weight-gradient calls differ from the one run alone (%ld elements)\n", N, H, W, trigger, bad,
           (long)iters * nvict, elems);
    return 0;
}
