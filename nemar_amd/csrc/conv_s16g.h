// Internal interface of conv_s16g.hip / conv_s16g_wgrad.hip: the GENERAL convolutions (any tap table, source stride 1 or 2, one or two
// sources / destinations, zero or reflect border, 16 .. 256 output channels) on the 16-bit matrix pipe at fp32 accuracy, with the
// fp32 -> 2 x fp16 operand split done INSIDE the kernel on the way into LDS (no split / max passes, no scratch arena).
// Called by conv.hip's operators only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

constexpr int S16G_MAX_TAPS = 64, S16G_MAX_CLS = 4, S16G_CLS_TAPS = 16;

// One launch = up to four tap classes (the output-parity classes of a stride-2 data gradient / transposed convolution; one class
// otherwise).  Class c: ntaps[c] taps (dy, dx) relative to output pixel * source stride, output pixel (oy, ox) of the class lands
// at ((oy * osy + ooy[c]) * OWf + ox * osx + oox[c]) of the destination plane.
struct S16gProblem {
    const float* src0; const float* src1; int C0, C1;      // sources [N, C0, Hs, Ws] (+ [N, C1, Hs, Ws]): reduction channels C0 + C1
    int Hs, Ws;
    int N, M, M0;                                          // output rows (channels); rows < M0 go to dst0, the rest to dst1
    float* dst0; float* dst1;
    const float* bias;                                     // [M] or null
    int act; float slope;
    int border;                                            // 0 zero, 1 reflect (source index mirrored)
    int dbg;                                               // ablation bits for tools/ (0 in the product)
    long long* tl;                                         // timeline buffer (NEMAR_TIMELINE builds) or null
    int sstride;                                           // source stride (1 or 2)
    int OHf, OWf, osy, osx;                                // destination plane extents, output stride
    int ncls;
    int ntaps[S16G_MAX_CLS], OH[S16G_MAX_CLS], OW[S16G_MAX_CLS], ooy[S16G_MAX_CLS], oox[S16G_MAX_CLS];
    short dy[S16G_MAX_CLS][S16G_MAX_TAPS], dx[S16G_MAX_CLS][S16G_MAX_TAPS];
    int wofs[S16G_MAX_CLS][S16G_MAX_TAPS];                 // offset of the tap inside one filter (pack only)
};

// Tile plan of a problem (host side; 0 = not eligible -> the caller keeps the exact-fp32 kernels)
struct S16gPlan {
    int ok;
    int MT, NT, ATAPS;                                     // 32-row tiles per wave, 32-pixel tiles per wave, LDS tap capacity
    int TW, RT, tiles_x, tiles_y, mblks, nchunks;
    int HR, HC, HCP, HCH, dymin, dxmin;
    size_t pack_words_per_class;                           // 16-byte words of one class's packed weights
    int CF;                                                // four classes fused into one workgroup per tile (conv_s16g.hip: s16g_kernel<..., CF = 1>)
};
S16gPlan nemar_s16g_plan(const S16gProblem& q);
size_t nemar_s16g_pack_bytes(const S16gProblem& q, const S16gPlan& pl);           // all classes + the max word
// w: weight tensor; element of output row m, reduction channel c, tap t = w[m * wsm + c * wsc + wofs[cls][t]]
void nemar_s16g_pack(const S16gProblem& q, const S16gPlan& pl, const float* w, long long wsm, long long wsc, void* packed,
                     hipStream_t st);
void nemar_s16g_conv(const S16gProblem& q, const S16gPlan& pl, const void* packed, hipStream_t st);

// ---- weight + bias gradient (conv_s16g_wgrad.hip) ----
bool nemar_s16g_wgrad_eligible(int N, int C0, int C1, int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad,
                               int pad_mode);
int nemar_s16g_wgrad_slabs(int N, int C0, int C1, int K, int OH, int W, int stride);
int nemar_s16g_wgrad_slabs_max(int N, int C, int K, int OH, int W, int stride);            // ... whatever the split C0 + C1 = C (workspace queries)         // slabs of K C R S (+ K for the bias) floats in `part`
// gw [K][C][KS][KS] += dW, gb [K] += sum gy (gb may be null); slabs summed in order (bitwise reproducible)
void nemar_s16g_wgrad(const float* x0, int C0, const float* x1, int C1, const float* gy, float* gw, float* gb, int N, int H, int W,
                      int K, int OH, int OW, int KS, int stride, int pad_mode, float* part, int dbg, hipStream_t st);
void nemar_sum_partials(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate, hipStream_t st);

#ifdef NEMAR_AB
void nemar_s16g_tune(int key, int value);      // tile-plan switches for A/B runs (nemar_tune keys 27, 28)
#endif

// measurement hook shared with conv_split16.hip (bench.py's roofline entry)
void nemar_s16g_timer(int on);
int nemar_s16g_timer_read(double* total_ms, double* total_flop);
