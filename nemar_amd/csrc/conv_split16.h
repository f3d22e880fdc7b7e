// Internal interface of conv_split16.hip (3x3 stride-1 convolutions of wide layers on the bf16 matrix pipe with a three-way
// operand split: fp32-equivalent products at 16/6 of the fp32-MFMA rate).  Called by conv.hip's operators only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

enum { SPLIT16_ZERO = 0, SPLIT16_REFLECT = 1, SPLIT16_DGRAD_REFLECT = 2 };

// M = output channels of the launch, Cred = reduction channels, source [N, Cred, H, W] -> destination [N, M, H, W]
// variant: the nemar_tune(21) setting the call will run under (4 fp16 x 3, 3 bf16 x 6, 0 first generation): LDS budgets differ
bool nemar_split16_eligible(int N, int H, int W, int M, int Cred, int R, int S, int stride, int pad, int mode, int variant);
size_t nemar_split16_scratch_bytes(int N, int Cred, int H, int W);      // split activation planes
int nemar_split16_ksplit(int N, int H, int W, int M, int Cred);          // reduction runs per tile (few-tile layers)
size_t nemar_split16_scratch_total(int N, int H, int W, int M, int Cred, int OH, int OW);     // planes + slabs of those runs
size_t nemar_split16_pack_bytes(int M, int Cred, int KS);               // split, tile-ordered weights (KS x KS taps)
// w [K, C, KS, KS].  dgrad == 0: M = K rows, reduction over C.  dgrad != 0: M = C rows, reduction over K, taps flipped.
void nemar_split16_pack(const float* w, void* packed, int K, int C, int KS, int dgrad, int variant, hipStream_t st);
// src [N, Cred, Hs, Ws_src] fp32, seen through a padded H x W domain whose position (src_pad, src_pad) is source texel (0, 0)
// -> dst [N, M, OH, OW] fp32 (+ bias[M] when non-null): outputs at rows >= OH / columns >= OW of the domain are not stored.
// 3x3: Hs = OH = H, Ws_src = OW = W, src_pad = 1.  4x4 / pad 1 forward: Hs = H, OH = H - 1, src_pad = 1; its data gradient (a full
// correlation of gy [H-1, W-1]): Hs = H - 1, OH = H, src_pad = 2.  `scratch` >= nemar_split16_scratch_bytes
// -> true when dual_g_out was filled (the data gradient's split pass also wrote the weight gradient's gy planes)
bool nemar_split16_conv(const float* src, const void* packed, const float* bias, float* dst, int N, int H, int W, int M, int Cred,
                         int KS, int src_pad, int Hs, int Ws_src, int OH, int OW, int mode, void* scratch, int xcd_map, int variant,
                         long long* tl, void* dual_g_out, hipStream_t st);

// ---- weight gradient of the same layers (conv_split16_wgrad.hip) ----
bool nemar_split16_wgrad_eligible(int N, int C, int H, int W, int K, int R, int S, int stride, int pad);
size_t nemar_split16_wgrad_scratch_bytes(int N, int C, int H, int W, int K, int KS);     // split gy (KS shifts) and padded x planes
int nemar_split16_wgrad_splits(int N, int C, int H, int W, int K, int KS);               // slabs of K C KS KS floats the caller provides
// gw [K][C][KS][KS] += dW from x [N,C,H,W] and gy [N,K,H+3-KS,W+3-KS]; slabs summed in order (bitwise reproducible)
#ifdef NEMAR_AB
void nemar_split16_wgrad_tune(int one_copy);          // nemar_tune(34): 1 (default) one gy copy + in-register shifts, 0 KS copies
#endif
// g_planes != NULL: the G_0 planes of gy already exist (nemar_split16_dual_split wrote them, scaled by the max words hinted for gy)
// x_planes != NULL: the X planes of x already exist (nemar_instnorm_fwd_planes wrote them, scaled by the max words hinted for x)
// the NEXT nemar_split16_wgrad call on this thread also reduces gb[K] += sum over the batch of bias_partials [N, K], inside its slab-sum launch
void nemar_split16_wgrad_set_bias(const float* bias_partials, float* gb);
void nemar_sum_partials_pair(const float* part_a, long long stride_a, int splits_a, float* dst_a, long long n_a,
                             const float* part_b, long long stride_b, int splits_b, float* dst_b, long long n_b, bool accumulate, hipStream_t st);
void nemar_split16_wgrad(const float* x, const float* gy, float* gw, int N, int C, int H, int W, int K, int KS, int reflect,
                         void* scratch, float* part, int xcd_map, const void* g_planes, const void* x_planes, hipStream_t st);
size_t nemar_split16_wgrad_g_bytes(int N, int H, int W, int K, int KS);      // bytes of the G_0 planes (two 16-bit planes)
// one pass over gy [N, K, H, W] (3x3 / pad 1 layers): the data gradient's channel-blocked padded planes (mode SPLIT16_ZERO or
// SPLIT16_DGRAD_REFLECT, bit-identical to split_planes_kernel's) into `dplanes` AND the weight gradient's G_0 planes into `gplanes`
void nemar_split16_dual_split(const float* gy, void* dplanes, void* gplanes, int N, int K, int H, int W, int mode, const unsigned* maxbits,
                              int mstride, hipStream_t st);
void nemar_sum_partials(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate, hipStream_t st);
void nemar_sum_partials_two(const float* part, long long stride, int splits, float* d0, float* d1, int N, int C0, int C1, int HW, hipStream_t st);
void nemar_sum_partials_act(const float* part, long long stride, int splits, float* dst, long long n, int act, float slope, hipStream_t st);

// ---- max |t| of a source tensor (the fp16 form's power-of-two scale follows from it) ----
// The scale is PER SAMPLE: out = `samples` words, ZERO on entry (the kernel takes an atomic max of the finite elements of sample i
// — `per` consecutive floats — into word i)
void nemar_split16_absmax(const float* x, int samples, long long per, void* out, hipStream_t st);
void nemar_split16_set_hint(const void* tensor, const void* word, int count);  // word == NULL clears; count = words (N or 1)
const unsigned* nemar_split16_hint(const void* tensor, int* count);
void nemar_split16_set_planes_hint(const void* tensor, const void* planes, int N, int C, int H, int W, int kind);      // kind: SPLIT16_* content
// epilogue side inputs of the NEXT nemar_split16_conv call on this thread: addend [N, M, OH, OW] added to the result, max_words
// (NEMAR_MAX_WORDS(N)) <- per-sample max |result|.  Honoured only when the tile's reduction is not split over workgroups:
// nemar_split16_epilogue_done() says whether the last call did both.  set_epilogue(nullptr, nullptr) clears.
void nemar_split16_set_epilogue(const float* addend, void* max_words);
int nemar_split16_epilogue_done();
const unsigned* nemar_split16_source_max(const float* src, int N, long long per, unsigned* own, int* stride, hipStream_t st);

// ---- measurement hook: HIP events on the launch stream around the main kernel of every nemar_split16_conv call while enabled ----
void nemar_split16_timer(int on);
int nemar_split16_timer_read(double* total_ms, double* total_flop);      // -> launches timed since enabled (their summed duration
                                                                         // and algorithmic flop); synchronises on the events; resets
