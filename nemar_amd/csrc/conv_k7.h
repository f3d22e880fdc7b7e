// Internal interface of conv_k7.hip: the 7x7 / stride-1 / pad-3 convolutions between a FEW channels (<= 4: RGB images, flow heads) and
// MANY (a multiple of 32) on the 16-bit matrix pipe at fp32 accuracy — the translation net's stem (3 -> 64) and head (64 -> 3),
// reference models/networks.py:349-350 and :375-377.  Forward, data gradient and weight gradient of both; called by conv.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// ---- weight gradient of both layers: G[m][c][dy][dx] = sum over pixels of Big[m][p] * Small[c][p + (dy, dx)] ----------------
// stem (C <= 4 input channels, K = 32 n outputs): Big = gy, Small = padded x;  head (C = 32 n inputs, K <= 4 outputs): Big = padded x,
// Small = gy (taps flipped on the way out).  Slabs of K * C * 49 floats in `part`, summed in order by nemar_sum_partials.
bool nemar_k7_wgrad_eligible(int N, int C, int H, int W, int K, int R, int S, int stride, int pad);
int nemar_k7_wgrad_slabs(int N, int C, int H, int W, int K);
size_t nemar_k7_wgrad_floats(int N, int C, int H, int W, int K);          // slabs (+ bias partials for the stem) + the max words of the small tensor
// -> true: gb (when given) has been accumulated too (stem: the kernel's Big operand IS gy); false: the caller reduces the bias itself
bool nemar_k7_wgrad(const float* x, const float* gy, float* gw, float* gb, int N, int C, int H, int W, int K, int pad_mode, float* part,
                    hipStream_t st);
void nemar_sum_partials(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate, hipStream_t st);
void nemar_sum_partials_pair(const float* part_a, long long stride_a, int splits_a, float* dst_a, long long n_a,
                             const float* part_b, long long stride_b, int splits_b, float* dst_b, long long n_b, bool accumulate, hipStream_t st);


// ---- few -> many: out[n][m][y][x] = bias[m] + sum_{c, dy, dx} Wt[m][c][dy][dx] * Small[n][c][y + dy][x + dx]  (stem forward; head data
// gradient with Wt[c][k][dy][dx] = w[k][c][6 - dy][6 - dx], Small = gy through a zero border).  M = 32 n rows, Cs <= 4.
bool nemar_k7_fm_eligible(int Cs, int M, int R, int S, int stride, int pad);
size_t nemar_k7_fm_pack_floats(int M);
// element (m, c, tap) of the weight tensor = w[m * wsm + c * wsc + tap]; flip: taps mirrored (data gradient)
void nemar_k7_fm_pack(const float* w, long long wsm, long long wsc, int flip, int M, int Cs, void* packed, hipStream_t st);
// src [N][Cs][Hs][Ws] seen through a border of `pad` texels (reflect: mirrored, else zero); dst [N][M][Hv][Wv], view = (Hv + 6) x (Wv + 6)
void nemar_k7_fm_conv(const float* src, int Cs, int Hs, int Ws, int pad, int reflect, const void* packed, const float* bias, float* dst,
                      int M, int N, int Hv, int Wv, int act, float slope, int fold, hipStream_t st);
// fold = 1: dst is the IMAGE-domain gradient of a reflect-padded layer (Hv x Wv = the image, src through a 6-texel zero border): the
// mirrored contributions are accumulated inside the kernel.  Needs nemar_k7_fm_fold_ok(Hv, Wv).
static inline bool nemar_k7_fm_fold_ok(int H, int W) { return H % 4 == 0 && H >= 8 && W % 64 == 0; }
