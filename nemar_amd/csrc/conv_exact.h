// The exact-fp32 implicit-GEMM engine (conv_exact.hip) as the operator entry points in conv.hip see it: the tap-table problem
// description, the packed-weight format's sizes, pack + launch.  Internal to the library (hidden visibility).
#pragma once
#include "common.h"

namespace nemar_exact {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;        // reduction depth per LDS stage (fwd/dgrad)
constexpr int MAX_TAPS = 64;  // 7x7 = 49
constexpr int BORDER_ZERO = 0, BORDER_REFLECT = 1;
constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3;
constexpr int ZERO_PAGE = 64;   // floats of zeros appended to every packed-weight buffer (target of masked gathers)

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ int reflect(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

struct TapTable {
    int n;
    int dyx[MAX_TAPS];   // (dy << 16) | (dx & 0xffff): one dword per tap so that a wave-uniform lookup is a scalar load
    short dy[MAX_TAPS], dx[MAX_TAPS];
    int wofs[MAX_TAPS];  // offset of the tap inside one [R][S] filter (pack kernel only)
};

struct IgemmParams {
    const float* src0; const float* src1; int C0, C1, Hs, Ws;
    const float* wp; int Mpad, M, Kred;
    const float* zero;   // >= 16 readable bytes of zeros (tail of the packed-weight buffer)
    long long* tl;       // optional timeline buffer (nemar_tune_ptr, A/B build): per-stage s_memtime stamps of workgroup 0
    int dbg;             // ablation switches (nemar_tune key 2, A/B build): 1 = skip staging, 2 = skip MFMAs, 4 = skip barriers
    const float* bias;
    float* dst0; float* dst1; int M0;
    int OH, OW, OHf, OWf, osy, ooy, osx, oox;
    int N, P;
    int sy, sx, border, act, pad;
    float slope;
    // ring mode (ring_p > 0): the "pixels" of this launch are the border ring of width ring_p around a ring_H x ring_W
    // image, in padded coordinates; OH*OW = ring length; results are atomically ADDED at the reflected in-image position
    int ring_p, ring_H, ring_W, ksplit;
    // ksplit > 1 (reduction split over grid.z: few, deep tiles): split z stores its partial tile to slab z of `part`
    // (same indexing as dst0) and nemar_sum_partials adds the slabs in order — no atomics, bitwise reproducible
    float* part; long long part_stride;
    // reflect data gradient without a ring launch (wave-specialised 16-byte-load kernel, 3x3 pad 1): pre-folded border rows
    // [2][3][N][K][Ws] and border column groups [2][N][K][Hs][4] of the source (reflect_aux_kernel)
    int rf; const float* rf_row; const float* rf_col;
    int xcd;             // XCD-aware workgroup -> tile mapping of the wave-specialised kernel (grid.x % 8 == 0)
    FastDiv fd_ohw, fd_ow, fd_cs;
    TapTable taps;
};

// Switches of the engine (common.h, NEMAR_SWITCH): variables defined in conv_exact.hip and written by nemar_tune in the A/B build,
// constants here in the product build.
#define NEMAR_EXACT_SWITCHES(X)                                                                                                     \
    X(int, g_cfg128, 0)      /* key 0: 0 = ws2 (all FAST shapes), 5 = ws2 without 16-byte B loads, 6 / 7 = ws2 experiments;     */ \
                             /*        128x128 only: 1 = 4-wave, 2 = 8-wave, 3 = 256x128, 4 = first-generation loader waves     */ \
    X(int, g_lds_pad, 0)     /* key 1: extra dynamic LDS bytes per workgroup (limits workgroups per CU)                         */ \
    X(int, g_dbg, 0)         /* key 2: ablation bits handed to the kernels                                                      */ \
    X(int, g_min_blocks, 384) /* key 6: workgroups below which the pixel/channel tile shrinks                                   */ \
    X(int, g_ws2_mt, 0)      /* key 7: force the wave-specialised kernel's channel tile (1, 2, 4 x 32)                          */ \
    X(int, g_deep64, 0)      /* key 10: 4-deep LDS ring for every FAST 64x64 launch (default: ring launches only)               */ \
    X(int, g_nl4_scalar, 1)  /* key 11: 4 loader waves for the gathered-B wave-specialised kernel                               */ \
    X(int, g_xcd_map, 1)     /* key 15: XCD-aware workgroup -> tile mapping in the wave-specialised kernels                     */ \
    X(int, g_adir, 0)        /* key 16: MFMA waves fetch their A fragments straight from global memory (measured 3 % SLOWER:    */ \
                             /*         370.8 vs 358.4 us on the 256->256 3x3 layer) / through LDS (0)                          */ \
    X(int, g_mt8, 0)         /* key 17: 256x128 tiles in the wave-specialised kernel: 0 off, 1 = 2 loader waves, 2 = 4          */ \
    X(int, g_ring, 3)        /* key 18: LDS ring depth of the wave-specialised 16-byte-load kernel (3, 4, 5)                    */ \
    X(long long*, g_tl, nullptr)   /* nemar_tune_ptr: device buffer for per-stage cycle stamps                                  */
#ifdef NEMAR_AB
#define NEMAR_EXACT_SWITCH_DECL(type, name, def) extern type name;
#else
#define NEMAR_EXACT_SWITCH_DECL(type, name, def) constexpr type name = def;
#endif
NEMAR_EXACT_SWITCHES(NEMAR_EXACT_SWITCH_DECL)

struct TileChoice { int bm, bn; };
TileChoice igemm_tile(int M, int P, int stages, int ksplit = 1);
int igemm_mpad(int M);
// packed weights are padded to a multiple of 256 channels (32 when M <= 32) so every tile config can read them
size_t packed_core_floats(int M, int Kred);
size_t packed_floats(int M, int Kred);
// Does this launch go to the wave-specialised kernel (128 pixels x 128 channels), and with 16-byte B loads?
bool route_ws2(const IgemmParams& p, bool* vec_out);
void launch_igemm(const IgemmParams& p, hipStream_t st);
void launch_pack(const float* w, float* wp, int M, int Cs, int wsm, int wsc, const TapTable& taps, hipStream_t st);

}  // namespace nemar_exact
