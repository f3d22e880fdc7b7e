// K2: weight gradient of the wide (> 32 output channel) convolutions, wave-specialised.
//
// Replaces the weight half of autograd's conv backward for nn.Conv2d / nn.ConvTranspose2d at reference
// models/networks.py:349-377,418-439,576-597 and models/stn/layers.py:85 (driven by loss.backward() at
// models/nemar_model.py:223,260).
//
//     gw[m][j] += sum_{p in split} gy[m][p] * src[c(j)][p (+) tap(j)],     j = c*R*S + r*S + s
//
// GEMM view: rows m = output channels, columns j = the weight tensor's own (c,r,s) order, reduction = output pixels,
// split across workgroups (grid.z).  Each split stores its tile to its own slab of the caller's workspace (plain stores)
// and nemar_sum_partials adds the slabs into the gradient buffer in split order: bitwise reproducible.  (part == nullptr:
// fp32 atomics straight into the gradient buffer, the round-1 scheme, kept for A/B timing behind nemar_tune(14, 0).)  Both operands are stored in
// memory with the REDUCTION index (pixels) contiguous, which is what makes this kernel different from the forward one:
//   * LDS tiles are [row][16 pixels] (64 B rows), filled by direct global->LDS loads: gy rows 16 B per lane, source
//     rows one gathered texel per lane (tap shift + zero/reflect border resolved per lane, masked lanes read a zero page);
//   * v_mfma_f32_32x32x2_f32 wants A[i][k], B[k][j] with k = lane>>5.  The order in which pixels are fed to the
//     reduction is free, so lane (i, kk) takes ONE ds_read_b128 = pixels 8g+4kk+{0..3} of its row and uses component s
//     in MFMA step s: four MFMA steps per 16-byte read per operand, against one ds_read_b32 per step per operand in
//     the classic layout;
//   * 64-byte rows would put a 16-lane ds_read_b128 group on 4 of the 16 bank slots; the loaders XOR the 16-byte chunk
//     index with (row>>2)&3 when they place the data (each lane picks its own global address, so this costs nothing),
//     and the readers apply the same XOR: conflict-free;
//   * 4 MFMA waves (64x64 each) issue only ds_read_b128 + MFMA; 4 loader waves own all vector-memory traffic, run two
//     stages ahead through a 3-deep LDS ring with a counted s_waitcnt vmcnt(N).  The MFMA waves prefetch the next
//     8-pixel group's fragments into a second register set before issuing the current group's 16 MFMAs, so LDS latency
//     and the stage barrier are hidden behind matrix work.
// Bias gradient (gb[m] += sum_p gy[m][p]) falls out of the A fragments of column-tile 0.
#include "common.h"

void nemar_sum_partials_pair(const float* part_a, long long stride_a, int splits_a, float* dst_a, long long n_a,
                             const float* part_b, long long stride_b, int splits_b, float* dst_b, long long n_b, bool accumulate, hipStream_t st);
void nemar_sum_partials(const float* part, long long stride, int splits, float* dst, long long n, bool accumulate,
                        hipStream_t st);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BORDER_REFLECT = 1;
constexpr int BN = 128, BKP = 16, NBUF = 3;
constexpr int NC = 4, NL = 4, NT = (NC + NL) * 64;
constexpr int B_PER = (BN / 4) / NL;    // scalar path: 4-row source wave-instructions per loader per stage (8)
constexpr int BV_PER = (BN / 16) / NL;  // vector path: 16-row (16 B per lane) source wave-instructions per loader (2)
static_assert(NL == 4, "loader l owns the source rows whose swizzle key (row>>2)&3 == l");

__device__ __attribute__((aligned(16))) float wg_zero_page[64];

__device__ __forceinline__ int reflect_i(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

struct Wgrad2Params {
    const float* src0; const float* src1; int C0, C1, Hs, Ws;
    const float* gy; int K, OH, OW;
    float* gw; int J;
    float* gb;
    float* part;       // [splits][K*J] slabs then [splits][K] bias slabs (nullptr: atomics into gw / gb)
    float* partb;
    int P, sy, sx, R, S, pad, border;
    int pix_per_split;
    int dbg;
    FastDiv fd_ohw, fd_ow, fd_rs, fd_s;
};

// VEC: the source tile is staged like the gy tile — 16 bytes per lane = 4 consecutive pixels of one (channel, tap)
// row, whose texels are consecutive in memory for a stride-1 layer (the address is only 4-byte aligned:
// global_load_lds_dwordx4 takes that).  A chunk whose shifted window would stick out of the source row by one texel
// (first chunk of an image row for dx = -1, last for dx = +1) is loaded from the clamped address; the MFMA wave that
// consumes it rotates the three good texels into place and inserts the mirrored texel (reflect) or 0 (zero padding).
// 4 wave-instructions per loader per stage instead of 10.
//
// Channel tile BM = WM*TMW*32 (the 4 MFMA waves form a WM x (4/WM) grid of TMW x TNW 32x32 MFMA tiles; the column tile is
// always 128): <2,2,2> = 128 channels, <2,1,2> = 64, <1,1,1> = 32 — narrower layers do not pay for empty MFMA rows.
template <bool VEC, int WM, int TMW, int TNW>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void wgrad2_kernel(Wgrad2Params p) {
    constexpr int WN = NC / WM;
    constexpr int BM = WM * TMW * 32;
    static_assert(WN * TNW * 32 == BN, "column tile is 128");
    constexpr int TILE = BM * BKP;                       // floats of the gy tile per stage; the source tile follows it
    constexpr int STAGE = TILE + BN * BKP;
    constexpr int A_INSTR = BM / 16;                     // 16-row gy wave-instructions per stage (8, 4, 2)
    constexpr int A_PER = A_INSTR >= NL ? A_INSTR / NL : 1;   // BM = 32: loaders 2,3 re-load rows 0..31 (same data, same slots)
    constexpr int LOADS = A_PER + (VEC ? BV_PER : B_PER);
    constexpr int NROW = VEC ? BV_PER : B_PER;
    __shared__ __attribute__((aligned(16))) float smem[NBUF * STAGE];
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.x * BM, j0 = blockIdx.y * BN;
    const int pbeg = blockIdx.z * p.pix_per_split;
    const int pend = min(p.P, pbeg + p.pix_per_split);
    const int nk = (pend - pbeg + BKP - 1) / BKP;
    if (nk <= 0) return;
    const int OHW = p.OH * p.OW, HW = p.Hs * p.Ws;
    if ((p.dbg & 256) && (((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >> 8) & 1))
        __builtin_amdgcn_s_sleep(15);   // experiment: de-phase the two workgroups of a CU

    if (wid >= NC) {
        // ================================ loader waves ================================
        const int l = wid - NC;
        // gy: instruction q covers rows 16q..16q+15 x 4 chunks; lane -> (row, stored chunk); swizzle key = (lane>>4)&3
        const int a_row = lane >> 2;
        const int a_pix = 4 * ((lane & 3) ^ ((lane >> 4) & 3));
        const int a_q0 = (l * A_PER) % A_INSTR;                 // this loader's first gy instruction
        const int a_m = m0 + a_q0 * 16 + a_row;                 // its channel; the next instruction is +16
        // source: instruction q = l + 4i covers rows 4q..4q+3 x 16 pixels; swizzle key of all its rows = l
        const int b_pix = 4 * (((lane & 15) >> 2) ^ l) + (lane & 3);
        const float* rowp[NROW];
        int rowns[NROW], rowt[NROW];
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            // vector path: instruction l*BV_PER + i covers rows 16*(l*BV_PER + i) .. +15, lane -> (row, stored chunk) as for gy
            const int j = VEC ? j0 + 16 * (l * BV_PER + i) + a_row : j0 + 4 * (l + 4 * i) + (lane >> 4);
            rowp[i] = nullptr;
            rowns[i] = 0;
            rowt[i] = 0;
            if (j < p.J) {
                const unsigned c = fd_div((unsigned)j, p.fd_rs);
                const unsigned t = (unsigned)j - c * (unsigned)(p.R * p.S);
                const unsigned r = fd_div(t, p.fd_s);
                const unsigned s = t - r * (unsigned)p.S;
                rowt[i] = (((int)r - p.pad) << 16) | (((int)s - p.pad) & 0xffff);
                if ((int)c < p.C0) {
                    rowp[i] = p.src0 + (size_t)c * HW;
                    rowns[i] = p.C0 * HW;
                } else {
                    rowp[i] = p.src1 + (size_t)((int)c - p.C0) * HW;
                    rowns[i] = p.C1 * HW;
                }
            }
        }
        // Scalar (gathered, 4 bytes per lane) source path: same idea as the VEC state machine below — the lane's gy chunk
        // and its source pixel advance 16 pixels per stage; divisions only at kernel start, row / image wrap by carry.
        int s_apix = pbeg + a_pix, s_arem = 0, s_buf = 0;
        const float* s_ga = p.gy;
        int s_bpix = pbeg + b_pix, s_oy = 0, s_ox = 0;
        const float* s_rb[NROW];
        int s_yo[NROW];
        if (!VEC) {
            {
                const unsigned up = (unsigned)min(s_apix, p.P - 1);
                const unsigned n = fd_div(up, p.fd_ohw);
                s_arem = (int)(up - n * (unsigned)OHW);
                s_ga = p.gy + ((size_t)n * p.K + a_m) * OHW + s_arem;
            }
            const unsigned up = (unsigned)min(s_bpix, p.P - 1);
            const unsigned n = fd_div(up, p.fd_ohw);
            const unsigned rem = up - n * (unsigned)OHW;
            s_oy = (int)fd_div(rem, p.fd_ow);
            s_ox = (int)rem - s_oy * p.OW;
#pragma unroll
            for (int i = 0; i < NROW; ++i) s_rb[i] = rowp[i] ? rowp[i] + (size_t)n * rowns[i] : nullptr;
        }
#define WG2_ROW_Y_SCALAR()                                                                                           \
        _Pragma("unroll") for (int i = 0; i < NROW; ++i) {                                                           \
            int y = s_oy * p.sy + (rowt[i] >> 16);                                                                   \
            if (p.border == BORDER_REFLECT) y = reflect_i(y, p.Hs);                                                  \
            s_yo[i] = ((unsigned)y < (unsigned)p.Hs && s_rb[i] != nullptr) ? y * p.Ws : -1;                          \
        }
        if (!VEC) { WG2_ROW_Y_SCALAR(); }
#define WG2_ISSUE_NEXT_SCALAR()                                                                                      \
        {                                                                                                            \
            float* const sb = smem + s_buf * STAGE;                                                                  \
            const bool pva = s_apix < pend;                                                                          \
            _Pragma("unroll") for (int q = 0; q < A_PER; ++q)                                                        \
                glds_b128((pva && a_m + 16 * q < p.K) ? s_ga + (size_t)(16 * q) * OHW : wg_zero_page,                \
                          sb + (a_q0 + q) * 256);                                                                    \
            const bool pvb = s_bpix < pend;                                                                          \
            const int bx = s_ox * p.sx;                                                                              \
            _Pragma("unroll") for (int i = 0; i < NROW; ++i) {                                                       \
                int x = bx + (int)(short)(rowt[i] & 0xffff);                                                         \
                bool inb = pvb && s_yo[i] >= 0;                                                                      \
                if (p.border == BORDER_REFLECT) x = reflect_i(x, p.Ws);                                              \
                else inb = inb && (unsigned)x < (unsigned)p.Ws;                                                      \
                glds_b32(inb ? s_rb[i] + (s_yo[i] + x) : wg_zero_page, sb + TILE + (l + 4 * i) * 64);                \
            }                                                                                                        \
            s_buf = s_buf + 1 == NBUF ? 0 : s_buf + 1;                                                               \
            s_apix += BKP;                                                                                           \
            s_arem += BKP;                                                                                           \
            s_ga += BKP;                                                                                             \
            while (s_arem >= OHW) {                                                                                  \
                s_arem -= OHW;                                                                                       \
                s_ga += (size_t)(p.K - 1) * OHW;                                                                     \
            }                                                                                                        \
            s_bpix += BKP;                                                                                           \
            s_ox += BKP;                                                                                             \
            if (s_ox >= p.OW) {                                                                                      \
                while (s_ox >= p.OW) {                                                                               \
                    s_ox -= p.OW;                                                                                    \
                    if (++s_oy == p.OH) {                                                                            \
                        s_oy = 0;                                                                                    \
                        _Pragma("unroll") for (int i = 0; i < NROW; ++i)                                             \
                            if (s_rb[i]) s_rb[i] += rowns[i];                                                        \
                    }                                                                                                \
                }                                                                                                    \
                WG2_ROW_Y_SCALAR();                                                                                  \
            }                                                                                                        \
        }
        // Stages are issued strictly in order, so the VEC loader is a state machine: the lane's 4-pixel chunk moves 16
        // pixels per stage (never across an image row mid-chunk: OW % 16 == 0), and everything that only changes with the
        // row or the image is recomputed there.  (The loaders are the last waves at the stage barrier; see conv.hip.)
        int v_pix = pbeg + a_pix;                       // first pixel of this lane's chunk in the next stage
        int v_buf = 0;
        int v_ox = 0, v_oy = 0;
        const float* v_ga = p.gy;                       // gy + (n*K + a_m)*OHW + rem of the chunk
        const float* v_rb[NROW];                        // source row base incl. the image offset
        int v_yo[NROW];                                 // y * Ws of the row's tap at the current oy, or -1: outside (zero border)
        if (VEC) {
            const unsigned upix = (unsigned)min(v_pix, p.P - 1);
            const unsigned n = fd_div(upix, p.fd_ohw);
            const unsigned rem = upix - n * (unsigned)OHW;
            v_oy = (int)fd_div(rem, p.fd_ow);
            v_ox = (int)rem - v_oy * p.OW;
            v_ga = p.gy + ((size_t)n * p.K + a_m) * OHW + rem;
#pragma unroll
            for (int i = 0; i < NROW; ++i) v_rb[i] = rowp[i] ? rowp[i] + (size_t)n * rowns[i] : nullptr;
        }
#define WG2_ROW_Y()                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NROW; ++i) {                                                           \
            int y = v_oy * p.sy + (rowt[i] >> 16);                                                                   \
            if (p.border == BORDER_REFLECT) y = reflect_i(y, p.Hs);                                                  \
            v_yo[i] = ((unsigned)y < (unsigned)p.Hs && v_rb[i] != nullptr) ? y * p.Ws : -1;                          \
        }
        if (VEC) { WG2_ROW_Y(); }
#define WG2_ISSUE_NEXT_VEC()                                                                                         \
        {                                                                                                            \
            float* const sb = smem + v_buf * STAGE;                                                                  \
            const bool pv = v_pix < pend;                                                                            \
            _Pragma("unroll") for (int q = 0; q < A_PER; ++q)                                                        \
                glds_b128((pv && a_m + 16 * q < p.K) ? v_ga + (size_t)(16 * q) * OHW : wg_zero_page,                 \
                          sb + (a_q0 + q) * 256);                                                                    \
            _Pragma("unroll") for (int i = 0; i < NROW; ++i) {                                                       \
                const int x = min(max(v_ox + (int)(short)(rowt[i] & 0xffff), 0), p.Ws - 4);                          \
                glds_b128((pv && v_yo[i] >= 0) ? v_rb[i] + (v_yo[i] + x) : wg_zero_page,                             \
                          sb + TILE + (l * BV_PER + i) * 256);                                                       \
            }                                                                                                        \
            v_buf = v_buf + 1 == NBUF ? 0 : v_buf + 1;                                                               \
            v_pix += BKP;                                                                                            \
            v_ga += BKP;                                                                                             \
            v_ox += BKP;                                                                                             \
            if (v_ox >= p.OW) {                                                                                      \
                v_ox -= p.OW;                                                                                        \
                if (++v_oy == p.OH) {                                                                                \
                    v_oy = 0;                                                                                        \
                    v_ga += (size_t)(p.K - 1) * OHW;                                                                 \
                    _Pragma("unroll") for (int i = 0; i < NROW; ++i)                                                 \
                        if (v_rb[i]) v_rb[i] += rowns[i];                                                            \
                }                                                                                                    \
                WG2_ROW_Y();                                                                                         \
            }                                                                                                        \
        }
#define WG2_ISSUE(ks_)  { if (VEC) WG2_ISSUE_NEXT_VEC() else WG2_ISSUE_NEXT_SCALAR() }   /* stages in order */
#define WG2_WAIT_ONE_IN_FLIGHT() __builtin_amdgcn_s_waitcnt(0x0F70 | (LOADS & 15) | ((LOADS >> 4) << 14))
        WG2_ISSUE(0);
        if (nk > 1) {
            WG2_ISSUE(1);
            WG2_WAIT_ONE_IN_FLIGHT();
        } else {
            wait_vmem();
        }
        __builtin_amdgcn_s_barrier();                 // stage 0 is in LDS
        if (nk > 2) WG2_ISSUE(2);
        for (int ks = 0; ks < nk; ++ks) {
            // the barrier of iteration ks needs stage ks+1 landed; stage ks+2 (if any) may stay in flight
            if (ks + 2 < nk) WG2_WAIT_ONE_IN_FLIGHT();
            else wait_vmem();
            __builtin_amdgcn_s_barrier();             // also: every MFMA wave has finished reading buffer ks % NBUF
            if (ks + 3 < nk) WG2_ISSUE(ks + 3);
        }
#undef WG2_ISSUE
#undef WG2_ISSUE_NEXT_SCALAR
#undef WG2_ROW_Y_SCALAR
#undef WG2_ISSUE_NEXT_VEC
#undef WG2_ROW_Y
#undef WG2_WAIT_ONE_IN_FLIGHT
        return;
    }

    // ================================ MFMA waves ================================
    if (p.dbg & 128) __builtin_amdgcn_s_setprio(3);           // experiment: issue priority over the loader waves
    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int sw = (l31 >> 2) & 3;
    const int arow = (wm * TMW * 32 + l31) * BKP;            // + i * 32 * BKP for MFMA row tile i
    const int brow = TILE + (wn * TNW * 32 + l31) * BKP;     // + j * 32 * BKP for MFMA column tile j
    const int co0 = ((0 + lhi) ^ sw) << 2, co1 = ((2 + lhi) ^ sw) << 2;   // pixel groups 0 and 1 of a stage
    const bool do_bias = p.gb != nullptr && blockIdx.y == 0 && wn == 0;
    // VEC: horizontal tap offset of this lane's two source rows, and the x position of the current stage in its image row
    int dxn[TNW], oxs = 0;
#pragma unroll
    for (int j = 0; j < TNW; ++j) dxn[j] = 0;
    const bool refl = p.border == BORDER_REFLECT;
    if (VEC) {
        const unsigned RS = (unsigned)(p.R * p.S);
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const unsigned jj = (unsigned)(j0 + (wn * TNW + j) * 32 + l31);
            const unsigned t = jj - fd_div(jj, p.fd_rs) * RS;
            dxn[j] = (int)(t - fd_div(t, p.fd_s) * (unsigned)p.S) - p.pad;
        }
        const unsigned rem = (unsigned)pbeg - fd_div((unsigned)pbeg, p.fd_ohw) * (unsigned)OHW;
        oxs = (int)(rem - fd_div(rem, p.fd_ow) * (unsigned)p.OW);
    }

    f32x16 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i) bsum[i] = 0.f;
    f32x4 a0[TMW], b0[TNW], a1[TMW], b1[TNW];

#define WG2_READ(buf_, co_, A_, B_)                                                                            \
    {                                                                                                              \
        const float* sb = smem + (buf_) * STAGE;                                                                   \
        _Pragma("unroll") for (int i = 0; i < TMW; ++i)                                                            \
            A_[i] = *reinterpret_cast<const f32x4*>(sb + arow + i * 32 * BKP + (co_));                             \
        _Pragma("unroll") for (int j = 0; j < TNW; ++j)                                                            \
            B_[j] = *reinterpret_cast<const f32x4*>(sb + brow + j * 32 * BKP + (co_));                             \
    }
// The border patches and the bias sum are wave-uniform rarities (first/last chunk of an image row; column-tile 0 only).
// Left to the compiler they are if-converted into v_cndmask / v_pk_add work that every wave executes in every block,
// scheduled BETWEEN the MFMAs (each stray VALU slot between MFMAs costs 6-40 cycles of matrix pipe).  The empty volatile
// asm keeps them real branches; the sched_barrier keeps the 16 MFMAs back to back.
#define WG2_MFMA(A_, B_, G_)                                                                                   \
    {                                                                                                              \
        if (VEC && (G_) == 0 && oxs == 0) {                       /* first chunk of an image row: dx = -1 rows */  \
            asm volatile("" ::: "memory");                                                                         \
            _Pragma("unroll") for (int j = 0; j < TNW; ++j) {                                                      \
                const f32x4 u = B_[j];                                                                             \
                const bool fx = lhi == 0 && dxn[j] < 0;                                                            \
                B_[j][0] = fx ? (refl ? u[1] : 0.f) : u[0];                                                        \
                B_[j][1] = fx ? u[0] : u[1];                                                                       \
                B_[j][2] = fx ? u[1] : u[2];                                                                       \
                B_[j][3] = fx ? u[2] : u[3];                                                                       \
            }                                                                                                      \
        }                                                                                                          \
        if (VEC && (G_) == 1 && oxs == p.OW - BKP) {              /* last chunk of an image row: dx = +1 rows */   \
            asm volatile("" ::: "memory");                                                                         \
            _Pragma("unroll") for (int j = 0; j < TNW; ++j) {                                                      \
                const f32x4 u = B_[j];                                                                             \
                const bool fx = lhi == 1 && dxn[j] > 0;                                                            \
                B_[j][0] = fx ? u[1] : u[0];                                                                       \
                B_[j][1] = fx ? u[2] : u[1];                                                                       \
                B_[j][2] = fx ? u[3] : u[2];                                                                       \
                B_[j][3] = fx ? (refl ? u[2] : 0.f) : u[3];                                                        \
            }                                                                                                      \
        }                                                                                                          \
        if (do_bias) {                                                                                             \
            asm volatile("" ::: "memory");                                                                         \
            _Pragma("unroll") for (int i = 0; i < TMW; ++i)                                                        \
                bsum[i] += (A_[i][0] + A_[i][1]) + (A_[i][2] + A_[i][3]);                                          \
        }                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                              \
            _Pragma("unroll") for (int i = 0; i < TMW; ++i)                                                        \
                _Pragma("unroll") for (int j = 0; j < TNW; ++j)                                                    \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[i][s], B_[j][s], acc[i][j], 0, 0, 0);      \
    }

    __builtin_amdgcn_s_barrier();                     // stage 0 is in LDS
    WG2_READ(0, co0, a0, b0);
    int buf = 0;
    for (int ks = 0; ks < nk; ++ks) {
        WG2_READ(buf, co1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        WG2_MFMA(a0, b0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): this wave is done reading buffer `buf`
        __builtin_amdgcn_s_barrier();                 // stage ks+1 has landed; buffer `buf` is released to the loaders
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        if (ks + 1 < nk) WG2_READ(buf, co0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        WG2_MFMA(a1, b1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (VEC) oxs = oxs + BKP == p.OW ? 0 : oxs + BKP;
    }
#undef WG2_READ
#undef WG2_MFMA

    float* const gw = p.part ? p.part + (size_t)blockIdx.z * ((size_t)p.K * p.J) : p.gw;
    if (do_bias) {
        // lanes l and l+32 hold the two pixel-halves of channel row l31
        float* const gb = p.part ? p.partb + (size_t)blockIdx.z * p.K : p.gb;
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            const int m = m0 + (wm * TMW + i) * 32 + l31;
            if (lhi == 0 && m < p.K) {
                if (p.part) gb[m] = v;
                else atomicAdd(gb + m, v);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int jj = j0 + (wn * TNW + j) * 32 + l31;
        if (jj >= p.J) continue;
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TMW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < p.K) {
                    if (p.part) gw[(size_t)m * p.J + jj] = acc[i][j][r];
                    else atomicAdd(gw + (size_t)m * p.J + jj, acc[i][j][r]);
                }
            }
    }
}

template <int WM, int TMW, int TNW>
void launch_wgrad2(const Wgrad2Params& p, bool vec, dim3 grid, hipStream_t st) {
    if (vec) hipLaunchKernelGGL((wgrad2_kernel<true, WM, TMW, TNW>), grid, dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((wgrad2_kernel<false, WM, TMW, TNW>), grid, dim3(NT), 0, st, p);
}

}  // namespace

// Shapes this kernel takes: layers whose gy rows can be read in aligned 16-byte chunks that never straddle two
// images (OH*OW % 4 == 0).  Everything else stays on the VGPR-staged kernel in conv.hip.
bool nemar_wgrad2_eligible(int K, int OH, int OW, const float* gy) {
    return K > 4 && (OH * OW) % 4 == 0 && (reinterpret_cast<uintptr_t>(gy) & 15) == 0;
}

// Split plan shared by the workspace query and the launch.  The pixel reduction is split so that the grid is ONE full
// round of resident workgroups (2 per CU x 256 CUs): every workgroup starts and ends together, so a grid of 1.1 or 2.04
// rounds pays for 2 or 3.  target_blocks is that capacity; the split count is rounded DOWN to fit it, >= 8 stages per split.
void nemar_wgrad2_plan(int K, int J, int P, int target_blocks, int* splits_out, int* pix_per_split_out) {
    const int BM = K <= 32 ? 32 : K <= 64 ? 64 : 128;
    const int mt = nemar_cdiv(K, BM), jt = nemar_cdiv(J, BN);
    if (BM < 128) target_blocks *= 2;   // the narrower tiles need half the LDS and registers: 4 workgroups per CU
    int splits = target_blocks / (mt * jt);
    const int max_splits = nemar_cdiv(P, BKP * 8);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const int pps = nemar_cdiv(nemar_cdiv(P, splits), BKP) * BKP;
    *pix_per_split_out = pps;
    *splits_out = nemar_cdiv(P, pps);       // every split owns at least one pixel: every slab is written in full
}

void nemar_wgrad2_launch(const float* x0, int C0, const float* x1, int C1, const float* gy, float* gw, float* gb, int N,
                         int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad, int pad_mode,
                         int target_blocks, bool vec_ok, int dbg, float* part, hipStream_t st) {
    Wgrad2Params p;
    p.dbg = dbg;
    p.src0 = x0; p.src1 = x1; p.C0 = C0; p.C1 = C1; p.Hs = H; p.Ws = W;
    p.gy = gy; p.K = K; p.OH = OH; p.OW = OW;
    p.gw = gw; p.J = (C0 + C1) * R * S; p.gb = gb;
    p.P = N * OH * OW; p.sy = stride; p.sx = stride; p.R = R; p.S = S; p.pad = pad; p.border = pad_mode;
    p.fd_ohw = make_fastdiv(OH * OW); p.fd_ow = make_fastdiv(OW);
    p.fd_rs = make_fastdiv(R * S); p.fd_s = make_fastdiv(S);
    const int BM = K <= 32 ? 32 : K <= 64 ? 64 : 128;
    const int mt = nemar_cdiv(K, BM), jt = nemar_cdiv(p.J, BN);
    int splits;
    nemar_wgrad2_plan(K, p.J, p.P, target_blocks, &splits, &p.pix_per_split);
    const size_t KJ = (size_t)K * p.J;
    p.part = part;
    p.partb = part ? part + (size_t)splits * KJ : nullptr;
    // 16-byte source loads: stride 1, image rows that are whole 16-pixel stages, horizontal tap offsets within +-1
    const bool vec = vec_ok && stride == 1 && OW % BKP == 0 && W == OW && pad <= 1 && S - 1 - pad <= 1;
    const dim3 grid(mt, jt, splits);
    if (BM == 128) launch_wgrad2<2, 2, 2>(p, vec, grid, st);
    else if (BM == 64) launch_wgrad2<2, 1, 2>(p, vec, grid, st);
    else launch_wgrad2<1, 1, 1>(p, vec, grid, st);
    if (part) nemar_sum_partials_pair(part, (long long)KJ, splits, gw, (long long)KJ, gb ? p.partb : nullptr, K, splits, gb, K, true, st);
}
