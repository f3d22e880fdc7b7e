// K2: weight gradient of the wide (> 32 output channel) convolutions, wave-specialised.
//
// Replaces the weight half of autograd's conv backward for nn.Conv2d / nn.ConvTranspose2d at reference
// models/networks.py:349-377,418-439,576-597 and models/stn/layers.py:85 (driven by loss.backward() at
// models/nemar_model.py:223,260).
//
//     gw[m][j] += sum_{p in split} gy[m][p] * src[c(j)][p (+) tap(j)],     j = c*R*S + r*S + s
//
// GEMM view: rows m = output channels, columns j = the weight tensor's own (c,r,s) order, reduction = output pixels,
// split across workgroups (grid.z) with fp32 atomics into the caller's gradient buffer.  Both operands are stored in
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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BORDER_REFLECT = 1;
constexpr int BM = 128, BN = 128, BKP = 16, NBUF = 3;
constexpr int NC = 4, NL = 4, NT = (NC + NL) * 64;
constexpr int TILE = BM * BKP;          // floats per operand per stage (BM == BN)
constexpr int STAGE = 2 * TILE;
constexpr int A_PER = (BM / 16) / NL;   // 16-row gy wave-instructions per loader per stage (2)
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
    int P, sy, sx, R, S, pad, border;
    int pix_per_split;
    FastDiv fd_ohw, fd_ow, fd_rs, fd_s;
};

// VEC: the source tile is staged like the gy tile — 16 bytes per lane = 4 consecutive pixels of one (channel, tap)
// row, whose texels are consecutive in memory for a stride-1 layer (the address is only 4-byte aligned:
// global_load_lds_dwordx4 takes that).  A chunk whose shifted window would stick out of the source row by one texel
// (first chunk of an image row for dx = -1, last for dx = +1) is loaded from the clamped address; the MFMA wave that
// consumes it rotates the three good texels into place and inserts the mirrored texel (reflect) or 0 (zero padding).
// 4 wave-instructions per loader per stage instead of 10.
template <bool VEC>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void wgrad2_kernel(Wgrad2Params p) {
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

    if (wid >= NC) {
        // ================================ loader waves ================================
        const int l = wid - NC;
        // gy: instruction q covers rows 16q..16q+15 x 4 chunks; lane -> (row, stored chunk); swizzle key = (lane>>4)&3
        const int a_row = lane >> 2;
        const int a_pix = 4 * ((lane & 3) ^ ((lane >> 4) & 3));
        const int a_m = m0 + (l * A_PER) * 16 + a_row;          // first instruction's channel; the next is +16
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
#define WG2_ISSUE(ks_)                                                                                           \
        {                                                                                                            \
            const int pb = pbeg + (ks_) * BKP;                                                                       \
            float* const sb = smem + ((ks_) % NBUF) * STAGE;                                                         \
            {                                                                                                        \
                const int pix = pb + a_pix;                                                                          \
                const unsigned n = fd_div((unsigned)pix, p.fd_ohw);                                                  \
                const unsigned rem = (unsigned)pix - n * (unsigned)OHW;                                              \
                const float* g = p.gy + ((size_t)n * p.K + a_m) * OHW + rem;                                         \
                _Pragma("unroll") for (int q = 0; q < A_PER; ++q)                                                    \
                    glds_b128((pix < pend && a_m + 16 * q < p.K) ? g + (size_t)(16 * q) * OHW : wg_zero_page,        \
                              sb + (l * A_PER + q) * 256);                                                           \
            }                                                                                                        \
            if (VEC) {                                                                                               \
                const int pix = pb + a_pix;                                                                          \
                const bool pv = pix < pend;                                                                          \
                const unsigned upix = pv ? (unsigned)pix : 0u;                                                       \
                const unsigned n = fd_div(upix, p.fd_ohw);                                                           \
                const unsigned rem = upix - n * (unsigned)OHW;                                                       \
                const unsigned oy = fd_div(rem, p.fd_ow);                                                            \
                const int by = (int)oy * p.sy, bx = (int)(rem - oy * (unsigned)p.OW);                                \
                _Pragma("unroll") for (int i = 0; i < NROW; ++i) {                                                   \
                    int y = by + (rowt[i] >> 16);                                                                    \
                    const int x = min(max(bx + (int)(short)(rowt[i] & 0xffff), 0), p.Ws - 4);                        \
                    bool inb = pv && rowp[i] != nullptr;                                                             \
                    if (p.border == BORDER_REFLECT) y = reflect_i(y, p.Hs);                                          \
                    else inb = inb && (unsigned)y < (unsigned)p.Hs;                                                  \
                    glds_b128(inb ? rowp[i] + (size_t)n * rowns[i] + (y * p.Ws + x) : wg_zero_page,                  \
                              sb + TILE + (l * BV_PER + i) * 256);                                                   \
                }                                                                                                    \
            } else {                                                                                                 \
                const int pix = pb + b_pix;                                                                          \
                const bool pv = pix < pend;                                                                          \
                const unsigned upix = pv ? (unsigned)pix : 0u;                                                       \
                const unsigned n = fd_div(upix, p.fd_ohw);                                                           \
                const unsigned rem = upix - n * (unsigned)OHW;                                                       \
                const unsigned oy = fd_div(rem, p.fd_ow);                                                            \
                const int by = (int)oy * p.sy, bx = (int)(rem - oy * (unsigned)p.OW) * p.sx;                         \
                _Pragma("unroll") for (int i = 0; i < NROW; ++i) {                                                   \
                    int y = by + (rowt[i] >> 16), x = bx + (int)(short)(rowt[i] & 0xffff);                           \
                    bool inb = pv && rowp[i] != nullptr;                                                             \
                    if (p.border == BORDER_REFLECT) {                                                                \
                        y = reflect_i(y, p.Hs);                                                                      \
                        x = reflect_i(x, p.Ws);                                                                      \
                    } else {                                                                                         \
                        inb = inb && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;                   \
                    }                                                                                                \
                    glds_b32(inb ? rowp[i] + (size_t)n * rowns[i] + (y * p.Ws + x) : wg_zero_page,                   \
                             sb + TILE + (l + 4 * i) * 64);                                                          \
                }                                                                                                    \
            }                                                                                                        \
        }
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
#undef WG2_WAIT_ONE_IN_FLIGHT
        return;
    }

    // ================================ MFMA waves ================================
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int sw = (l31 >> 2) & 3;
    const int arow0 = ((wm * 2 + 0) * 32 + l31) * BKP, arow1 = ((wm * 2 + 1) * 32 + l31) * BKP;
    const int brow0 = TILE + ((wn * 2 + 0) * 32 + l31) * BKP, brow1 = TILE + ((wn * 2 + 1) * 32 + l31) * BKP;
    const int co0 = ((0 + lhi) ^ sw) << 2, co1 = ((2 + lhi) ^ sw) << 2;   // pixel groups 0 and 1 of a stage
    const bool do_bias = p.gb != nullptr && blockIdx.y == 0 && wn == 0;
    // VEC: horizontal tap offset of this lane's two source rows, and the x position of the current stage in its image row
    int dxn0 = 0, dxn1 = 0, oxs = 0;
    const bool refl = p.border == BORDER_REFLECT;
    if (VEC) {
        const int ja = j0 + (wn * 2 + 0) * 32 + l31, jb = ja + 32;
        const unsigned RS = (unsigned)(p.R * p.S);
        const unsigned ta = (unsigned)ja - fd_div((unsigned)ja, p.fd_rs) * RS, tb = (unsigned)jb - fd_div((unsigned)jb, p.fd_rs) * RS;
        dxn0 = (int)(ta - fd_div(ta, p.fd_s) * (unsigned)p.S) - p.pad;
        dxn1 = (int)(tb - fd_div(tb, p.fd_s) * (unsigned)p.S) - p.pad;
        const unsigned rem = (unsigned)pbeg - fd_div((unsigned)pbeg, p.fd_ohw) * (unsigned)OHW;
        oxs = (int)(rem - fd_div(rem, p.fd_ow) * (unsigned)p.OW);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum0 = 0.f, bsum1 = 0.f;
    f32x4 a0[2], b0[2], a1[2], b1[2];

#define WG2_READ(buf_, co_, A_, B_)                                            \
    {                                                                              \
        const float* sb = smem + (buf_) * STAGE;                                   \
        A_[0] = *reinterpret_cast<const f32x4*>(sb + arow0 + (co_));               \
        A_[1] = *reinterpret_cast<const f32x4*>(sb + arow1 + (co_));               \
        B_[0] = *reinterpret_cast<const f32x4*>(sb + brow0 + (co_));               \
        B_[1] = *reinterpret_cast<const f32x4*>(sb + brow1 + (co_));               \
    }
#define WG2_MFMA(A_, B_, G_)                                                                                   \
    {                                                                                                              \
        if (VEC && (G_) == 0 && oxs == 0 && lhi == 0) {          /* first chunk of an image row: dx = -1 rows */    \
            const f32x4 u = B_[0], v = B_[1];                                                                      \
            if (dxn0 < 0) { B_[0][0] = refl ? u[1] : 0.f; B_[0][1] = u[0]; B_[0][2] = u[1]; B_[0][3] = u[2]; }     \
            if (dxn1 < 0) { B_[1][0] = refl ? v[1] : 0.f; B_[1][1] = v[0]; B_[1][2] = v[1]; B_[1][3] = v[2]; }     \
        }                                                                                                          \
        if (VEC && (G_) == 1 && oxs == p.OW - BKP && lhi == 1) { /* last chunk of an image row: dx = +1 rows */    \
            const f32x4 u = B_[0], v = B_[1];                                                                      \
            if (dxn0 > 0) { B_[0][0] = u[1]; B_[0][1] = u[2]; B_[0][2] = u[3]; B_[0][3] = refl ? u[2] : 0.f; }     \
            if (dxn1 > 0) { B_[1][0] = v[1]; B_[1][1] = v[2]; B_[1][2] = v[3]; B_[1][3] = refl ? v[2] : 0.f; }     \
        }                                                                                                          \
        if (do_bias) {                                                                                             \
            bsum0 += (A_[0][0] + A_[0][1]) + (A_[0][2] + A_[0][3]);                                                \
            bsum1 += (A_[1][0] + A_[1][1]) + (A_[1][2] + A_[1][3]);                                                \
        }                                                                                                          \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                              \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
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

    if (do_bias) {
        // lanes l and l+32 hold the two pixel-halves of channel row l31
        bsum0 += __shfl_xor(bsum0, 32, 64);
        bsum1 += __shfl_xor(bsum1, 32, 64);
        if (lhi == 0) {
            const int ma = m0 + (wm * 2 + 0) * 32 + l31, mb = ma + 32;
            if (ma < p.K) atomicAdd(p.gb + ma, bsum0);
            if (mb < p.K) atomicAdd(p.gb + mb, bsum1);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int jj = j0 + (wn * 2 + j) * 32 + l31;
        if (jj >= p.J) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < p.K) atomicAdd(p.gw + (size_t)m * p.J + jj, acc[i][j][r]);
            }
    }
}

}  // namespace

// Shapes this kernel takes: wide layers whose gy rows can be read in aligned 16-byte chunks that never straddle two
// images (OH*OW % 4 == 0).  Everything else stays on the VGPR-staged kernel in conv.hip.
bool nemar_wgrad2_eligible(int K, int OH, int OW, const float* gy) {
    return K > 32 && (OH * OW) % 4 == 0 && (reinterpret_cast<uintptr_t>(gy) & 15) == 0;
}

void nemar_wgrad2_launch(const float* x0, int C0, const float* x1, int C1, const float* gy, float* gw, float* gb, int N,
                         int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad, int pad_mode,
                         int target_blocks, bool vec_ok, hipStream_t st) {
    Wgrad2Params p;
    p.src0 = x0; p.src1 = x1; p.C0 = C0; p.C1 = C1; p.Hs = H; p.Ws = W;
    p.gy = gy; p.K = K; p.OH = OH; p.OW = OW;
    p.gw = gw; p.J = (C0 + C1) * R * S; p.gb = gb;
    p.P = N * OH * OW; p.sy = stride; p.sx = stride; p.R = R; p.S = S; p.pad = pad; p.border = pad_mode;
    p.fd_ohw = make_fastdiv(OH * OW); p.fd_ow = make_fastdiv(OW);
    p.fd_rs = make_fastdiv(R * S); p.fd_s = make_fastdiv(S);
    const int mt = nemar_cdiv(K, BM), jt = nemar_cdiv(p.J, BN);
    // Split the pixel reduction so that the grid is ONE full round of resident workgroups (2 per CU x 256 CUs): every
    // workgroup starts and ends together, so a grid of 1.1 or 2.04 rounds pays for 2 or 3.  target_blocks is that
    // capacity; the split count is rounded DOWN to fit it, with >= 8 stages per split.
    int splits = target_blocks / (mt * jt);
    const int max_splits = nemar_cdiv(p.P, BKP * 8);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    p.pix_per_split = nemar_cdiv(nemar_cdiv(p.P, splits), BKP) * BKP;
    splits = nemar_cdiv(p.P, p.pix_per_split);
    // 16-byte source loads: stride 1, image rows that are whole 16-pixel stages, horizontal tap offsets within +-1
    const bool vec = vec_ok && stride == 1 && OW % BKP == 0 && W == OW && pad <= 1 && S - 1 - pad <= 1;
    if (vec) hipLaunchKernelGGL(wgrad2_kernel<true>, dim3(mt, jt, splits), dim3(NT), 0, st, p);
    else hipLaunchKernelGGL(wgrad2_kernel<false>, dim3(mt, jt, splits), dim3(NT), 0, st, p);
}
