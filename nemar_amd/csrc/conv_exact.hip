// K1 + K2 + K4 + K8, the exact-fp32 engine: one tap-table implicit GEMM on the gfx950 matrix cores that serves the
// forward pass AND the data gradient (and therefore ConvTranspose2d) of every layer the 16-bit-pipe routes do not take — reference
// models/networks.py:349-377,418-439,576-597, models/stn/layers.py:85, models/stn/unet_stn.py:80,97, models/stn/affine_stn.py:69-72,79.
// conv.hip holds the operator entry points that build the IgemmParams and choose between this engine and the other routes.
//
//     out[n, m, oy, ox] = act(bias[m] + sum_{t < ntaps} sum_{ch < Cs} A[(t*Cs+ch), m] *
//                              src[n, ch, oy*sy + dy[t], ox*sx + dx[t]])
//   - A is the weight tensor re-laid-out ("packed", see pack_weights_kernel) so that tile loads are dense 16 B/lane
//     runs and one ds_read_b128 feeds four MFMA steps;
//   - src is the logical channel concat of two tensors (never materialised);
//   - out-of-range taps read 0 (zero padding / strided data-gradient) or the mirrored texel (reflect).
// Kernels: igemm_ws2_kernel (wave-specialised, layers big enough for 128x128 tiles), igemm_kernel (generic, every wave
// stages and multiplies; small / odd layers and the ring), igemm_ws_kernel (first-generation wave-specialised: A/B build only).
// GEMM view: M = output channels, N = output pixels, K = taps x source channels.  MFMA tile
// v_mfma_f32_32x32x2_f32 with A = weights (rows = channels) and B = gathered pixels (cols = pixels), so the
// accumulator's col = lane&31 runs along pixels and NCHW stores are coalesced.  f32-in MFMA is an fmaf chain
// (bit-exact fp32, 157 TF peak): no reduced precision anywhere.
#include "conv_exact.h"
#include "pack_plan.h"

namespace nemar_exact {
#ifdef NEMAR_AB
#define NEMAR_EXACT_SWITCH_DEF(type, name, def) type name = def;
NEMAR_EXACT_SWITCHES(NEMAR_EXACT_SWITCH_DEF)
#endif
}  // namespace nemar_exact

using namespace nemar_exact;

namespace {

// ---- weight packing -----------------------------------------------------------------------------------------------
// W[m*wsm + ch*wsc + wofs[t]] with reduction index k = t*Cs + ch goes to
//     A[((k/8)*2 + (k&1)) * Mpad*4 + m*4 + ((k%8)>>1)]            (zero padded to KredPad x Mpad)
// i.e. blocks of 8 reduction rows, split by the MFMA k-slot (k&1), channel-major, with the four MFMA steps of the block
// adjacent: lane (m, kslot) of v_mfma_f32_32x32x2_f32 fetches its A operand for 4 consecutive steps with ONE
// ds_read_b128, the rows of a 16-lane read group fall on 16 different bank slots, and a tile stage is still a dense
// run of 16-byte chunks for global_load_lds.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int M,
                                                           int Mpad, int Cs, int Kred, int KredPad, int wsm, int wsc,
                                                           int zero_tail, TapTable taps) {
    __shared__ int s_wofs[MAX_TAPS];
    for (int i = threadIdx.x; i < MAX_TAPS; i += blockDim.x) s_wofs[i] = i < taps.n ? taps.wofs[i] : 0;
    __syncthreads();
    const int core = KredPad * Mpad, total = core + zero_tail;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int blk = idx / (Mpad * 4), within = idx - blk * (Mpad * 4);
        const int m = within >> 2, s = within & 3;
        const int kk = 8 * (blk >> 1) + 2 * s + (blk & 1);
        float v = 0.f;
        if (idx < core && m < M && kk < Kred) {
            const int t = kk / Cs, ch = kk - t * Cs;
            v = w[(size_t)m * wsm + (size_t)ch * wsc + s_wofs[t]];
        }
        wp[idx] = v;
    }
}

// pixel index inside one image -> (oy, ox).  Ring mode enumerates, in padded coordinates of a (H+2p) x (W+2p) plane:
// top band (p rows), bottom band (p rows), then for each image row the p left and p right columns.
__device__ __forceinline__ void decode_ring(unsigned rp, unsigned H, unsigned W, unsigned rem, unsigned& oy, unsigned& ox) {
    const unsigned Wp = W + 2 * rp, band = rp * Wp;
    if (rem < 2 * band) {
        const unsigned q = rem < band ? rem : rem - band;
        const unsigned r = q / Wp;
        oy = rem < band ? r : H + rp + r;
        ox = q - r * Wp;
    } else {
        const unsigned q = rem - 2 * band;
        const unsigned r = q / (2 * rp), e = q - r * (2 * rp);
        oy = rp + r;
        ox = e < rp ? e : W + e;
    }
}
__device__ __forceinline__ void decode_pixel(const IgemmParams& p, unsigned rem, unsigned& oy, unsigned& ox) {
    if (p.ring_p == 0) {
        oy = fd_div(rem, p.fd_ow);
        ox = rem - oy * (unsigned)p.OW;
        return;
    }
    decode_ring((unsigned)p.ring_p, (unsigned)p.ring_H, (unsigned)p.ring_W, rem, oy, ox);
}

// WM x WN waves, each TM x TN MFMA tiles of 32x32 (workgroup = WM*WN*64 threads).  FAST: Cs % BK == 0 && C0 % BK == 0,
// so a whole BK-deep stage shares one tap and one source tensor (address math once per stage instead of per element).
//
// Staging is direct global -> LDS (global_load_lds_*): no VGPR round trip, no ds_write pass, and no register hazards for
// the compiler to guard with early waits.  The stage for step k+1 is issued before the MFMAs of step k and drained
// (s_waitcnt vmcnt(0) + barrier) after them.  A tile: packed weights, 16 B per lane, dense LDS rows (one wave
// instruction = 1 KiB).  B tile: one 4-byte gather per lane, 64 consecutive pixels of one reduction row per wave
// instruction; taps that fall outside a zero-padded source (and tile tails) are pointed at a zero page.
template <int WM, int WN, int TM, int TN, bool FAST, int NBUF = 2>
__global__ __launch_bounds__(WM * WN * 64) void igemm_kernel(IgemmParams p) {
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDB = BN + 4;
    constexpr int SEGS = BN / 64;               // 64-pixel segments per B row
    constexpr int RGRP = NW / SEGS;             // waves working on the same segment (each takes every RGRP-th row)
    constexpr int BROWS = BK / RGRP;            // B rows per wave per stage
    constexpr int A_INSTR = BK * BM / 256;      // 1 KiB wave-instructions per A stage
    constexpr int A_PER = (A_INSTR + NW - 1) / NW;
    static_assert(BN % 64 == 0 && NW % SEGS == 0 && BK % RGRP == 0, "tile");

    // ONE __shared__ object: with several, hipcc guards every ds_read of a stage with s_waitcnt vmcnt(0) while a
    // global_load_lds is in flight (it cannot tell which object the DMA writes), which would serialise the pipeline
    constexpr int A_FLOATS = BK * BM, B_FLOATS = BK * LDB;
    constexpr int LOADS = A_PER + BROWS;          // global->LDS instructions per wave per stage (uniform when NBUF > 2)
    static_assert(NBUF == 2 || (FAST && A_INSTR % NW == 0), "counted waits need the same number of loads in every wave");
    static_assert(NBUF >= 2 && NBUF <= 4 && 2 * LOADS < 64, "ring depth");
    __shared__ __attribute__((aligned(16))) float smem[NBUF * A_FLOATS + NBUF * B_FLOATS + MAX_TAPS];
    float* const As0 = smem;
    float* const Bs0 = smem + NBUF * A_FLOATS;
    int* const s_tap = reinterpret_cast<int*>(smem + NBUF * A_FLOATS + NBUF * B_FLOATS);

    const int tid = threadIdx.x;
    const int wid = tid >> 6, lane = tid & 63;
    for (int i = tid; i < MAX_TAPS; i += NT)
        s_tap[i] = i < p.taps.n ? (((int)p.taps.dy[i] << 16) | ((int)p.taps.dx[i] & 0xffff)) : 0;

    const int m0 = blockIdx.y * BM;
    const int p0 = blockIdx.x * BN;
    const int Cs = p.C0 + p.C1;
    const int HW = p.Hs * p.Ws;

    // ---- this lane's pixel column of the B tile (fixed for the whole reduction) ----------------------------------
    const int seg = wid % SEGS, rgrp = wid / SEGS;
    const int pc = seg * 64 + lane;
    const int pix = p0 + pc;
    const bool pvalid = pix < p.P;
    int by = 0, bx = 0;
    const float* s0n = p.src0;
    const float* s1n = p.src1;
    {
        const unsigned upix = pvalid ? (unsigned)pix : 0u;
        const unsigned n = fd_div(upix, p.fd_ohw);
        const unsigned rem = upix - n * (unsigned)(p.OH * p.OW);
        unsigned oy, ox;
        decode_pixel(p, rem, oy, ox);
        by = (int)oy * p.sy;
        bx = (int)ox * p.sx;
        s0n = p.src0 + (size_t)n * p.C0 * HW;
        if (p.C1) s1n = p.src1 + (size_t)n * p.C1 * HW;
    }
    // ---- this lane's 16-byte slots of the A tile -----------------------------------------------------------------------
    const float* wsrc[A_PER];
    int a_lds[A_PER];
#pragma unroll
    for (int q = 0; q < A_PER; ++q) {
        const int e = (wid + q * NW) * 256 + lane * 4;      // float index inside the dense [4 blocks][BM][4] stage
        const int blk = e / (BM * 4), m = (e - blk * (BM * 4)) >> 2;
        wsrc[q] = p.wp + ((size_t)blk * p.Mpad + m0 + m) * 4;
        a_lds[q] = (wid + q * NW) * 256;                    // wave-uniform LDS base (floats)
    }
    __syncthreads();  // s_tap visible

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int f_tap = -1, f_off = 0;      // FAST staging: tap whose border arithmetic is cached, this lane's y*Ws+x for it
    bool f_inb = false;
#define IGEMM_ISSUE_STAGE(k0_, buf_)                                                                                 \
    {                                                                                                                \
        const int k0 = (k0_);                                                                                        \
        _Pragma("unroll") for (int q = 0; q < A_PER; ++q) {                                                          \
            if (A_INSTR % NW == 0 || wid + q * NW < A_INSTR)                                                         \
                glds_b128(wsrc[q] + (size_t)k0 * p.Mpad, As0 + (buf_) * A_FLOATS + a_lds[q]);                                       \
        }                                                                                                            \
        if (FAST) {                                                                                                  \
            const int t = (int)fd_div((unsigned)k0, p.fd_cs);                                                        \
            const int ch0 = k0 - t * Cs;                                                                             \
            if (t != f_tap) {          /* border arithmetic + tap-table read once per tap, not per stage */          \
                f_tap = t;                                                                                           \
                const int tp = s_tap[t];                                                                             \
                int y = by + (tp >> 16), x = bx + (int)(short)(tp & 0xffff);                                         \
                f_inb = pvalid;                                                                                      \
                if (p.border == BORDER_REFLECT) {                                                                    \
                    y = reflect(y, p.Hs);                                                                            \
                    x = reflect(x, p.Ws);                                                                            \
                } else {                                                                                             \
                    f_inb = f_inb && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;                   \
                }                                                                                                    \
                f_off = f_inb ? y * p.Ws + x : 0;                                                                    \
            }                                                                                                        \
            const bool inb = f_inb && k0 < p.Kred;                                                                   \
            const float* base = (ch0 < p.C0) ? s0n + (size_t)ch0 * HW : s1n + (size_t)(ch0 - p.C0) * HW;           \
            base += (size_t)rgrp * HW + f_off;                                                                       \
            _Pragma("unroll") for (int i = 0; i < BROWS; ++i)                                                        \
                glds_b32(inb ? base + (size_t)(RGRP * i) * HW : p.zero,                                              \
                         Bs0 + (buf_) * B_FLOATS + (rgrp + RGRP * i) * LDB + seg * 64);       \
        } else {                                                                                                     \
            _Pragma("unroll") for (int i = 0; i < BROWS; ++i) {                                                      \
                const int kk = k0 + rgrp + RGRP * i;                                                                 \
                const float* src = p.zero;                                                                           \
                if (pvalid && kk < p.Kred) {                                                                         \
                    const unsigned t = fd_div((unsigned)kk, p.fd_cs);                                                \
                    const int ch = kk - (int)t * Cs;                                                                 \
                    const int tp = s_tap[t];                                                                         \
                    int y = by + (tp >> 16), x = bx + (int)(short)(tp & 0xffff);                                     \
                    bool inb = true;                                                                                 \
                    if (p.border == BORDER_REFLECT) {                                                                \
                        y = reflect(y, p.Hs);                                                                        \
                        x = reflect(x, p.Ws);                                                                        \
                    } else {                                                                                         \
                        inb = (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;                          \
                    }                                                                                                \
                    if (inb) src = ((ch < p.C0) ? s0n + (size_t)ch * HW : s1n + (size_t)(ch - p.C0) * HW) + y * p.Ws + x; \
                }                                                                                                    \
                glds_b32(src, Bs0 + (buf_) * B_FLOATS + (rgrp + RGRP * i) * LDB + seg * 64);                                                 \
            }                                                                                                        \
        }                                                                                                            \
    }

    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ring mode splits the reduction over blockIdx.z (the atomic epilogue sums the partial results)
    const int nk_all = (p.Kred + BK - 1) / BK;
    const int nk_per = (nk_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int ks0 = blockIdx.z * nk_per, nk = min(nk_all, ks0 + nk_per);
    if (ks0 >= nk) return;
    // Ring mode: most (tile, tap) pairs have no source texel in range (a top-band ring pixel is reached by the r = 0 taps
    // only), i.e. an all-zero B tile.  The workgroup collects the set of taps that reach ANY of its pixels and skips the
    // stages of the others (~2/3 of them for a 3x3 layer).
    unsigned long long amask = ~0ull;
    if (p.ring_p) {
        unsigned lo = 0, hi = 0;
        for (int t = 0; t < p.taps.n; ++t) {
            const int tp = s_tap[t];
            const int y = by + (tp >> 16), x = bx + (int)(short)(tp & 0xffff);
            const bool hit = pvalid && (p.border == BORDER_REFLECT ||
                                        ((unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws));
            if (__ballot(hit) != 0ull) { if (t < 32) lo |= 1u << t; else hi |= 1u << (t - 32); }
        }
        unsigned* const s_mask = reinterpret_cast<unsigned*>(s_tap + MAX_TAPS - 2);   // taps.n <= 49: last slots are free
        if (tid == 0) { s_mask[0] = 0u; s_mask[1] = 0u; }
        __syncthreads();
        if (lane == 0) { atomicOr(&s_mask[0], lo); atomicOr(&s_mask[1], hi); }
        __syncthreads();
        amask = ((unsigned long long)s_mask[1] << 32) | s_mask[0];
    }
#define IGEMM_STAGE_ACTIVE(ks_)                                                                                   \
    (amask == ~0ull ||                                                                                                \
     ((amask >> fd_div((unsigned)((ks_) * BK), p.fd_cs)) &                                                            \
      ((2ull << (fd_div((unsigned)min((ks_) * BK + BK - 1, p.Kred - 1), p.fd_cs) - fd_div((unsigned)((ks_) * BK), p.fd_cs))) - 1ull)) != 0ull)
#define IGEMM_NEXT_ACTIVE(var_)  while ((var_) < nk && !IGEMM_STAGE_ACTIVE(var_)) ++(var_)
    // NBUF-deep LDS ring: up to NBUF-2 further stages stay in flight while one is consumed (counted s_waitcnt vmcnt);
    // NBUF = 2 is plain double buffering.  One barrier per stage: it publishes the stage and retires the buffer consumed
    // before it.  `slot` counts consumed stages, `n_iss` issued ones (inactive stages take no slot).
    int iss = ks0;
    IGEMM_NEXT_ACTIVE(iss);
    int first = iss, n_iss = 0;
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d)
        if (iss < nk) {
            IGEMM_ISSUE_STAGE(iss * BK, n_iss % NBUF);
            ++n_iss;
            ++iss;
            IGEMM_NEXT_ACTIVE(iss);
        }
    int slot = 0;
    for (int ks = first; ks < nk; ++slot) {
        const int buf = slot % NBUF;
        if (NBUF == 2) {
            wait_vmem();
        } else {
            const int ahead = n_iss - slot - 1;       // stages issued beyond this one
            if (ahead >= 2 && NBUF >= 4) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * LOADS) & 15) | (((2 * LOADS) >> 4) << 14));
            else if (ahead >= 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (LOADS & 15) | ((LOADS >> 4) << 14));
            else wait_vmem();
        }
        if (!(p.dbg & 4)) __syncthreads();
        if (iss < nk && !(p.dbg & 1)) {
            IGEMM_ISSUE_STAGE(iss * BK, n_iss % NBUF);
            ++n_iss;
            ++iss;
            IGEMM_NEXT_ACTIVE(iss);
        }
        ++ks;
        IGEMM_NEXT_ACTIVE(ks);
        if (!(p.dbg & 2)) {
            // all MFMA operands of the stage into registers first (one LDS round trip per stage), then the MFMAs
            // back to back: reading fragments just-in-time makes hipcc reuse the operand registers, and the
            // write-after-read wait on them puts an LDS latency between every pair of MFMAs
            f32x4 a[BK / 8][TM];
            float b[BK / 2][TN];
#pragma unroll
            for (int kg = 0; kg < BK / 8; ++kg)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[kg][i] = *reinterpret_cast<const f32x4*>(
                        As0 + buf * A_FLOATS + ((kg * 2 + lhi) * BM + (wm * TM + i) * 32 + l31) * 4);
#pragma unroll
            for (int k2 = 0; k2 < BK / 2; ++k2) {
                const int kr = 2 * k2 + lhi;
#pragma unroll
                for (int j = 0; j < TN; ++j) b[k2][j] = Bs0[buf * B_FLOATS + kr * LDB + (wn * TN + j) * 32 + l31];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the fragment loads above, in their own registers
#pragma unroll
            for (int k2 = 0; k2 < BK / 2; ++k2)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2 >> 2][i][k2 & 3], b[k2][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of `buf` are done before the next barrier
        }
    }
#undef IGEMM_ISSUE_STAGE
#undef IGEMM_STAGE_ACTIVE
#undef IGEMM_NEXT_ACTIVE

    // ---- epilogue: bias + activation, NCHW store (lane&31 runs along pixels => coalesced rows) -----------------
    const size_t oplane = (size_t)p.OHf * p.OWf;
    const int M1 = p.M - p.M0;
    // split reduction (tiny, deep problems: a 2x2-pixel 128->128 layer is 72 serial stages in one or two workgroups): this
    // split's partial tile goes to its own slab (no bias / activation in this mode: the launcher guarantees it)
    float* const d0 = gridDim.z > 1 ? p.part + (size_t)blockIdx.z * (size_t)p.part_stride : p.dst0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int opix = p0 + (wn * TN + j) * 32 + l31;
        if (opix >= p.P) continue;
        const unsigned n = fd_div((unsigned)opix, p.fd_ohw);
        const unsigned rem = (unsigned)opix - n * (unsigned)(p.OH * p.OW);
        unsigned oy, ox;
        decode_pixel(p, rem, oy, ox);
        if (p.ring_p) {
            // ring results go to a compact [n][m][ring] scratch (lanes = consecutive ring positions: coalesced stores);
            // ring_gather_kernel adds them to the texels the reflect padding mirrored
            const size_t rlen = (size_t)p.OH * p.OW;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (m < p.M) d0[((size_t)n * p.M + m) * rlen + rem] = acc[i][j][r];
                }
            continue;
        }
        const size_t sp = (size_t)((int)oy * p.osy + p.ooy) * p.OWf + ((int)ox * p.osx + p.oox);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < p.M) {
                    float v = acc[i][j][r];
                    if (p.bias && blockIdx.z == 0) v += p.bias[m];       // split reductions: slab 0 carries the bias
                    v = apply_act(v, p.act, p.slope);
                    if (m < p.M0) {
                        if (d0) d0[((size_t)n * p.M0 + m) * oplane + sp] = v;
                    } else {
                        p.dst1[((size_t)n * M1 + (m - p.M0)) * oplane + sp] = v;
                    }
                }
            }
        }
    }
}

#ifdef NEMAR_AB
// ---- wave-specialised 128x128 tile (FAST shapes): 8 MFMA waves + 2 loader waves ---------------------------------------
// Measured on the 4-/8-wave kernels above (tools/timeline_conv.py, ablation switches): the MFMA phase of a stage and the
// global->LDS staging of the next one do not overlap — every wave first spends ~1-3k cycles getting its
// global_load_lds accepted by the memory pipeline and only then starts its MFMAs, and all waves of a workgroup do so
// together after each barrier.  Here the MFMA waves issue no vector-memory instructions at all: two extra waves own
// the staging (each: one 64-pixel segment of every B row + half of the A rows), run TWO stages ahead through a 3-deep
// LDS ring, and keep one stage in flight across the barrier with a counted s_waitcnt vmcnt(N).
constexpr int WS_NC = 8, WS_NL = 2, WS_NT = (WS_NC + WS_NL) * 64, WS_NBUF = 3;
__global__ __launch_bounds__(WS_NT) void igemm_ws_kernel(IgemmParams p) {
    constexpr int BM = 128, BN = 128, LDB = BN + 4, TM = 2, WN = 4;
    constexpr int A_FLOATS = BK * BM, B_FLOATS = BK * LDB;
    constexpr int A_PER_LOADER = BK * BM / 256 / WS_NL;      // 1 KiB wave-instructions of A per loader per stage (4)
    constexpr int LOADS_PER_STAGE = A_PER_LOADER + BK;       // + one 64-pixel segment of each of the BK rows (16)
    __shared__ __attribute__((aligned(16))) float smem[WS_NBUF * (A_FLOATS + B_FLOATS)];
    float* const As0 = smem;
    float* const Bs0 = smem + WS_NBUF * A_FLOATS;

    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * BM, p0 = blockIdx.x * BN;
    const int Cs = p.C0 + p.C1, HW = p.Hs * p.Ws;
    const int nk = (p.Kred + BK - 1) / BK;

    if (wid >= WS_NC) {
        // ================================ loader waves ================================
        const int seg = wid - WS_NC;                       // 64-pixel segment of the B tile owned by this wave
        const int pix = p0 + seg * 64 + lane;
        const bool pvalid = pix < p.P;
        const unsigned upix = pvalid ? (unsigned)pix : 0u;
        const unsigned n = fd_div(upix, p.fd_ohw);
        const unsigned rem = upix - n * (unsigned)(p.OH * p.OW);
        const unsigned oy = fd_div(rem, p.fd_ow);
        const int by = (int)oy * p.sy, bx = (int)(rem - oy * (unsigned)p.OW) * p.sx;
        const float* s0n = p.src0 + (size_t)n * p.C0 * HW;
        const float* s1n = p.C1 ? p.src1 + (size_t)n * p.C1 * HW : p.src0;
        const float* wsrc[A_PER_LOADER];
        int a_lds[A_PER_LOADER];
#pragma unroll
        for (int q = 0; q < A_PER_LOADER; ++q) {
            const int inst = seg * A_PER_LOADER + q;
            const int e = inst * 256 + lane * 4;
            const int blk = e / (BM * 4), m = (e - blk * (BM * 4)) >> 2;
            wsrc[q] = p.wp + ((size_t)blk * p.Mpad + m0 + m) * 4;
            a_lds[q] = inst * 256;
        }
        auto issue = [&](int ks) {
            const int k0 = ks * BK, buf = ks % WS_NBUF;
#pragma unroll
            for (int q = 0; q < A_PER_LOADER; ++q) glds_b128(wsrc[q] + (size_t)k0 * p.Mpad, As0 + buf * A_FLOATS + a_lds[q]);
            const unsigned t = fd_div((unsigned)k0, p.fd_cs);
            const int ch0 = k0 - (int)t * Cs;
            int y = by + p.taps.dy[t], x = bx + p.taps.dx[t];
            bool inb = pvalid;
            if (p.border == BORDER_REFLECT) {
                y = reflect(y, p.Hs);
                x = reflect(x, p.Ws);
            } else {
                inb = inb && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
            }
            const float* base = (ch0 < p.C0) ? s0n + (size_t)ch0 * HW : s1n + (size_t)(ch0 - p.C0) * HW;
            base += inb ? y * p.Ws + x : 0;
#pragma unroll
            for (int r = 0; r < BK; ++r)
                glds_b32(inb ? base + (size_t)r * HW : p.zero, Bs0 + buf * B_FLOATS + r * LDB + seg * 64);
        };
        issue(0);
        if (nk > 1) issue(1);
        // stage 0 must have landed before the first barrier; stage 1 may stay in flight
        if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (LOADS_PER_STAGE & 15) | ((LOADS_PER_STAGE >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        for (int ks = 0; ks < nk; ++ks) {
            // buffer (ks+2)%3 == (ks-1)%3 was released by the barrier that ended stage ks-1
            if (ks + 2 < nk) {
                issue(ks + 2);
                __builtin_amdgcn_s_waitcnt(0x0F70 | (LOADS_PER_STAGE & 15) | ((LOADS_PER_STAGE >> 4) << 14));  // ks+1 landed
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ================================ MFMA waves ================================
    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __builtin_amdgcn_s_barrier();      // stage 0 is in LDS
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks % WS_NBUF;
        f32x4 a[BK / 8][TM];
        float b[BK / 2];
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[kg][i] = *reinterpret_cast<const f32x4*>(
                    As0 + buf * A_FLOATS + ((kg * 2 + lhi) * BM + (wm * TM + i) * 32 + l31) * 4);
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2) b[k2] = Bs0[buf * B_FLOATS + (2 * k2 + lhi) * LDB + wn * 32 + l31];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2 >> 2][i][k2 & 3], b[k2], acc[i], 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): all LDS reads of this buffer returned before releasing it
        __builtin_amdgcn_s_barrier();
    }

    const size_t oplane = (size_t)p.OHf * p.OWf;
    const int M1 = p.M - p.M0;
    const int opix = p0 + wn * 32 + l31;
    if (opix >= p.P) return;
    const unsigned n = fd_div((unsigned)opix, p.fd_ohw);
    const unsigned rem = (unsigned)opix - n * (unsigned)(p.OH * p.OW);
    const unsigned oy = fd_div(rem, p.fd_ow);
    const unsigned ox = rem - oy * (unsigned)p.OW;
    const size_t sp = (size_t)((int)oy * p.osy + p.ooy) * p.OWf + ((int)ox * p.osx + p.oox);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < p.M) {
                float v = acc[i][r];
                if (p.bias) v += p.bias[m];
                v = apply_act(v, p.act, p.slope);
                if (m < p.M0) {
                    if (p.dst0) p.dst0[((size_t)n * p.M0 + m) * oplane + sp] = v;
                } else {
                    p.dst1[((size_t)n * M1 + (m - p.M0)) * oplane + sp] = v;
                }
            }
        }
    }
}
#endif  // NEMAR_AB

// ---- wave-specialised 128x128 tile, second generation: 4 MFMA waves of 32 channels x 128 pixels + 2 loader waves -------
// Same staging as igemm_ws_kernel (the loader code is shared in spirit: A dense 16 B/lane, B one gathered texel per lane),
// different consumer:
//   * B fragments are read with ds_read_b128: lane (J, kslot) takes pixels 4J..4J+3 of reduction row 2s+kslot and feeds
//     component t to MFMA tile t, i.e. the four 32-pixel MFMA tiles of a wave interleave the 128 pixels (tile t owns
//     pixels == t mod 4).  One 16-byte read per MFMA step serves four MFMAs; a 16-lane read group covers one 256-byte run
//     of a row, so it is conflict-free for any row pitch; and in the epilogue every lane owns 4 consecutive pixels of
//     each of its channel rows: 16-byte stores.
//   * A fragments: one ds_read_b128 per 4 MFMA steps (k-interleaved packing, see pack_weights_kernel).
//   * fragments of the next 8 reduction rows are prefetched into a second register set before the 16 MFMAs of the
//     current 8 rows are issued; the stage barrier sits between the two halves of a stage, when the reads of the current
//     buffer have all been issued, so neither LDS latency nor the barrier idles the matrix pipe.
// MT = MFMA waves = 32-channel row tiles (BM = 32*MT: 128 / 64 / 32 output channels per workgroup).
// VEC = the B tile is staged with 16-byte global->LDS loads: lane = 4 consecutive output pixels of one reduction row
// (stride-1 layers with OW % 4 == 0 and |dx| <= 1: the four source texels are consecutive in memory, at an address that
// is only 4-byte aligned — global_load_lds_dwordx4 takes that).  A group whose shifted window would stick out of the
// source row by one texel is loaded from the clamped address instead, and the MFMA wave that consumes it rotates the
// three good texels into place and inserts the mirrored texel (reflect) or 0 (zero padding) — a handful of
// v_cndmask per stage on the two border lanes of a row.  8 wave-instructions per stage instead of 32.
// SPB = stages per barrier.  1: 3-deep stage ring, one workgroup barrier per 16 reduction rows.  2: 4 stage slots used as
// two 32-row halves — the loaders fill one half while the MFMA waves consume the other, one barrier per 32 rows.
// ADIR = the A operand (packed weights) bypasses LDS.  A wave's A fragments are private to it (wave w owns channel rows
// 32w..32w+31 of the tile; only the pixel tile B is shared by the four MFMA waves), and the packed layout IS the fragment
// layout — lane (m, kslot) needs one 16-byte word per 8 reduction rows — so each MFMA wave fetches its own two words per stage
// straight from global memory (L2-resident: the whole packed tensor is 2.4 MB) into registers, one stage ahead, in the
// issue shadow of its MFMAs.  The loader waves then move HALF the bytes per stage (the B tile only): they are the critical path
// of this kernel, and global->LDS DMA throughput per CU is what bounds them.
// RING = LDS ring depth of the one-stage-per-barrier variant: the loaders run RING - 1 stages ahead of the MFMA waves (3: two
// stage times = ~1.7 us for a global->LDS copy to land; 4 / 5 trade LDS (16 KiB per stage) for more latency tolerance).
template <int MT, bool VEC, int SPB = 1, int NL = 2, bool ADIR = false, int RING = 3>   // NL = loader waves (4 only with VEC)
__global__ __launch_bounds__((MT + NL) * 64) void igemm_ws2_kernel(IgemmParams p) {
    static_assert(NL == 2 || NL == 4, "loader split");
    static_assert(!ADIR || SPB == 1, "direct A operands: one-stage barrier variant only");
    static_assert(RING >= 3 && RING <= 5, "ring depth");
    constexpr int W2_NBUF = SPB == 1 ? RING : 4;
    constexpr int BM = 32 * MT, BN = 128, LDB = VEC ? BN : BN + 4;
    constexpr int A_FLOATS = ADIR ? 0 : BK * BM, B_FLOATS = BK * LDB;
    constexpr int A_PER_LOADER = ADIR ? 0 : BK * BM / 256 / NL;   // 1 KiB wave-instructions of A per loader per stage
    constexpr int B_PER_LOADER = VEC ? BK / 2 / NL : BK / (NL / 2);   // VEC: 2 rows x 128 px per instruction; else 1 row x 64 px
    constexpr int LOADS_PER_STAGE = A_PER_LOADER + B_PER_LOADER;
    static_assert(BK == 16, "a stage is two 8-row fragment groups");
    __shared__ __attribute__((aligned(16))) float smem[W2_NBUF * (A_FLOATS + B_FLOATS)];
    float* const As0 = smem;
    float* const Bs0 = smem + W2_NBUF * A_FLOATS;

    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    // Workgroup -> tile.  Consecutive workgroup ids land on consecutive XCDs (id % 8), each with its own L2; neighbouring
    // pixel tiles share half of their source rows (a 3x3 tile of two image rows reads four), so with the identity mapping
    // every XCD fetches the halo rows its neighbours already hold.  p.xcd != 0: XCD k takes a CONTIGUOUS run of pixel tiles
    // (and all channel tiles of each), so halo rows and the second channel tile's B tile are L2 hits.
    int bx = blockIdx.x, by_ = blockIdx.y;
    if (p.xcd) {
        const int gxx = gridDim.x, gyy = gridDim.y;
        const int b = bx + gxx * by_;
        const int k = b & 7, j = b >> 3;
        bx = k * (gxx >> 3) + j / gyy;
        by_ = j - (j / gyy) * gyy;
    }
    const int m0 = by_ * BM, p0 = bx * BN;
    const int Cs = p.C0 + p.C1, HW = p.Hs * p.Ws;
    // grid.z > 1: the reduction is split over workgroups (few, deep tiles — D's 256->512 k4 data gradient is 128 tiles x 512
    // stages); partial results meet in the zero-filled destination through atomics (no bias / activation in that mode)
    const int nk_all = (p.Kred + BK - 1) / BK;
    const int nk_per = (nk_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int ks0 = blockIdx.z * nk_per;
    const int nk = min(nk_all, ks0 + nk_per) - ks0;          // stages of this workgroup (indices below are relative)
    if (nk <= 0) return;
    const int k_start = ks0 * BK;
    // experiment (nemar_tune 2=256): workgroups 256 apart in dispatch order share a CU; start every other one late
    if ((p.dbg & 256) && (((blockIdx.x + gridDim.x * blockIdx.y) >> 8) & 1)) __builtin_amdgcn_s_sleep(20);
    const int tap_start = (int)fd_div((unsigned)k_start, p.fd_cs), ch_start = k_start - tap_start * Cs;

    if (wid >= MT) {
        // ================================ loader waves ================================
        const int ldr = wid - MT;
        // VEC: lane = (pixel group lane&31, row parity lane>>5), this loader's rows are ldr*8 .. ldr*8+7
        // else: lane = pixel of 64-pixel segment `ldr`, rows 0..15
        const int seg = ldr & 1, row0 = (ldr >> 1) * B_PER_LOADER;     // non-VEC: 64-pixel segment and first row of this loader
        const int pix = VEC ? p0 + 4 * (lane & 31) : p0 + seg * 64 + lane;
        const bool pvalid = pix < p.P;
        const unsigned upix = pvalid ? (unsigned)pix : 0u;
        const unsigned n = fd_div(upix, p.fd_ohw);
        const unsigned rem = upix - n * (unsigned)(p.OH * p.OW);
        const unsigned oy = fd_div(rem, p.fd_ow);
        const int by = (int)oy * p.sy, bx = (int)(rem - oy * (unsigned)p.OW) * p.sx;
        const int rofs = VEC ? ldr * (BK / NL) + (lane >> 5) : row0;  // first reduction row (channel offset in the stage) of this lane
        const float* s0n = p.src0 + ((size_t)n * p.C0 + rofs) * HW;
        const float* s1n = p.C1 ? p.src1 + ((size_t)n * p.C1 + rofs) * HW : s0n;
        const float* wsrc[A_PER_LOADER > 0 ? A_PER_LOADER : 1];
        int a_lds[A_PER_LOADER > 0 ? A_PER_LOADER : 1];
#pragma unroll
        for (int q = 0; q < A_PER_LOADER; ++q) {
            const int inst = ldr * A_PER_LOADER + q;
            const int e = inst * 256 + lane * 4;
            const int blk = e / (BM * 4), m = (e - blk * (BM * 4)) >> 2;
            wsrc[q] = p.wp + ((size_t)blk * p.Mpad + m0 + m) * 4;
            a_lds[q] = inst * 256;
        }
        // The loader is a state machine over consecutive stages: everything that repeats is a pointer bump.  (Per-stage
        // tap-table lookups in the kernel arguments, divisions and border arithmetic between the loads made the loader
        // the last wave at the stage barrier: two extra scalar branches in this loop cost the whole kernel 7 %.)
        int a_buf = 0;                                        // LDS ring slot of the next stage
        int tap_t = tap_start, ch0 = ch_start;                // tap / channel offset of the next stage
        // Per tap, every lane resolves ONCE where its texels of channel 0 live (`lbase`) and how far apart consecutive
        // channels are (`lstride`); per stage only `lp` moves.  Lanes whose tap falls outside a zero-padded source read the
        // zero page with stride 0.  Reflect data gradient (p.rf, see reflect_aux_kernel): lanes on a border row / in a
        // border column group read the pre-folded side buffers instead of gy, with those buffers' channel pitch.
        const float* lp = p.zero;                             // this lane's address for reduction row `rofs` of the next stage
        size_t lstride = 0;                                   // floats between consecutive channels at that address
        int sp_off = 0;                                       // y * Ws + x of this lane for the current tap (in-range lanes)
        bool inb = false;
        int ndyx = p.taps.dyx[__builtin_amdgcn_readfirstlane(tap_start)];   // offsets of the tap about to start
#pragma unroll
        for (int q = 0; q < A_PER_LOADER; ++q) wsrc[q] += (size_t)k_start * p.Mpad;
#define WS2_ENTER_TAP(ch_)                                                                                           \
        {                                                                                                            \
            const int dy = ndyx >> 16, dx = (int)(short)(ndyx & 0xffff);                                             \
            int y = by + dy, x = bx + dx;                                                                            \
            inb = pvalid;                                                                                            \
            if (p.border == BORDER_REFLECT) y = reflect(y, p.Hs);                                                    \
            else inb = inb && (unsigned)y < (unsigned)p.Hs;                                                          \
            if (VEC) x = (p.dbg & 2048) ? (min(max(x, 0), p.Ws - 4) & ~3) : min(max(x, 0), p.Ws - 4);  /* 2048: aligned-B ablation */ \
            else if (p.border == BORDER_REFLECT) x = reflect(x, p.Ws);                                               \
            else inb = inb && (unsigned)x < (unsigned)p.Ws;                                                          \
            sp_off = inb ? y * p.Ws + x : 0;                                                                         \
            const float* lbase = inb ? s0n + sp_off : p.zero;                                                        \
            lstride = inb ? (size_t)HW : (size_t)0;                                                                  \
            if (VEC && p.rf && pvalid) {                                                                             \
                const bool top = dy == 1 && by == 1, bot = dy == -1 && by == p.Hs - 2;                               \
                if (top || bot) {            /* whole row pre-folded: [top|bot][dx+1][n][k][Ws] */                   \
                    lstride = (size_t)p.Ws;                                                                          \
                    lbase = p.rf_row + ((((size_t)(bot ? 3 : 0) + (dx + 1)) * p.N + n) * p.C0 + rofs) * p.Ws + x;     \
                } else if (inb && ((dx == 1 && bx == 0) || (dx == -1 && bx == p.Ws - 4))) {                         \
                    /* border column group pre-folded: [left|right][n][k][Hs][4] */                                  \
                    lstride = (size_t)p.Hs * 4;                                                                      \
                    lbase = p.rf_col + ((((size_t)(dx == 1 ? 0 : 1) * p.N + n) * p.C0 + rofs) * p.Hs + y) * 4;       \
                }                                                                                                    \
            }                                                                                                        \
            lp = lbase + (size_t)(ch_) * lstride;                                                                    \
            ndyx = p.taps.dyx[__builtin_amdgcn_readfirstlane(min(tap_t + 1, p.taps.n - 1))];                         \
        }
        // the first tap may be entered mid-way (split reductions): ch_start channels in, possibly already in source 1
        WS2_ENTER_TAP(ch0 < p.C0 ? ch0 : 0);
        if (ch0 >= p.C0 && p.C1) lp = inb ? s1n + sp_off + (size_t)(ch0 - p.C0) * HW : p.zero;
#define WS2_ISSUE_NEXT()                                                                                             \
        {                                                                                                            \
            /* ablations (nemar_tune key 2): 1024 = no A loads, 512 = no B loads (results are garbage; timing only) */       \
            if (!(p.dbg & 1024)) {                                                                                   \
                _Pragma("unroll") for (int q = 0; q < A_PER_LOADER; ++q) {                                           \
                    glds_b128(wsrc[q], As0 + a_buf * A_FLOATS + a_lds[q]);                                           \
                    wsrc[q] += (size_t)BK * p.Mpad;                                                                  \
                }                                                                                                    \
            }                                                                                                        \
            if (p.dbg & 512) {                                                                                       \
            } else if (VEC) {                                                                                        \
                _Pragma("unroll") for (int i = 0; i < B_PER_LOADER; ++i)                                             \
                    glds_b128(lp + (size_t)(2 * i) * lstride, Bs0 + a_buf * B_FLOATS + (ldr * B_PER_LOADER + i) * 256); \
            } else {                                                                                                 \
                _Pragma("unroll") for (int r = 0; r < B_PER_LOADER; ++r)                                             \
                    glds_b32(lp + (size_t)r * lstride, Bs0 + a_buf * B_FLOATS + (row0 + r) * LDB + seg * 64);        \
            }                                                                                                        \
            a_buf = a_buf + 1 == W2_NBUF ? 0 : a_buf + 1;                                                            \
            ch0 += BK;                                                                                               \
            lp += (size_t)BK * lstride;                                                                              \
            if (ch0 == p.C0 && p.C1) lp = inb ? s1n + sp_off : p.zero;                                               \
            if (ch0 == Cs) {                                                                                         \
                ch0 = 0;                                                                                             \
                ++tap_t;                                                                                             \
                WS2_ENTER_TAP(0);                                                                                    \
            }                                                                                                        \
        }
#define WS2_ISSUE(ks_) WS2_ISSUE_NEXT()     /* stages are issued strictly in order */
#define WS2_WAIT_ONE_IN_FLIGHT() \
        __builtin_amdgcn_s_waitcnt(0x0F70 | (LOADS_PER_STAGE & 15) | ((LOADS_PER_STAGE >> 4) << 14))
        if (SPB == 2) {
            WS2_ISSUE(0);
            if (nk > 1) WS2_ISSUE(1);
            wait_vmem();
            __builtin_amdgcn_s_barrier();             // stages 0, 1 are in LDS
            if (nk > 2) WS2_ISSUE(2);
            if (nk > 3) WS2_ISSUE(3);
            for (int S = 0; 2 * S < nk; ++S) {
                wait_vmem();                          // stages 2S+2, 2S+3 have landed (a whole 32-row half to do so)
                __builtin_amdgcn_s_barrier();         // and every MFMA wave has finished reading stages 2S, 2S+1
                if (2 * S + 4 < nk) WS2_ISSUE(2 * S + 4);
                if (2 * S + 5 < nk) WS2_ISSUE(2 * S + 5);
            }
            return;
        }
        // RING - 1 stages are issued before anything is consumed; the barrier of iteration ks needs stage ks + 1 landed, the
        // (up to RING - 2) stages issued after it may stay in flight (counted s_waitcnt: loads retire in issue order)
#define WS2_WAIT_IN_FLIGHT(n_)                                                                                        \
        {                                                                                                            \
            const int ns_ = (p.dbg & (512 | 1024)) ? 0 : (n_);     /* ablated loads: the counts below would be wrong */ \
            if (ns_ <= 0) wait_vmem();                                                                               \
            else if (ns_ == 1) WS2_WAIT_ONE_IN_FLIGHT();                                                             \
            else if (ns_ == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * LOADS_PER_STAGE) & 15) | (((2 * LOADS_PER_STAGE) >> 4) << 14)); \
            else __builtin_amdgcn_s_waitcnt(0x0F70 | ((3 * LOADS_PER_STAGE) & 15) | (((3 * LOADS_PER_STAGE) >> 4) << 14)); \
        }
        static_assert(3 * LOADS_PER_STAGE < 64, "vmcnt is a 6-bit counter");
        int issued = 0;
        for (; issued < W2_NBUF - 1 && issued < nk; ++issued) WS2_ISSUE(issued);
        WS2_WAIT_IN_FLIGHT(issued - 1);
        __builtin_amdgcn_s_barrier();                 // stage 0 is in LDS
        if (issued < nk) { WS2_ISSUE(issued); ++issued; }
#ifdef NEMAR_TIMELINE
        long long lts[4][4];
        const bool lprobe = p.tl != nullptr && bx == 0 && by_ == 0;
#define WS2_LSTAMP(i_) if (lprobe && ks >= 40 && ks < 44) lts[ks - 40][i_] = clock64();
#else
#define WS2_LSTAMP(i_)
#endif
        for (int ks = 0; ks < nk; ++ks) {
            // stages ks + 2 .. issued - 1 may stay in flight; stage ks + 1 must have landed
            WS2_LSTAMP(0)
            WS2_WAIT_IN_FLIGHT(issued - (ks + 2));
            WS2_LSTAMP(1)
            if (!(p.dbg & 4)) __builtin_amdgcn_s_barrier();   // also: every MFMA wave has finished reading buffer ks % NBUF
            WS2_LSTAMP(2)
            if (issued < nk) { WS2_ISSUE(issued); ++issued; }
            WS2_LSTAMP(3)
        }
#ifdef NEMAR_TIMELINE
        if (lprobe && lane == 0 && nk >= 44) {
            long long* o = p.tl + (MT + ldr) * 24;          // behind the MFMA waves' 4 x 6 stamps per stage
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) o[i * 6 + j] = lts[i][j];
        }
#endif
#undef WS2_LSTAMP
#undef WS2_WAIT_IN_FLIGHT
#undef WS2_ISSUE
#undef WS2_ISSUE_NEXT
#undef WS2_ENTER_TAP
#undef WS2_WAIT_ONE_IN_FLIGHT
        return;
    }

    // ================================ MFMA waves ================================
    if (p.dbg & 128) __builtin_amdgcn_s_setprio(3);           // experiment: issue priority over the loader waves
    const int l31 = lane & 31, lhi = lane >> 5;
    const int a_off = (lhi * BM + wid * 32 + l31) * 4;        // + kg * 2*BM*4 floats for the second 8-row group
    const int b_off = lhi * LDB + 4 * l31;                    // + (8*kg + 2*s) * LDB
    // VEC: is this lane's 4-pixel group the first / last of a source row?  (those are the groups the loaders clamp)
    bool first_grp = false, last_grp = false;
    if (VEC) {
        const unsigned gp = (unsigned)min(p0 + 4 * l31, p.P - 1);
        const unsigned gn = fd_div(gp, p.fd_ohw);
        const unsigned grem = gp - gn * (unsigned)(p.OH * p.OW);
        const unsigned gox = grem - fd_div(grem, p.fd_ow) * (unsigned)p.OW;
        first_grp = gox == 0;
        last_grp = (int)gox == p.OW - 4;
    }
    const bool refl = p.border == BORDER_REFLECT;
    // sign of every tap's dx, 2 bits per tap (<= 32 taps on the VEC path: |dx| <= 1 means at most a 3x3 footprint... any
    // tap count up to 32 is representable); current tap / channel offset advance with the stages
    unsigned long long dxbits = 0ull;
    int tap_i = tap_start, tap_ch = ch_start;
    if (VEC)
        for (int t = 0; t < p.taps.n && t < 32; ++t)
            dxbits |= (unsigned long long)(p.taps.dx[t] < 0 ? 1u : p.taps.dx[t] > 0 ? 2u : 0u) << (2 * t);
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 a0, a1, b0[4], b1[4];

#define WS2_READ(buf_, kg_, A_, B_)                                                                        \
    {   /* (an ablation switch around these reads makes hipcc count the LDS waits of the MFMA blocks conservatively: measured \
           once — MFMA-only skeleton 302 us vs 370 us for the full kernel, gpurun_out/abl2 — and removed again) */  \
        const float* sb = Bs0 + (buf_) * B_FLOATS + (kg_) * (8 * LDB) + b_off;                                 \
        if (!ADIR) A_ = *reinterpret_cast<const f32x4*>(As0 + (buf_) * A_FLOATS + (kg_) * (2 * BM * 4) + a_off); \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) B_[s] = *reinterpret_cast<const f32x4*>(sb + 2 * s * LDB); \
    }
#define WS2_MFMA(A_, B_)                                                                                   \
    if (!(p.dbg & 2)) {                                                                                        \
        if (VEC && fix_l) {                                                                                    \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                    \
                const f32x4 v = B_[s];                                                                         \
                B_[s][0] = first_grp ? (refl ? v[1] : 0.f) : v[0];                                             \
                B_[s][1] = first_grp ? v[0] : v[1];                                                            \
                B_[s][2] = first_grp ? v[1] : v[2];                                                            \
                B_[s][3] = first_grp ? v[2] : v[3];                                                            \
            }                                                                                                  \
        }                                                                                                      \
        if (VEC && fix_r) {                                                                                    \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                    \
                const f32x4 v = B_[s];                                                                         \
                B_[s][0] = last_grp ? v[1] : v[0];                                                             \
                B_[s][1] = last_grp ? v[2] : v[1];                                                             \
                B_[s][2] = last_grp ? v[3] : v[2];                                                             \
                B_[s][3] = last_grp ? (refl ? v[2] : 0.f) : v[3];                                              \
            }                                                                                                  \
        }                                                                                                      \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                          \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                      \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[s], B_[s][t], acc[t], 0, 0, 0);               \
    }
    // timeline probe (nemar_tune_ptr): s_memtime stamps of stages 40..43 of every MFMA wave of workgroup (0,0), kept in
    // scalar registers and written once at the end: 6 stamps per stage = loop top | reads issued | MFMA block 1 issued |
    // lgkmcnt(0) | barrier passed | MFMA block 2 issued
    long long ts[4][6];
    const bool probe = p.tl != nullptr && bx == 0 && by_ == 0;
#ifdef NEMAR_TIMELINE      /* the probe's s_memtime + waits perturb the loop's wait counts: compiled in on demand only */
#define WS2_STAMP(i_)                                                     \
    if (probe && ks >= 40 && ks < 44) {                                       \
        const long long c_ = clock64();                                       \
        if (ks == 40) ts[0][i_] = c_; else if (ks == 41) ts[1][i_] = c_;      \
        else if (ks == 42) ts[2][i_] = c_; else ts[3][i_] = c_;               \
    }
#else
#define WS2_STAMP(i_)
#endif
    if (SPB == 2) {
        bool fix_l = false, fix_r = false;
#define WS2_STAGE_FLAGS()                                                       \
        {                                                                       \
            const unsigned dc_ = VEC ? (unsigned)(dxbits >> (2 * tap_i)) & 3u : 0u; \
            fix_l = dc_ == 1u;                                                  \
            fix_r = dc_ == 2u;                                                  \
            if (VEC) {                                                          \
                tap_ch += BK;                                                   \
                if (tap_ch >= Cs) { tap_ch -= Cs; ++tap_i; }                    \
            }                                                                   \
        }
        __builtin_amdgcn_s_barrier();                 // stages 0, 1 are in LDS
        WS2_READ(0, 0, a0, b0);
        for (int s0 = 0; s0 < nk; s0 += 2) {
            const int slot0 = s0 & 3, slot1 = (s0 + 1) & 3;
            const bool has1 = s0 + 1 < nk;
            WS2_STAGE_FLAGS();                        // border flags of stage s0
            WS2_READ(slot0, 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            WS2_MFMA(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (has1) {
                WS2_READ(slot1, 0, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                WS2_MFMA(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                WS2_STAGE_FLAGS();                    // border flags of stage s0 + 1
                WS2_READ(slot1, 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                WS2_MFMA(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): this wave is done reading both stages of the half
            __builtin_amdgcn_s_barrier();             // the next half has landed; this one goes back to the loaders
            if (s0 + 2 < nk) WS2_READ((s0 + 2) & 3, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            WS2_MFMA(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WS2_STAGE_FLAGS
    } else {
    // ADIR: this lane's A words of the two 8-row groups of a stage: gA + stage * (4 * Mpad * 4) + {0, 2 * Mpad * 4} floats
    const float* gA = p.wp + ((size_t)(ks0 * 4 + lhi) * p.Mpad + m0 + wid * 32 + l31) * 4;
    const size_t gA_kg = (size_t)2 * p.Mpad * 4, gA_stage = (size_t)4 * p.Mpad * 4;
    f32x4 na0, na1, a1n;
    if (ADIR) {
        na0 = *reinterpret_cast<const f32x4*>(gA);
        na1 = *reinterpret_cast<const f32x4*>(gA + gA_kg);
        gA += gA_stage;
    }
    __builtin_amdgcn_s_barrier();                     // stage 0 is in LDS
    WS2_READ(0, 0, a0, b0);
    if (ADIR) {
        a0 = na0; a1n = na1;
        if (nk > 1) {
            na0 = *reinterpret_cast<const f32x4*>(gA);
            na1 = *reinterpret_cast<const f32x4*>(gA + gA_kg);
            gA += gA_stage;
        }
    }
    int buf = 0;
    for (int ks = 0; ks < nk; ++ks) {
        // the tap of this stage decides which border groups need patching (wave-uniform, from registers: a table
        // lookup in the kernel arguments here costs a scalar-memory round trip per stage)
        const unsigned dcode = VEC ? (unsigned)(dxbits >> (2 * tap_i)) & 3u : 0u;
        const bool fix_l = dcode == 1u, fix_r = dcode == 2u;
        if (VEC) {
            tap_ch += BK;
            if (tap_ch >= Cs) { tap_ch -= Cs; ++tap_i; }
        }
        WS2_STAMP(0)
        WS2_READ(buf, 1, a1, b1);
        if (ADIR) a1 = a1n;
        __builtin_amdgcn_sched_barrier(0);
        WS2_STAMP(1)
        WS2_MFMA(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        WS2_STAMP(2)
        __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): this wave is done reading buffer `buf`
        WS2_STAMP(3)
        if (!(p.dbg & 4)) __builtin_amdgcn_s_barrier();   // stage ks+1 has landed; buffer `buf` goes back to the loaders (ablation 4: none)
        WS2_STAMP(4)
        buf = buf + 1 == W2_NBUF ? 0 : buf + 1;
        if (ks + 1 < nk) {
            WS2_READ(buf, 0, a0, b0);
            if (ADIR) {                               // next stage's A words (fetched a stage ago); fetch the one after
                a0 = na0; a1n = na1;
                if (ks + 2 < nk) {
                    na0 = *reinterpret_cast<const f32x4*>(gA);
                    na1 = *reinterpret_cast<const f32x4*>(gA + gA_kg);
                    gA += gA_stage;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        WS2_MFMA(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        WS2_STAMP(5)
    }
    }   // SPB == 1
#undef WS2_READ
#undef WS2_MFMA
#undef WS2_STAMP
#ifdef NEMAR_TIMELINE
    if (SPB == 1 && probe && lane == 0 && nk >= 44) {
        long long* o = p.tl + wid * 24;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 6; ++j) o[i * 6 + j] = ts[i][j];
    }
#else
    (void)ts; (void)probe;
#endif

    // ---- epilogue: lane owns pixels p0 + 4*l31 + {0..3} (tile t -> pixel t) of 16 channel rows ---------------------
    const size_t oplane = (size_t)p.OHf * p.OWf;
    const int M1 = p.M - p.M0;
    const int opix0 = p0 + 4 * l31;
    if (opix0 >= p.P) return;
    // split reduction: this split's partial tile goes to its own slab (no bias / activation / second destination then)
    float* const d0 = gridDim.z > 1 ? p.part + (size_t)blockIdx.z * (size_t)p.part_stride : p.dst0;
    // 16-byte stores when the 4 pixels are consecutive, in range and aligned in the destination
    const bool vec = p.osx == 1 && p.osy == 1 && p.oox == 0 && p.ooy == 0 && (p.OW & 3) == 0 &&
                     p.OW == p.OWf && p.OH == p.OHf && opix0 + 3 < p.P &&
                     ((reinterpret_cast<uintptr_t>(d0) | reinterpret_cast<uintptr_t>(p.dst1)) & 15) == 0;
    if (vec) {
        const unsigned n = fd_div((unsigned)opix0, p.fd_ohw);
        const unsigned rem = (unsigned)opix0 - n * (unsigned)(p.OH * p.OW);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < p.M) {
                const float bv = (p.bias && blockIdx.z == 0) ? p.bias[m] : 0.f;   // split reductions: slab 0 carries the bias
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = apply_act(acc[t][r] + bv, p.act, p.slope);
                float* dst = (m < p.M0) ? (d0 ? d0 + ((size_t)n * p.M0 + m) * oplane + rem : nullptr)
                                        : p.dst1 + ((size_t)n * M1 + (m - p.M0)) * oplane + rem;
                if (dst) *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int opix = opix0 + t;
        if (opix >= p.P) continue;
        const unsigned n = fd_div((unsigned)opix, p.fd_ohw);
        const unsigned rem = (unsigned)opix - n * (unsigned)(p.OH * p.OW);
        const unsigned oy = fd_div(rem, p.fd_ow);
        const unsigned ox = rem - oy * (unsigned)p.OW;
        const size_t sp = (size_t)((int)oy * p.osy + p.ooy) * p.OWf + ((int)ox * p.osx + p.oox);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < p.M) {
                float v = acc[t][r];
                if (p.bias && blockIdx.z == 0) v += p.bias[m];
                v = apply_act(v, p.act, p.slope);
                if (m < p.M0) {
                    if (d0) d0[((size_t)n * p.M0 + m) * oplane + sp] = v;
                } else {
                    p.dst1[((size_t)n * M1 + (m - p.M0)) * oplane + sp] = v;
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN>
void launch_igemm_cfg(const IgemmParams& p, bool fast, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    dim3 grid(nemar_cdiv(p.P, BN), nemar_cdiv(p.M, BM), p.ksplit), block(WM * WN * 64);
    if (fast)
        hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, true>), grid, block, g_lds_pad, st, p);
    else
        hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, false>), grid, block, g_lds_pad, st, p);
}

template <int MT>
void launch_ws2(const IgemmParams& p, bool vec, hipStream_t st) {
    dim3 grid(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 32 * MT), p.ring_p ? 1 : p.ksplit), block((MT + 2) * 64);
#ifdef NEMAR_AB
    if (vec && g_adir) hipLaunchKernelGGL((igemm_ws2_kernel<MT, true, 1, 2, true>), grid, block, g_lds_pad, st, p);
    else if (vec && g_ring == 4) hipLaunchKernelGGL((igemm_ws2_kernel<MT, true, 1, 2, false, 4>), grid, block, g_lds_pad, st, p);
    else if (vec && g_ring == 5) hipLaunchKernelGGL((igemm_ws2_kernel<MT, true, 1, 2, false, 5>), grid, block, g_lds_pad, st, p);
    else if (!vec) hipLaunchKernelGGL((igemm_ws2_kernel<MT, false>), grid, block, g_lds_pad, st, p);     // (key 11 = 0: 2 loader waves)
    else
#endif
    hipLaunchKernelGGL((igemm_ws2_kernel<MT, true>), grid, block, g_lds_pad, st, p);      // product build: vec only (launch_igemm)
}

// Tile selection.  The channel tile follows M; the pixel tile shrinks when the grid would leave most of the 256 CUs
// idle (the small-spatial discriminator / bottleneck layers): ~2 workgroups per CU is the target.
}  // namespace

namespace nemar_exact {

TileChoice igemm_tile(int M, int P, int stages, int ksplit) {
    const int kMinBlocks = g_min_blocks;
    TileChoice t;
    if (M > 64) {
        // 128x128 (wave-specialised for FAST shapes) needs ~1.5 workgroups per CU — or, measured on D's k4 layers, just
        // ~1 per CU when the reduction is deep enough (>= 64 stages) to amortise the lock-step prologue/epilogue
        t.bm = 128; t.bn = 128;
        const long long tiles = (long long)nemar_cdiv(M, 128) * nemar_cdiv(P, 128) * ksplit;
        const int need = (stages / ksplit >= 64 && kMinBlocks > 200) ? 200 : kMinBlocks;
        if (tiles < need) { t.bm = 64; t.bn = 64; }
    } else if (M > 32) {
        t.bm = 64; t.bn = 128;
        if ((long long)nemar_cdiv(P, 128) < kMinBlocks) t.bn = 64;
    } else {
        t.bm = 32; t.bn = 256;
        if ((long long)nemar_cdiv(P, 256) < kMinBlocks) t.bn = 128;
    }
    return t;
}
int igemm_mpad(int M) { return M > 32 ? nemar_cdiv(M, 256) * 256 : 32; }

// Does this launch go to the wave-specialised kernel (128 pixels x 32*MT channels), and with 16-byte B loads?  Measured:
// MT = 4 beats every generic configuration on layers big enough for 128x128 tiles; MT = 1, 2 (fewer MFMAs per staged B
// tile) lose to the generic 64x64 / 32x256 kernels and are only reachable through the tuning switch.
bool route_ws2(const IgemmParams& p, bool* vec_out) {
    const int Cs = p.C0 + p.C1;
    const bool fast = (Cs % BK == 0) && (p.C0 % BK == 0);
    const TileChoice t = igemm_tile(p.M, p.P, nemar_cdiv(p.Kred, BK), p.ring_p ? 1 : p.ksplit);
    if (!(fast && !p.ring_p && (g_cfg128 == 0 || (g_cfg128 >= 5 && g_cfg128 <= 7)) && (t.bm == 128 || g_ws2_mt))) return false;
    bool vec = g_cfg128 != 5 && p.sx == 1 && (p.OW & 3) == 0 && p.Ws == p.OW && p.Ws >= 4 && p.taps.n <= 32;
    for (int i = 0; i < p.taps.n && vec; ++i) vec = p.taps.dx[i] >= -1 && p.taps.dx[i] <= 1;
    *vec_out = vec;
    return true;
}

void launch_igemm(const IgemmParams& p, hipStream_t st) {
    const int Cs = p.C0 + p.C1;
    const bool fast = (Cs % BK == 0) && (p.C0 % BK == 0);
    TileChoice t = igemm_tile(p.M, p.P, nemar_cdiv(p.Kred, BK), p.ring_p ? 1 : p.ksplit);
    if (p.ring_p) {   // few pixels, full reduction depth: small tiles so the launch spreads over the CUs (generic kernel only)
        t.bm = p.M > 32 ? 64 : 32;
        t.bn = p.M > 32 ? 64 : 128;
    }
    bool vec = false;
    if (route_ws2(p, &vec)) {
        int mt = g_ws2_mt ? g_ws2_mt : 4;
        if (32 * mt > p.Mpad) mt = p.Mpad / 32;       // a forced tile must not read past the packed rows (M <= 32 packs 32 rows)
        IgemmParams q = p;
        q.xcd = (g_xcd_map && nemar_cdiv(p.P, 128) % 8 == 0) ? 1 : 0;
        const IgemmParams& p = q;
        // (product build: route_ws2 only says yes to 128-channel tiles, mt == 4, and the two default kernels below are all there is)
#ifdef NEMAR_AB
        if (mt == 4 && !vec && g_nl4_scalar && g_adir)
            hipLaunchKernelGGL((igemm_ws2_kernel<4, false, 1, 4, true>), dim3(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 128), p.ksplit),
                               dim3(8 * 64), g_lds_pad, st, p);
        else
#endif
        if (mt == 4 && !vec && g_nl4_scalar)     // gathered (non-VEC) B tile: 4 loader waves share the 32 4-byte loads
            hipLaunchKernelGGL((igemm_ws2_kernel<4, false, 1, 4>), dim3(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 128), p.ksplit),
                               dim3(8 * 64), g_lds_pad, st, p);
#ifdef NEMAR_AB
        else if (mt == 4 && vec && g_cfg128 == 7)     // experiment: 4 loader waves
            hipLaunchKernelGGL((igemm_ws2_kernel<4, true, 1, 4>), dim3(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 128)), dim3(8 * 64),
                               g_lds_pad, st, p);
        else if (mt == 4 && vec && g_cfg128 == 6)     // experiment: one barrier per 32 reduction rows
            hipLaunchKernelGGL((igemm_ws2_kernel<4, true, 2>), dim3(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 128)), dim3(6 * 64),
                               g_lds_pad, st, p);
        else if (mt == 4 && vec && g_mt8 && p.M > 128 && p.ksplit == 1) {
            // 256 channels x 128 pixels per workgroup (8 MFMA waves): one workgroup per CU, the B tile staged once per pixel
            // tile instead of once per 128-channel half, 24 instead of 32 KiB of global->LDS traffic per 2 x 16 reduction rows
            const dim3 g8(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 256));
            if (g_mt8 == 2) hipLaunchKernelGGL((igemm_ws2_kernel<8, true, 1, 4>), g8, dim3(12 * 64), g_lds_pad, st, p);
            else hipLaunchKernelGGL((igemm_ws2_kernel<8, true, 1, 2>), g8, dim3(10 * 64), g_lds_pad, st, p);
        }
        else if (mt == 2) launch_ws2<2>(p, vec, st);
        else if (mt == 1) launch_ws2<1>(p, vec, st);
#endif
        else launch_ws2<4>(p, vec, st);
        return;
    }
#ifdef NEMAR_AB
    if (t.bm == 128 && g_cfg128 == 3 && p.M >= 256) launch_igemm_cfg<4, 2, 2, 2>(p, fast, st);   // 256 x 128, 8 waves of 64x64
    else if (t.bm == 128 && fast && g_cfg128 == 4)                                     // 128 x 128, 8 MFMA + 2 loader waves
        hipLaunchKernelGGL(igemm_ws_kernel, dim3(nemar_cdiv(p.P, 128), nemar_cdiv(p.M, 128)), dim3(WS_NT), g_lds_pad, st, p);
    else if (t.bm == 128 && g_cfg128 == 1) launch_igemm_cfg<2, 2, 2, 2>(p, fast, st); // 128 x 128, 4 waves of 64x64
    else
#endif
    if (t.bm == 128) launch_igemm_cfg<2, 4, 2, 1>(p, fast, st);                       // 128 x 128, 8 waves of 64x32
    else if (t.bm == 64 && t.bn == 128) launch_igemm_cfg<1, 4, 2, 1>(p, fast, st);    // 64 x 128
    else if (t.bm == 64 && fast && (p.ring_p NEMAR_AB_ONLY(|| g_deep64)))                            // 64 x 64, 4-deep LDS ring
        hipLaunchKernelGGL((igemm_kernel<2, 2, 1, 1, true, 4>), dim3(nemar_cdiv(p.P, 64), nemar_cdiv(p.M, 64), p.ksplit),
                           dim3(256), g_lds_pad, st, p);
    else if (t.bm == 64) launch_igemm_cfg<2, 2, 1, 1>(p, fast, st);                   // 64 x 64
    else if (t.bn == 256) launch_igemm_cfg<1, 4, 1, 2>(p, fast, st);                  // 32 x 256
    else launch_igemm_cfg<1, 4, 1, 1>(p, fast, st);                                   // 32 x 128
}

size_t packed_core_floats(int M, int Kred) { return (size_t)nemar_cdiv(Kred, BK) * BK * (size_t)igemm_mpad(M); }
size_t packed_floats(int M, int Kred) { return packed_core_floats(M, Kred) + ZERO_PAGE; }

}  // namespace nemar_exact

namespace {

// the same pack as a job of a weight-pack plan (pack_plan.h): arguments from device memory, grid.z = job
struct ExactPackArgs {
    const float* w; float* wp;
    int M, Mpad, Cs, Kred, KredPad, wsm, wsc, zero_tail;
    int gx, gy;
    int wofs[MAX_TAPS];
};
__device__ __forceinline__ void exact_pack_body(const ExactPackArgs& a, int bx, int, int gx) {
    const int core = a.KredPad * a.Mpad, total = core + a.zero_tail;
    for (int idx = bx * 256 + threadIdx.x; idx < total; idx += gx * 256) {
        const int blk = idx / (a.Mpad * 4), within = idx - blk * (a.Mpad * 4);
        const int m = within >> 2, s = within & 3;
        const int kk = 8 * (blk >> 1) + 2 * s + (blk & 1);
        float v = 0.f;
        if (idx < core && m < a.M && kk < a.Kred) {
            const int t = kk / a.Cs, ch = kk - t * a.Cs;
            v = a.w[(size_t)m * a.wsm + (size_t)ch * a.wsc + a.wofs[t]];
        }
        a.wp[idx] = v;
    }
}
NEMAR_PACK_MULTI(exact_pack_multi_kernel, ExactPackArgs, exact_pack_body, 256)
void exact_pack_multi(const void* jobs, int njobs, int gx, int gy, hipStream_t st) {
    hipLaunchKernelGGL(exact_pack_multi_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, (const ExactPackArgs*)jobs);
}
struct RegExactPack {
    RegExactPack() { nemar_pack_register(PACK_FAM_EXACT, sizeof(ExactPackArgs), exact_pack_multi); }
} g_reg_exact_pack;

}  // namespace

namespace nemar_exact {

void launch_pack(const float* w, float* wp, int M, int Cs, int wsm, int wsc, const TapTable& taps, hipStream_t st) {
    const int Kred = taps.n * Cs;
    const int KredPad = nemar_cdiv(Kred, BK) * BK;
    const int Mpad = igemm_mpad(M);
    const int total = KredPad * Mpad + ZERO_PAGE;
    if (nemar_pack_recording()) {
        ExactPackArgs a;
        a.w = w; a.wp = wp; a.M = M; a.Mpad = Mpad; a.Cs = Cs; a.Kred = Kred; a.KredPad = KredPad; a.wsm = wsm; a.wsc = wsc;
        a.zero_tail = ZERO_PAGE; a.gx = nemar_stream_grid(total, 256); a.gy = 1;
        for (int i = 0; i < MAX_TAPS; ++i) a.wofs[i] = i < taps.n ? taps.wofs[i] : 0;
        nemar_pack_record_job(PACK_FAM_EXACT, &a, a.gx, 1);
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, w, wp, M, Mpad, Cs,
                       Kred, KredPad, wsm, wsc, ZERO_PAGE, taps);
}

}  // namespace nemar_exact
