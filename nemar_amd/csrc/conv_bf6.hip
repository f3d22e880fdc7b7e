// nemar_amd — 3x3 / stride-1 / pad-1 convolutions of the wide layers (the translation net's 256-channel residual blocks:
// reference models/networks.py:418-439, 18 convolutions per pass) on the BF16 matrix pipe, at fp32-equivalent accuracy.
//
// gfx950 has no TF32-like mode and its fp32 MFMA runs at the vector rate (157 TFLOP/s); the bf16 MFMA runs 16x faster.
// Every fp32 operand is split EXACTLY into three bf16 terms  v = b0 + b1 + b2  (b0 = RN(v), b1 = RN(v - b0), b2 = v - b0 - b1:
// 3 x 8 significant bits cover fp32's 24), and a product a*c is accumulated in fp32 from the six partial products of weight
// 2^-16 and above:  a0c0 + (a0c1 + a1c0) + (a0c2 + a1c1 + a2c0).  Each bf16 x bf16 product is exact in fp32; the three dropped terms
// (a1c2, a2c1, a2c2) are <= 2^-23 |a c| together, the size of the rounding error the fp32 MFMA commits on the product itself —
// so the result carries fp32-class error (tests/test_conv_real_shapes_gpu.py measures both paths against fp64 on the same
// inputs) at 6/16 of the fp32-MFMA issue time.
//
// Data flow of one convolution (all inside nemar_conv2d_fwd / nemar_conv2d_bwd_data):
//   1. split_planes_kernel: source [N,C,H,W] fp32 -> three bf16 planes, CHANNEL-BLOCKED and PADDED:
//         plane[t][n][c/8][row 0..H+3][slot 0..W+3][8 channels]      (16 bytes per (pixel, channel group))
//      rows 1..H / slots 1..W hold the image, row 0 / H+1 and slot 0 / W+1 the padding ALREADY MATERIALISED (zeros, or the
//      mirrored texels of nn.ReflectionPad2d(1)), rows H+2, H+3 and slots W+2, W+3 the pre-folded border sums the reflect DATA
//      gradient needs (below).  One lane's MFMA operand (8 consecutive reduction channels of one pixel) is one 16-byte word,
//      and a tile's halo (its rows + 2, full padded width) is ONE contiguous run per (plane, channel group).
//   2. igemm_bf6_kernel: implicit GEMM, workgroup tile = 128 output channels x 256 pixels (whole image rows), 4 MFMA waves
//      (128 channels x 64 pixels each: 8 accumulators of 32x32) + 2 loader waves.  Per 16-channel chunk of the reduction the
//      loaders DMA the tile's halo once (global_load_lds, 16 bytes per lane, no VGPRs) into a double-buffered LDS region and all
//      nine taps read it at shifted addresses — the source is fetched once per chunk instead of once per tap — plus one 12 KiB
//      stage of packed weights per tap into a ring.  One workgroup barrier per tap (48 MFMAs per wave).
//   3. the reflect data gradient  dx = Pad^T(Conv^T(gy))  folds the padded border back: output row 1 receives the gradient of
//      padded row 0, which only the first filter row produces, i.e. for that (row, tap) pair the source row is gy[0] + gy[2]
//      instead of gy[2]; same for row H-2, columns 1 and W-2, and the four corners.  The split kernel writes those sums once
//      (rows H+2 / H+3, slots W+2 / W+3) and the MFMA waves select them by address: no ring launch, no extra taps.
// Weights are split and re-ordered once per optimizer step (bf6_pack_kernel) into [chunk][tap][128-row block][plane][k group]
// [128][8]: a stage is one contiguous 12 KiB run.
#include "common.h"
#include "conv_bf6.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rn(float v) {
    const __bf16 h = (__bf16)v;                                   // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf16_val(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// v == b0 + b1 + b2 exactly (both subtractions are exact in fp32: Sterbenz-type cancellation of the leading bits)
__device__ __forceinline__ void split3(float v, unsigned short& b0, unsigned short& b1, unsigned short& b2) {
    b0 = bf16_rn(v);
    const float r1 = v - bf16_val(b0);
    b1 = bf16_rn(r1);
    const float r2 = r1 - bf16_val(b1);
    b2 = bf16_rn(r2);
}

__device__ __forceinline__ int mirror(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// value of plane position (row, slot) of one channel image xc [H, W]
__device__ __forceinline__ float plane_value(const float* xc, int row, int slot, int H, int W, int mode) {
    if (mode != BF6_DGRAD_REFLECT) {
        if (row > H + 1 || slot > W + 1) return 0.f;
        int y = row - 1, x = slot - 1;
        if (mode == BF6_REFLECT) {
            y = mirror(y, H);
            x = mirror(x, W);
        } else if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) {
            return 0.f;
        }
        return xc[y * W + x];
    }
    int ya, yb = -1, xa, xb = -1;
    if (row >= 1 && row <= H) ya = row - 1;
    else if (row == H + 2) { ya = 0; yb = 2; }
    else if (row == H + 3) { ya = H - 3; yb = H - 1; }
    else return 0.f;
    if (slot >= 1 && slot <= W) xa = slot - 1;
    else if (slot == W + 2) { xa = 0; xb = 2; }
    else if (slot == W + 3) { xa = W - 3; xb = W - 1; }
    else return 0.f;
    float v = xc[ya * W + xa];
    if (yb >= 0) v += xc[yb * W + xa];
    if (xb >= 0) {
        float u = xc[ya * W + xb];
        if (yb >= 0) u += xc[yb * W + xb];
        v += u;
    }
    return v;
}

__device__ __forceinline__ u32x4 pack8(const unsigned short* b) {
    u32x4 o;
    o[0] = (unsigned)b[0] | ((unsigned)b[1] << 16);
    o[1] = (unsigned)b[2] | ((unsigned)b[3] << 16);
    o[2] = (unsigned)b[4] | ((unsigned)b[5] << 16);
    o[3] = (unsigned)b[6] | ((unsigned)b[7] << 16);
    return o;
}

// one thread = one (n, channel group, row, slot): 8 strided reads (coalesced across the slots of a row), 3 x 16-byte writes
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, u32x4* __restrict__ out, int N, int C,
                                                           int H, int W, int mode, long long total) {
    const int Hp = H + 4, Ws = W + 4, CG = C >> 3;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int slot = (int)(t % Ws);
        long long q = t / Ws;
        const int row = (int)(q % Hp);
        q /= Hp;
        const int cg = (int)(q % CG), n = (int)(q / CG);
        const float* xc = x + ((size_t)n * C + (size_t)cg * 8) * H * W;
        unsigned short b0[8], b1[8], b2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split3(plane_value(xc + (size_t)j * H * W, row, slot, H, W, mode), b0[j], b1[j], b2[j]);
        out[t] = pack8(b0);
        out[total + t] = pack8(b1);
        out[2 * total + t] = pack8(b2);
    }
}

// packed weights: 16-byte word index (((chunk * 9 + tap) * mblks + mblk) * 3 + plane) * 256 + kgroup * 128 + m
__global__ __launch_bounds__(256) void bf6_pack_kernel(const float* __restrict__ w, u32x4* __restrict__ out, int M, int Cred,
                                                       int dgrad) {
    const int mblks = M >> 7;
    const long long total = (long long)(Cred >> 4) * 9 * mblks * 256;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(t & 127), kg = (int)((t >> 7) & 1);
        long long q = t >> 8;
        const int mblk = (int)(q % mblks);
        q /= mblks;
        const int tap = (int)(q % 9), chunk = (int)(q / 9);
        const int mg = mblk * 128 + m;
        unsigned short b0[8], b1[8], b2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cr = chunk * 16 + kg * 8 + j;
            // forward: w[K = M][C = Cred][3][3];  data gradient: w[K = Cred][C = M][3][3] with the taps flipped
            const float v = dgrad ? w[((size_t)cr * M + mg) * 9 + (8 - tap)] : w[((size_t)mg * Cred + cr) * 9 + tap];
            split3(v, b0[j], b1[j], b2[j]);
        }
        u32x4* o = out + (((size_t)(chunk * 9 + tap) * mblks + mblk) * 3) * 256 + kg * 128 + m;
        o[0] = pack8(b0);
        o[256] = pack8(b1);
        o[512] = pack8(b2);
    }
}

struct Bf6Params {
    const u32x4* planes;       // split source, see split_planes_kernel
    const u32x4* wp;           // packed weights
    const float* bias;         // [M] or null
    float* dst;                // [N, M, H, W]
    int N, H, W, M, Cred;
    int Ws, HpWs;              // slots per plane row, 16-byte words per (plane, n, channel group) image
    int wshift, RT;            // log2 W, output rows per tile (256 / W)
    int tiles_per_img, mblks;
    int halo_instr, aux_instr; // 1 KiB DMA instructions per (plane, k group) for the halo rows / the two folded rows
    int halo16, aux16;         // exact 16-byte words of those two runs (second-generation kernel: exact-sized LDS regions)
    int fold;                  // reflect data gradient: select the folded rows / slots
    int xcd;                   // workgroup -> tile mapping keeps neighbouring tiles on one XCD
    long long plane16;         // 16-byte words per plane
    long long* tl;             // NEMAR_TIMELINE builds: cycle stamps of workgroup 0 (tools/timeline_bf6.py)
    int rot;                   // > 1: chunk order rotated per tile in `rot` groups (second-generation kernel)
    int dbg;                   // ablation bits (nemar_tune key 2): 0x10000 no MFMAs, 0x20000 no LDS fragment reads, 0x40000 no global->LDS copies
};

__device__ __forceinline__ void glds16(const u32x4* g, u32x4* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// scalar (wave-uniform) base + per-lane byte offset: hipcc selects the SGPR-base form of global_load_lds for it
__device__ __forceinline__ void glds16u(const u32x4* ubase, unsigned lane_bytes, u32x4* lds) {
    glds16(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(ubase) + lane_bytes), lds);
}

// REGION_KB = KiB of LDS per (plane, k group) halo region = DMA instructions per region; RING = weight-stage ring depth
template <int REGION_KB, int RING>
__global__ __launch_bounds__(384) void igemm_bf6_kernel(Bf6Params p) {
    constexpr int REGION16 = REGION_KB * 64;             // 16-byte words per region
    constexpr int BBUF16 = 6 * REGION16;                 // 3 planes x 2 k groups
    constexpr int ASTAGE16 = 6 * 128;                    // 3 planes x 2 k groups x 128 rows
    constexpr int NBL = 3 * REGION_KB;                   // halo instructions per loader per chunk (loader = k group)
    constexpr int WINDOW = 10 - RING;                    // stages (taps RING-1 .. 8) that carry the next chunk's halo
    constexpr int NB = (NBL + WINDOW - 1) / WINDOW;      // halo instruction slots per stage
    constexpr int PER = 6 + NB;                          // DMA instructions per loader per stage (constant: counted waits)
    static_assert((RING - 1) * PER < 64, "vmcnt is a 6-bit counter");
    static_assert((2 * BBUF16 + RING * ASTAGE16) * 16 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) u32x4 smem[2 * BBUF16 + RING * ASTAGE16];
    u32x4* const Bs = smem;
    u32x4* const As = smem + 2 * BBUF16;

    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    // consecutive workgroup ids sit on consecutive XCDs: give every XCD a contiguous run of tiles, and both channel halves of a
    // pixel tile (same halo) to the same one
    int t = blockIdx.x;
    if (p.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);
    const int ptile = t / p.mblks, mblk = t - ptile * p.mblks;
    const int n = ptile / p.tiles_per_img, y0 = (ptile - n * p.tiles_per_img) * p.RT;
    const int nchunks = p.Cred >> 4, nstage = nchunks * 9;
    const int CG = p.Cred >> 3;

    if (wid >= 4) {
        // ================================ loader waves: wave 4 = k group 0, wave 5 = k group 1 ================================
        const int kg = wid - 4;
        const int ipr = p.halo_instr + p.aux_instr;                      // <= REGION_KB
        const int nbl = 3 * ipr;
        const u32x4* const wsrc0 = p.wp + (size_t)mblk * 768 + kg * 384 + lane;
        const size_t wstage = (size_t)p.mblks * 768;
        // source of this loader's halo words of chunk c, plane t: planes + t * plane16 + ((n * CG + 2c + kg) * HpWs) + ...
        const u32x4* const bsrc0 = p.planes + ((size_t)n * CG + kg) * p.HpWs + lane;
        const int halo_off = y0 * p.Ws, aux_off = (p.H + 2) * p.Ws;
        int islot = 0, ci = 0, ti = 0;        // ring slot / chunk / tap of the next stage to issue
        int bpl = 0, bin = 0, bcnt = nbl;      // halo stream of chunk ci + 1: plane, instruction within the region, issued count
        const u32x4* asrc = wsrc0;

#define BF6_HALO_ONE(chunk_, pl_, in_)                                                                                  \
        {                                                                                                               \
            const u32x4* g_ = bsrc0 + (size_t)(pl_) * p.plane16 + (size_t)(2 * (chunk_)) * p.HpWs +                     \
                              ((in_) < p.halo_instr ? halo_off + (in_) * 64 : aux_off + ((in_) - p.halo_instr) * 64);   \
            glds16(g_, Bs + ((chunk_) & 1) * BBUF16 + ((pl_) * 2 + kg) * REGION16 + (in_) * 64);                        \
        }
#define BF6_ISSUE()                                                                                                     \
        {                                                                                                               \
            u32x4* const ad_ = As + islot * ASTAGE16 + kg * 384;                                                        \
            _Pragma("unroll") for (int q = 0; q < 6; ++q) glds16(asrc + q * 64, ad_ + q * 64);                          \
            if (ti == RING - 1) { bpl = 0; bin = 0; bcnt = (ci + 1 < nchunks) ? 0 : nbl; }                              \
            _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                                            \
                if (bcnt < nbl) {                                                                                       \
                    BF6_HALO_ONE(ci + 1, bpl, bin);                                                                     \
                    ++bcnt;                                                                                             \
                    if (++bin == ipr) { bin = 0; ++bpl; }                                                               \
                } else {                                                                                                \
                    glds16(asrc, ad_);             /* filler: keeps the per-stage instruction count constant */          \
                }                                                                                                       \
            }                                                                                                           \
            asrc += wstage;                                                                                             \
            islot = islot + 1 == RING ? 0 : islot + 1;                                                                  \
            if (++ti == 9) { ti = 0; ++ci; }                                                                            \
        }
#define BF6_WAIT_IN_FLIGHT(n_)                                                                                          \
        {                                                                                                               \
            const int ns_ = (n_);                                                                                       \
            if (ns_ <= 0) wait_vmem();                                                                                  \
            else if (ns_ == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (PER & 15) | ((PER >> 4) << 14));                    \
            else __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * PER) & 15) | (((2 * PER) >> 4) << 14));                      \
        }
        static_assert(RING <= 4, "two stages in flight at most");
        // halo of chunk 0, then RING - 1 weight stages, before anything is consumed
        for (int pl = 0; pl < 3; ++pl)
            for (int in = 0; in < ipr; ++in) BF6_HALO_ONE(0, pl, in);
        int issued = 0;
        for (; issued < RING - 1 && issued < nstage; ++issued) BF6_ISSUE();
        BF6_WAIT_IN_FLIGHT(issued - 1);
        __builtin_amdgcn_s_barrier();                 // stage 0 (and the first halo) are in LDS
        if (issued < nstage) { BF6_ISSUE(); ++issued; }
        for (int ks = 0; ks < nstage; ++ks) {
            BF6_WAIT_IN_FLIGHT(issued - (ks + 2));    // stage ks + 1 has landed (with everything issued before it)
            __builtin_amdgcn_s_barrier();             // every MFMA wave has finished reading ring slot ks % RING
            if (issued < nstage) { BF6_ISSUE(); ++issued; }
        }
#undef BF6_WAIT_IN_FLIGHT
#undef BF6_ISSUE
#undef BF6_HALO_ONE
        return;
    }

    // ================================ MFMA waves: 128 channels x 64 pixels each ================================
    const int l31 = lane & 31, lhi = lane >> 5;
    int row[2], col[2];
    bool top[2], bot[2], lft[2], rgt[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int px = 64 * wid + 32 * nt + l31;
        row[nt] = px >> p.wshift;
        col[nt] = px & (p.W - 1);
        const int y = y0 + row[nt];
        top[nt] = p.fold && y == 1;
        bot[nt] = p.fold && y == p.H - 2;
        lft[nt] = p.fold && col[nt] == 1;
        rgt[nt] = p.fold && col[nt] == p.W - 2;
    }
    const int auxoff = p.halo_instr * 64;             // the two folded rows sit behind the halo rows of a region
    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    __builtin_amdgcn_s_barrier();                     // stage 0 is in LDS
    int slot = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const u32x4* const Bb = Bs + (chunk & 1) * BBUF16 + lhi * REGION16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, sx = tap - 3 * (tap / 3);
            u32x4 bf[2][3], af[4][3];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                int ra = (row[nt] + r) * p.Ws;
                if (r == 2) ra = top[nt] ? auxoff : ra;
                if (r == 0) ra = bot[nt] ? auxoff + p.Ws : ra;
                int sl = col[nt] + sx;
                if (sx == 2) sl = lft[nt] ? p.W + 2 : sl;
                if (sx == 0) sl = rgt[nt] ? p.W + 3 : sl;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[nt][pl] = Bb[pl * 2 * REGION16 + ra + sl];
            }
            const u32x4* const Ab = As + slot * ASTAGE16 + lhi * 128 + l31;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[mt][pl] = Ab[pl * 256 + mt * 32];
            // six partial products, smallest first; consecutive MFMAs go to different accumulators
#define BF6_TERM(pa_, pb_)                                                                                              \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                            \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                        \
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mt][pa_]),      \
                                                                          __builtin_bit_cast(bf16x8, bf[nt][pb_]),      \
                                                                          acc[mt][nt], 0, 0, 0);
            BF6_TERM(2, 0)
            BF6_TERM(1, 1)
            BF6_TERM(0, 2)
            BF6_TERM(1, 0)
            BF6_TERM(0, 1)
            BF6_TERM(0, 0)
#undef BF6_TERM
            __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): this wave is done reading the stage
            __builtin_amdgcn_s_barrier();             // the next stage has landed; this ring slot goes back to the loaders
            slot = slot + 1 == RING ? 0 : slot + 1;
        }
    }

    // epilogue: D register r of lane l = channel (r & 3) + 8 (r >> 2) + 4 (l >> 5), pixel l & 31 of the 32 x 32 tile
    const size_t HW = (size_t)p.H * p.W;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float* const d0 = p.dst + (size_t)n * p.M * HW + (size_t)(y0 + row[nt]) * p.W + col[nt];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mblk * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float v = acc[mt][nt][r];
                if (p.bias) v += p.bias[m];
                d0[(size_t)m * HW] = v;
            }
        }
    }
}

// ---- second generation: same data flow, MFMA waves software-pipelined -------------------------------------------------
// First generation above: per tap every MFMA wave reads its 18 fragments, waits for them, then issues 48 MFMAs — LDS latency and
// the workgroup barrier sit exposed between two MFMA blocks (measured: 2.2x the pure MFMA time).  Here every fragment is fetched
// one 12-MFMA block ahead of its use: the A fragments of channel tile mt + 1 while the MFMAs of tile mt issue (two tiles' worth
// of registers instead of four), tile 0 of tap T + 1 during the last block of tap T, the B fragments of tap T + 1 during the
// second half of tap T (pixel tile 1 into spare registers, pixel tile 0 into the registers its predecessor vacates after the
// first half of the last block).  For that the loaders keep one more stage landed: barrier j certifies stage j + 2 (a stage
// is issued RING - 2 taps before it is needed).  LDS regions are exact-sized (the last 1 KiB copy of a run is lane-masked) so
// that the reflect data gradient fits a 4-slot ring too.  hipcc's scheduler undoes the interleaving (it sinks every read to the
// end of the tap and the MFMAs below the barrier): the groups are pinned with sched_barrier.
template <int NB, int ABL = 0>      // ABL: timing ablations (results are garbage): 1 no MFMAs, 2 no LDS fragment reads
__global__ __launch_bounds__(384) void igemm_bf6p_kernel(Bf6Params p) {
    constexpr int RING = 4, ASTAGE16 = 768, PER = 6 + NB;
    constexpr int WINDOW_FIRST = RING - 2;               // taps WINDOW_FIRST .. 8 of a chunk carry the next chunk's halo copies
    constexpr int SMEM16 = 9728;                          // 152 KiB: ring 48 KiB + two halo buffers of <= 52 KiB
    static_assert(2 * PER < 64, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) u32x4 smem[SMEM16];
    u32x4* const As = smem;
    u32x4* const Bs = smem + RING * ASTAGE16;
    const int region16 = p.halo16 + p.aux16, bbuf16 = 6 * region16;

    // wave id as a SCALAR: everything the loaders derive from it (k group, global bases, LDS bases) then lives in SGPRs and the
    // copies use the scalar-base + lane-offset address form — no address VGPR is rewritten between two copies in flight
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int t = blockIdx.x;
    if (p.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);
    const int ptile = t / p.mblks, mblk = t - ptile * p.mblks;
    const int n = ptile / p.tiles_per_img, y0 = (ptile - n * p.tiles_per_img) * p.RT;
    const int nchunks = p.Cred >> 4, nstage = nchunks * 9;
    const int CG = p.Cred >> 3;

    if (wid >= 4) {
        // ================================ loader waves ================================
        const int kg = wid - 4;
        const int hi = (p.halo16 + 63) >> 6, ai = (p.aux16 + 63) >> 6, ipr = hi + ai, nbl = 3 * ipr;
        const unsigned lane16 = (unsigned)lane * 16u;                    // this lane's byte offset inside a 1 KiB copy
        const u32x4* const wsrc0 = p.wp + (size_t)mblk * 768 + kg * 384;
        const size_t wstage = (size_t)p.mblks * 768;
        const u32x4* const bsrc0 = p.planes + ((size_t)n * CG + kg) * p.HpWs;
        const int halo_off = y0 * p.Ws, aux_off = (p.H + 2) * p.Ws;
        const bool offA = (p.dbg & (0x40000 | 0x80000)) != 0, offB = (p.dbg & (0x40000 | 0x100000)) != 0, off = offA || offB;
        // Every workgroup walks the same packed weights; started together they would all ask the same few L2 channels for
        // the same 12 KiB stage at the same moment.  p.rot > 1: workgroups start the reduction at different 16-channel chunks
        // (chunk order rotated by tile index; the summation order of a tile is still fixed, run to run).
        const int rot = p.rot > 1 ? (ptile % p.rot) * (nchunks / p.rot) : 0;
        int islot = 0, ci = 0, ti = 0;
        int ce = rot;                          // reduction chunk that sequence position ci maps to
        int ce1 = rot + 1 == nchunks ? 0 : rot + 1;
        int bpl = 0, bin = 0, bcnt = nbl;
        const u32x4* asrc = wsrc0 + (size_t)rot * 9 * wstage;
#define BF6P_HALO_ONE(seq_, chunk_, pl_, in_)             /* seq_ = position in this workgroup's chunk order */        \
        {                                                                                                               \
            const bool aux_ = (in_) >= hi;                                                                              \
            const int j_ = aux_ ? (in_) - hi : (in_);                                                                   \
            const int w_ = j_ * 64 + lane;                                                                              \
            if (w_ < (aux_ ? p.aux16 : p.halo16) && !offB)                                                              \
                glds16u(bsrc0 + (size_t)(pl_) * p.plane16 + (size_t)(2 * (chunk_)) * p.HpWs + (aux_ ? aux_off : halo_off) + j_ * 64, lane16, \
                       Bs + ((seq_) & 1) * bbuf16 + ((pl_) * 2 + kg) * region16 + (aux_ ? p.halo16 : 0) + j_ * 64);     \
        }
#define BF6P_ISSUE()                                                                                                    \
        {                                                                                                               \
            u32x4* const ad_ = As + islot * ASTAGE16 + kg * 384;                                                        \
            if (!offA) {                                                                                                \
                /* all six addresses first, in six DIFFERENT register pairs: rewriting the address VGPRs of a copy that is   \
                   still in flight stalls the wave until that copy completes (measured: 290 cycles per copy, L2 latency) */   \
                const u32x4* a_[6];                                                                                     \
                _Pragma("unroll") for (int q = 0; q < 6; ++q)                                                           \
                    a_[q] = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(asrc + q * 64) + lane16);      \
                __builtin_amdgcn_sched_barrier(0);                                                                      \
                _Pragma("unroll") for (int q = 0; q < 6; ++q) glds16(a_[q], ad_ + q * 64);                              \
                __builtin_amdgcn_sched_barrier(0);                                                                      \
            }                                                                                                           \
            BF6P_ASTAMP()                                                                                               \
            if (ti == WINDOW_FIRST) { bpl = 0; bin = 0; bcnt = (ci + 1 < nchunks) ? 0 : nbl; }                          \
            _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                                            \
                if (bcnt < nbl) {                                                                                       \
                    BF6P_HALO_ONE(ci + 1, ce1, bpl, bin);                                                               \
                    ++bcnt;                                                                                             \
                    if (++bin == ipr) { bin = 0; ++bpl; }                                                               \
                } else if (!offA) {                                                                                     \
                    glds16u(asrc, lane16, ad_);    /* filler: keeps the per-stage instruction count constant */          \
                }                                                                                                       \
            }                                                                                                           \
            asrc += wstage;                                                                                             \
            islot = (islot + 1) & (RING - 1);                                                                           \
            if (++ti == 9) {                                                                                            \
                ti = 0; ++ci;                                                                                           \
                ce = ce1;                                                                                               \
                ce1 = ce1 + 1 == nchunks ? 0 : ce1 + 1;                                                                 \
                if (ce == 0) asrc = wsrc0;                 /* wrapped around the end of the packed weights */           \
            }                                                                                                           \
        }
#ifdef NEMAR_TIMELINE
        long long astamp = 0;
#define BF6P_ASTAMP() astamp = clock64();
#else
#define BF6P_ASTAMP()
#endif
#define BF6P_WAIT_IN_FLIGHT(n_)                                                                                         \
        {                                                                                                               \
            const int ns_ = off ? 0 : (n_);                                                                             \
            if (ns_ <= 0) wait_vmem();                                                                                  \
            else if (ns_ == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (PER & 15) | ((PER >> 4) << 14));                    \
            else __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * PER) & 15) | (((2 * PER) >> 4) << 14));                      \
        }
        for (int pl = 0; pl < 3; ++pl)
            for (int in = 0; in < ipr; ++in) BF6P_HALO_ONE(0, rot, pl, in);
        int issued = 0;
        for (; issued < RING && issued < nstage; ++issued) BF6P_ISSUE();
        BF6P_WAIT_IN_FLIGHT(issued - 2);              // stages 0 and 1 (and the first halo) have landed
        __builtin_amdgcn_s_barrier();                 // B_-1
#ifdef NEMAR_TIMELINE
        const bool lprobe = p.tl != nullptr && blockIdx.x == 0 && lane == 0;
#define BF6P_LSTAMP(i_) if (lprobe && j >= 40 && j < 48) p.tl[(wid * 8 + (j - 40)) * 4 + (i_)] = ((i_) == 3 ? astamp : clock64());
#else
#define BF6P_LSTAMP(i_)
#endif
        for (int j = 0; j < nstage; ++j) {
            BF6P_LSTAMP(0)
            BF6P_WAIT_IN_FLIGHT(issued - (j + 3));    // stage j + 2 has landed
            BF6P_LSTAMP(1)
            __builtin_amdgcn_s_barrier();             // B_j: the MFMA waves have finished reading stage j
            BF6P_LSTAMP(2)
            if (issued < nstage) { BF6P_ISSUE(); ++issued; }
            BF6P_LSTAMP(3)                            // (the stamp taken after the six weight copies of this issue)
        }
#undef BF6P_LSTAMP
#undef BF6P_ASTAMP
#undef BF6P_WAIT_IN_FLIGHT
#undef BF6P_ISSUE
#undef BF6P_HALO_ONE
        return;
    }

    // ================================ MFMA waves ================================
    const int l31 = lane & 31, lhi = lane >> 5;
    int row[2], col[2];
    bool top[2], bot[2], lft[2], rgt[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int px = 64 * wid + 32 * nt + l31;
        row[nt] = px >> p.wshift;
        col[nt] = px & (p.W - 1);
        const int y = y0 + row[nt];
        top[nt] = p.fold && y == 1;
        bot[nt] = p.fold && y == p.H - 2;
        lft[nt] = p.fold && col[nt] == 1;
        rgt[nt] = p.fold && col[nt] == p.W - 2;
    }
    const int auxoff = p.halo16;
    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    constexpr bool noread = (ABL & 2) != 0, nomfma = (ABL & 1) != 0;

    // B fragments of pixel tile nt_ for tap (r, sx) from halo buffer `hb_`
#define BF6P_READ_B(dst_, nt_, hb_, r_, sx_)                                                                            \
    if constexpr (!noread) {                                                                                                      \
        const u32x4* const Bb_ = Bs + (hb_) * bbuf16 + lhi * region16;                                                  \
        int ra_ = (row[nt_] + (r_)) * p.Ws;                                                                             \
        if ((r_) == 2) ra_ = top[nt_] ? auxoff : ra_;                                                                   \
        if ((r_) == 0) ra_ = bot[nt_] ? auxoff + p.Ws : ra_;                                                            \
        int sl_ = col[nt_] + (sx_);                                                                                     \
        if ((sx_) == 2) sl_ = lft[nt_] ? p.W + 2 : sl_;                                                                 \
        if ((sx_) == 0) sl_ = rgt[nt_] ? p.W + 3 : sl_;                                                                 \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) dst_[pl] = Bb_[pl * 2 * region16 + ra_ + sl_];                 \
    }
#define BF6P_READ_A(dst_, slot_, mt_)                                                                                   \
    if constexpr (!noread) {                                                                                                      \
        const u32x4* const Ab_ = As + (slot_) * ASTAGE16 + lhi * 128 + l31 + (mt_) * 32;                                \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) dst_[pl] = Ab_[pl * 256];                                      \
    }
#define BF6P_MFMA1(mt_, nt_, A_, B_, q_)                                                                               \
    acc[mt_][nt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_[PA[q_]]),                     \
                                                            __builtin_bit_cast(bf16x8, B_[PB[q_]]), acc[mt_][nt_], 0, 0, 0);
    // six partial products, smallest first; consecutive MFMAs alternate between the two pixel tiles' accumulators
#define BF6P_MFMA(mt_, A_)                                                                                              \
    if constexpr (!nomfma) {                                                                                                      \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                                                 \
            BF6P_MFMA1(mt_, 0, A_, b0, q)                                                                               \
            BF6P_MFMA1(mt_, 1, A_, b1, q)                                                                               \
        }                                                                                                               \
    }
#define BF6P_MFMA_NT(mt_, nt_, A_, B_)                                                                                  \
    if constexpr (!nomfma) {                                                                                                      \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) BF6P_MFMA1(mt_, nt_, A_, B_, q)                                   \
    }
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
    const u32x4 one8 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    u32x4 a0[3], a1[3], a2[3], a3[3], b0[3], b1[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) a0[pl] = a1[pl] = a2[pl] = a3[pl] = b0[pl] = b1[pl] = one8;

    __builtin_amdgcn_s_barrier();                     // B_-1: stages 0, 1 and the first halo are in LDS
    BF6P_READ_B(b0, 0, 0, 0, 0)
    BF6P_READ_B(b1, 1, 0, 0, 0)
    BF6P_READ_A(a0, 0, 0)
    int slot = 0;
#define BF6P_PIN() __builtin_amdgcn_sched_barrier(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int hb = chunk & 1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // tap T + 1 (after the last tap: harmless reads of stale LDS)
            const int ntap = tap == 8 ? 0 : tap + 1;
            const int nr = ntap / 3, nsx = ntap - 3 * (ntap / 3);
            const int nhb = tap == 8 ? hb ^ 1 : hb;
            const int nslot = (slot + 1) & (RING - 1);
            u32x4 m1[3];
            BF6P_PIN()
            BF6P_READ_A(a1, slot, 1)
            BF6P_PIN()
            BF6P_MFMA(0, a0)
            BF6P_PIN()
            BF6P_READ_A(a2, slot, 2)
            BF6P_PIN()
            BF6P_MFMA(1, a1)
            BF6P_PIN()
            BF6P_READ_A(a3, slot, 3)
            BF6P_READ_B(m1, 1, nhb, nr, nsx)
            BF6P_PIN()
            BF6P_MFMA(2, a2)
            BF6P_PIN()
            BF6P_READ_A(a0, nslot, 0)
            BF6P_PIN()
            BF6P_MFMA_NT(3, 0, a3, b0)
            BF6P_PIN()
            BF6P_READ_B(b0, 0, nhb, nr, nsx)
            BF6P_PIN()
            BF6P_MFMA_NT(3, 1, a3, b1)
            BF6P_PIN()
#ifdef NEMAR_TIMELINE
            const int T_ = chunk * 9 + tap;
            const bool mprobe = p.tl != nullptr && blockIdx.x == 0 && lane == 0 && T_ >= 40 && T_ < 48;
            if (mprobe) p.tl[(wid * 8 + (T_ - 40)) * 4 + 0] = clock64();
#endif
            __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): this wave is done with stage T (and holds tile 0 of T + 1)
#ifdef NEMAR_TIMELINE
            if (mprobe) p.tl[(wid * 8 + (T_ - 40)) * 4 + 1] = clock64();
#endif
            __builtin_amdgcn_s_barrier();             // B_T: stage T + 2 has landed, the slot of stage T goes back to the loaders
#ifdef NEMAR_TIMELINE
            if (mprobe) p.tl[(wid * 8 + (T_ - 40)) * 4 + 2] = clock64();
#endif
            if constexpr (!noread) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b1[pl] = m1[pl];
            }
            slot = nslot;
        }
    }
#undef BF6P_PIN
#undef BF6P_MFMA_NT
#undef BF6P_MFMA
#undef BF6P_MFMA1
#undef BF6P_READ_A
#undef BF6P_READ_B

    const size_t HW = (size_t)p.H * p.W;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float* const d0 = p.dst + (size_t)n * p.M * HW + (size_t)(y0 + row[nt]) * p.W + col[nt];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mblk * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float v = acc[mt][nt][r];
                if (p.bias) v += p.bias[m];
                d0[(size_t)m * HW] = v;
            }
        }
    }
}

// ---- third generation: no loader waves ---------------------------------------------------------------------------------
// Measured on the second generation (tools/timeline_bf6.py, tools/probes/dma_probe2.hip): a wave that streams MFMAs back to back
// monopolises the instruction issue of its SIMD.  The loader waves share SIMDs 0 and 1 with MFMA waves 0 and 1 and got ONE
// instruction issued per ~8 MFMAs (a six-copy burst that takes 120 cycles on a quiet SIMD took 1750, s_setprio makes no
// difference), so every tap ended with the MFMA waves parked at the barrier while the loaders finished issuing: tap time =
// MFMA block + loader tail, 2900 cycles instead of 1750.  Here the four MFMA waves issue the copies themselves, a few per 12-MFMA
// block, in the issue shadow of their own MFMAs: wave w moves a quarter of every weight stage (3 x 1 KiB) and every fourth halo
// copy, waits for ITS copies of stage T + 2 (counted vmcnt) before barrier T, and the barrier makes all four quarters visible.
// Four waves per workgroup = one per SIMD = the whole 512-register file for each.
template <int NBW>       // halo copy slots per wave per tap (taps 2..8 of a chunk carry the next chunk's halo)
__global__ __launch_bounds__(256) void igemm_bf6m_kernel(Bf6Params p) {
    constexpr int RING = 4, ASTAGE16 = 768, PER = 3 + NBW;
    constexpr int SMEM16 = 9728;
    __shared__ __attribute__((aligned(16))) u32x4 smem[SMEM16];
    u32x4* const As = smem;
    u32x4* const Bs = smem + RING * ASTAGE16;
    const int region16 = p.halo16 + p.aux16, bbuf16 = 6 * region16;

    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int t = blockIdx.x;
    if (p.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);
    const int ptile = t / p.mblks, mblk = t - ptile * p.mblks;
    const int n = ptile / p.tiles_per_img, y0 = (ptile - n * p.tiles_per_img) * p.RT;
    const int nchunks = p.Cred >> 4, nstage = nchunks * 9;
    const int CG = p.Cred >> 3;

    // ---- this wave's share of the copies ----
    const int hi = (p.halo16 + 63) >> 6, ai = (p.aux16 + 63) >> 6, ipr = hi + ai, ncopies = 6 * ipr;
    const u32x4* asrc = p.wp + (size_t)mblk * 768 + wid * 192 + lane;        // + stage * mblks * 768
    const size_t wstage = (size_t)p.mblks * 768;
    const u32x4* const bsrc0 = p.planes + (size_t)n * CG * p.HpWs + lane;
    const int halo_off = y0 * p.Ws, aux_off = (p.H + 2) * p.Ws;
    int bi = ncopies, breg = 0, bin = 0;           // halo stream of the chunk being fetched: linear copy index, region, copy in region
    int bchunk = 0;                                // ... and that chunk
    // one halo copy (or, when this wave's share of the chunk is done, a re-issue of its first weight copy: the per-tap copy count
    // stays constant, which is what the counted waits rely on)
#define BF6M_COPY_B(aslot_)                                                                                             \
    {                                                                                                                   \
        if (bi < ncopies) {                                                                                             \
            const bool aux_ = bin >= hi;                                                                                \
            const int j_ = aux_ ? bin - hi : bin;                                                                       \
            const int pl_ = breg >> 1, kg_ = breg & 1;                                                                  \
            if (j_ * 64 + lane < (aux_ ? p.aux16 : p.halo16))                                                           \
                glds16(bsrc0 + (size_t)pl_ * p.plane16 + (size_t)(2 * bchunk + kg_) * p.HpWs + (aux_ ? aux_off : halo_off) + j_ * 64, \
                       Bs + (bchunk & 1) * bbuf16 + breg * region16 + (aux_ ? p.halo16 : 0) + j_ * 64);                 \
            bi += 4;                                                                                                    \
            bin += 4;                                                                                                   \
            while (bin >= ipr) { bin -= ipr; ++breg; }                                                                  \
        } else {                                                                                                        \
            glds16(asrc - wstage, As + (aslot_) * ASTAGE16 + wid * 192);                                                \
        }                                                                                                               \
    }
#define BF6M_HALO_BEGIN(chunk_)                                                                                         \
    {                                                                                                                   \
        bchunk = (chunk_);                                                                                              \
        bi = bchunk < nchunks ? wid : ncopies;                                                                          \
        breg = 0;                                                                                                       \
        bin = wid;                                                                                                      \
        while (bin >= ipr) { bin -= ipr; ++breg; }                                                                      \
    }
    // weight copy q (0..2) of the stage being issued into ring slot aslot_
#define BF6M_COPY_A(aslot_, q_) glds16(asrc + (q_) * 64, As + (aslot_) * ASTAGE16 + wid * 192 + (q_) * 64);
    // a whole stage at once (prologue): ti_ = its tap
#define BF6M_ISSUE_STAGE(aslot_, ci_, ti_)                                                                              \
    {                                                                                                                   \
        BF6M_COPY_A(aslot_, 0) BF6M_COPY_A(aslot_, 1) BF6M_COPY_A(aslot_, 2)                                            \
        asrc += wstage;                                                                                                 \
        if ((ti_) == 2) BF6M_HALO_BEGIN((ci_) + 1)                                                                      \
        _Pragma("unroll") for (int q = 0; q < NBW; ++q) BF6M_COPY_B(aslot_)                                             \
    }
#define BF6M_WAIT_OWN(n_)      /* at most n_ stages' worth of this wave's copies still in flight */                     \
    {                                                                                                                   \
        if ((n_) <= 0) wait_vmem();                                                                                     \
        else __builtin_amdgcn_s_waitcnt(0x0F70 | (PER & 15) | ((PER >> 4) << 14));                                      \
    }
    static_assert(PER < 16, "one stage of copies per wave");

    // ---- MFMA side ----
    const int l31 = lane & 31, lhi = lane >> 5;
    int row[2], col[2];
    bool top[2], bot[2], lft[2], rgt[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int px = 64 * wid + 32 * nt + l31;
        row[nt] = px >> p.wshift;
        col[nt] = px & (p.W - 1);
        const int y = y0 + row[nt];
        top[nt] = p.fold && y == 1;
        bot[nt] = p.fold && y == p.H - 2;
        lft[nt] = p.fold && col[nt] == 1;
        rgt[nt] = p.fold && col[nt] == p.W - 2;
    }
    const int auxoff = p.halo16;
    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
#define BF6M_READ_B(dst_, nt_, hb_, r_, sx_)                                                                            \
    {                                                                                                                   \
        const u32x4* const Bb_ = Bs + (hb_) * bbuf16 + lhi * region16;                                                  \
        int ra_ = (row[nt_] + (r_)) * p.Ws;                                                                             \
        if ((r_) == 2) ra_ = top[nt_] ? auxoff : ra_;                                                                   \
        if ((r_) == 0) ra_ = bot[nt_] ? auxoff + p.Ws : ra_;                                                            \
        int sl_ = col[nt_] + (sx_);                                                                                     \
        if ((sx_) == 2) sl_ = lft[nt_] ? p.W + 2 : sl_;                                                                 \
        if ((sx_) == 0) sl_ = rgt[nt_] ? p.W + 3 : sl_;                                                                 \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) dst_[pl] = Bb_[pl * 2 * region16 + ra_ + sl_];                 \
    }
#define BF6M_READ_A(dst_, slot_, mt_)                                                                                   \
    {                                                                                                                   \
        const u32x4* const Ab_ = As + (slot_) * ASTAGE16 + lhi * 128 + l31 + (mt_) * 32;                                \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) dst_[pl] = Ab_[pl * 256];                                      \
    }
#define BF6M_MFMA1(mt_, nt_, A_, B_, q_)                                                                               \
    acc[mt_][nt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_[PA[q_]]),                     \
                                                            __builtin_bit_cast(bf16x8, B_[PB[q_]]), acc[mt_][nt_], 0, 0, 0);
#define BF6M_MFMA(mt_, A_)                                                                                              \
    _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                                                     \
        BF6M_MFMA1(mt_, 0, A_, b0, q)                                                                                   \
        BF6M_MFMA1(mt_, 1, A_, b1, q)                                                                                   \
    }
#define BF6M_MFMA_NT(mt_, nt_, A_, B_) _Pragma("unroll") for (int q = 0; q < 6; ++q) BF6M_MFMA1(mt_, nt_, A_, B_, q)
#define BF6M_PIN() __builtin_amdgcn_sched_barrier(0);
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
    u32x4 a0[3], a1[3], a2[3], a3[3], b0[3], b1[3];

    // ---- prologue: this wave's quarter of the first halo and of stages 0, 1, 2 ----
    BF6M_HALO_BEGIN(0)
    while (bi < ncopies) BF6M_COPY_B(0)
    BF6M_ISSUE_STAGE(0, 0, 0)
    BF6M_ISSUE_STAGE(1, 0, 1)
    BF6M_ISSUE_STAGE(2, 0, 2)
    BF6M_WAIT_OWN(1)                                  // everything but stage 2
    __builtin_amdgcn_s_barrier();                     // B_-1: stages 0, 1 and the first halo are in LDS, all four quarters
    BF6M_READ_B(b0, 0, 0, 0, 0)
    BF6M_READ_B(b1, 1, 0, 0, 0)
    BF6M_READ_A(a0, 0, 0)
    int slot = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int hb = chunk & 1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ntap = tap == 8 ? 0 : tap + 1;
            const int nr = ntap / 3, nsx = ntap - 3 * (ntap / 3);
            const int nhb = tap == 8 ? hb ^ 1 : hb;
            const int nslot = (slot + 1) & (RING - 1);
            // the stage issued during this tap: T + 3, into the slot stage T - 1 vacated at the last barrier
            const int iti = tap + 3 >= 9 ? tap + 3 - 9 : tap + 3;
            const int ici = tap + 3 >= 9 ? chunk + 1 : chunk;
            const int islot = (slot + 3) & (RING - 1);
            const bool issue = ici < nchunks;
            u32x4 m1[3];
#ifdef NEMAR_TIMELINE
            const int T_ = chunk * 9 + tap;
            const bool mprobe = p.tl != nullptr && blockIdx.x == 0 && lane == 0 && T_ >= 40 && T_ < 48;
#define BF6M_STAMP(i_) if (mprobe) p.tl[(wid * 8 + (T_ - 40)) * 8 + (i_)] = clock64();
#else
#define BF6M_STAMP(i_)
#endif
            BF6M_STAMP(0)
            BF6M_PIN()
            BF6M_READ_A(a1, slot, 1)
            BF6M_PIN()
            BF6M_MFMA(0, a0)
            BF6M_PIN()
            BF6M_STAMP(1)
            if (issue) { BF6M_COPY_A(islot, 0) BF6M_COPY_A(islot, 1) }
            BF6M_STAMP(2)
            BF6M_READ_A(a2, slot, 2)
            BF6M_PIN()
            BF6M_MFMA(1, a1)
            BF6M_PIN()
            if (issue) {
                BF6M_COPY_A(islot, 2)
                asrc += wstage;
                if (iti == 2) BF6M_HALO_BEGIN(ici + 1)
#pragma unroll
                for (int q = 0; q < NBW; ++q) BF6M_COPY_B(islot)
            }
            BF6M_STAMP(3)
            BF6M_READ_A(a3, slot, 3)
            BF6M_READ_B(m1, 1, nhb, nr, nsx)
            BF6M_PIN()
            BF6M_MFMA(2, a2)
            BF6M_PIN()
            BF6M_READ_A(a0, nslot, 0)
            BF6M_PIN()
            BF6M_MFMA_NT(3, 0, a3, b0)
            BF6M_PIN()
            BF6M_READ_B(b0, 0, nhb, nr, nsx)
            BF6M_PIN()
            BF6M_MFMA_NT(3, 1, a3, b1)
            BF6M_PIN()
            BF6M_STAMP(4)
            BF6M_WAIT_OWN(issue ? 1 : 0)              // this wave's copies of stage T + 2 have landed
            BF6M_STAMP(5)
            __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): done reading stage T (tile 0 of T + 1 is in registers)
            BF6M_STAMP(6)
            __builtin_amdgcn_s_barrier();             // B_T: stage T + 2 complete for everyone; slot of stage T is free
            BF6M_STAMP(7)
#undef BF6M_STAMP
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b1[pl] = m1[pl];
            slot = nslot;
        }
    }
#undef BF6M_PIN
#undef BF6M_MFMA_NT
#undef BF6M_MFMA
#undef BF6M_MFMA1
#undef BF6M_READ_A
#undef BF6M_READ_B
#undef BF6M_WAIT_OWN
#undef BF6M_ISSUE_STAGE
#undef BF6M_COPY_A
#undef BF6M_HALO_BEGIN
#undef BF6M_COPY_B

    const size_t HW = (size_t)p.H * p.W;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float* const d0 = p.dst + (size_t)n * p.M * HW + (size_t)(y0 + row[nt]) * p.W + col[nt];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mblk * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float v = acc[mt][nt][r];
                if (p.bias) v += p.bias[m];
                d0[(size_t)m * HW] = v;
            }
        }
    }
}

// ---- fourth generation: the third, rebuilt around what its timeline showed ------------------------------------------------
// (tools/timeline_bf6.py, gpurun_out/timeline_bf6_v3.txt: a 12-MFMA block takes 384 cycles of matrix pipe but 520-700 in the
// kernel, the two halo copies of a tap with their address arithmetic and branches 500, and the six-deep dependent MFMA chains of
// the last block 40+ cycles per MFMA.)  Same data flow and protocol, but
//   * every fragment of tap T + 1 is read during tap T into a second register set (one wave per SIMD = 512 registers), so all 48
//     MFMAs of a tap are independent of this tap's LDS reads and rotate over the eight accumulators;
//   * the tap body is branch-free: the halo copies of a chunk are described once per wave (source offset, LDS offset, lane limit)
//     in registers indexed by the unrolled tap position, the tail re-issues harmless copies instead of skipping, and the per-tap
//     copy count is a compile-time function of the tap, so the counted vmcnt waits need no filler copies outside the window;
//   * with A(T + 1) fully in registers at barrier T - 1... the weight stage for tap T + 4 is issued during tap T (two taps of lead);
//   * sched_group_barrier prescribes the interleaving (one LDS read or one copy and a couple of scalar/vector ALU instructions per
//     MFMA), instead of 12-MFMA blocks separated by everything else.
// Two chunks (18 taps) per loop iteration, so that the register-set parity is a compile-time constant of the tap position.
template <int NBW>       // halo copy slots per wave per tap on taps 3..8 of a chunk (they carry the next chunk's halo)
__global__ __launch_bounds__(256) void igemm_bf6x_kernel(Bf6Params p) {
    constexpr int RING = 4, ASTAGE16 = 768, KB = 6 * NBW;
    constexpr int SMEM16 = 9728;
    __shared__ __attribute__((aligned(16))) u32x4 smem[SMEM16];
    u32x4* const As = smem;
    u32x4* const Bs = smem + RING * ASTAGE16;
    const int region16 = p.halo16 + p.aux16, bbuf16 = 6 * region16;

    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int t = blockIdx.x;
    if (p.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);
    const int ptile = t / p.mblks, mblk = t - ptile * p.mblks;
    const int n = ptile / p.tiles_per_img, y0 = (ptile - n * p.tiles_per_img) * p.RT;
    const int nchunks = p.Cred >> 4, nstage = nchunks * 9;
    const int CG = p.Cred >> 3;

    // ---- this wave's copies: weights = words [192 wid, 192 wid + 192) of every 768-word stage; halo = every fourth 1 KiB copy ----
    const int hi = (p.halo16 + 63) >> 6, ai = (p.aux16 + 63) >> 6, ipr = hi + ai, ncopies = 6 * ipr;
    const size_t wstage = (size_t)p.mblks * 768;
    const u32x4* const wsrc0 = p.wp + (size_t)mblk * 768 + wid * 192;
    const u32x4* const bsrc0 = p.planes + (size_t)n * CG * p.HpWs;
    const int halo_off = y0 * p.Ws, aux_off = (p.H + 2) * p.Ws;
    unsigned goff[KB], blds[KB];      // per halo copy of this wave: source word offset from the chunk's images (+ lane), LDS word offset
    int blim[KB];                     // ... and the number of lanes that take part
    int share = 0;
    {
        int breg = 0, bin = wid;
        while (bin >= ipr) { bin -= ipr; ++breg; }
        unsigned g = 0, l = 0;
        int lim = 0;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            if (wid + 4 * k < ncopies) {
                const bool aux = bin >= hi;
                const int j = aux ? bin - hi : bin;
                const int pl = breg >> 1, kg = breg & 1;
                g = (unsigned)((size_t)pl * p.plane16 + (size_t)kg * p.HpWs + (aux ? aux_off : halo_off) + j * 64);
                l = (unsigned)(breg * region16 + (aux ? p.halo16 : 0) + j * 64);
                lim = (aux ? p.aux16 : p.halo16) - j * 64;
                share = k + 1;
                bin += 4;
                while (bin >= ipr) { bin -= ipr; ++breg; }
            }
            goff[k] = g + lane;       // (slots beyond the share repeat the last copy)
            blds[k] = l;
            blim[k] = lim;
        }
    }
    // copies of one stage: COUNT(ti) = 3 + (ti >= 3 ? NBW : 0) wave-instructions, in this order
#define BF6X_COPIES(stage_, ti_, ci_)                                                                                   \
    {                                                                                                                   \
        const int st_ = min((stage_), nstage - 1);                   /* tail: harmless re-copies of the last stage */   \
        const u32x4* const a_ = wsrc0 + (size_t)st_ * wstage + lane;                                                    \
        u32x4* const ad_ = As + ((stage_) & (RING - 1)) * ASTAGE16 + wid * 192;                                         \
        glds16(a_, ad_);                                                                                                \
        glds16(a_ + 64, ad_ + 64);                                                                                      \
        glds16(a_ + 128, ad_ + 128);                                                                                    \
        if ((ti_) >= 3) {                                                                                               \
            const int hc_ = min((ci_) + 1, nchunks - 1);             /* chunk whose halo travels with this stage */     \
            const u32x4* const bb_ = bsrc0 + (size_t)(2 * hc_) * p.HpWs;                                                \
            u32x4* const bd_ = Bs + (((ci_) + 1) & 1) * bbuf16;                                                         \
            _Pragma("unroll") for (int q = 0; q < NBW; ++q) {                                                           \
                const int k_ = ((ti_) - 3) * NBW + q;                                                                   \
                if (lane < blim[k_]) glds16(bb_ + goff[k_], bd_ + blds[k_]);                                            \
            }                                                                                                           \
        }                                                                                                               \
    }
#define BF6X_COUNT(ti_) (3 + ((ti_) >= 3 ? NBW : 0))
#define BF6X_VMCNT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14));

    // ---- MFMA side ----
    const int l31 = lane & 31, lhi = lane >> 5;
    int row[2], col[2];
    bool top[2], bot[2], lft[2], rgt[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int px = 64 * wid + 32 * nt + l31;
        row[nt] = px >> p.wshift;
        col[nt] = px & (p.W - 1);
        const int y = y0 + row[nt];
        top[nt] = p.fold && y == 1;
        bot[nt] = p.fold && y == p.H - 2;
        lft[nt] = p.fold && col[nt] == 1;
        rgt[nt] = p.fold && col[nt] == p.W - 2;
    }
    const int auxoff = p.halo16;
    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    u32x4 af[2][4][3], bf[2][2][3];
    // all fragments of tap (r_, sx_) of the chunk in halo buffer hb_, weights in ring slot slot_, into register set set_
#define BF6X_READ(set_, slot_, hb_, r_, sx_)                                                                            \
    {                                                                                                                   \
        const u32x4* const Bb_ = Bs + (hb_) * bbuf16 + lhi * region16;                                                  \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                              \
            int ra_ = (row[nt] + (r_) + z) * p.Ws;       /* z: see the loop head */                                     \
            if ((r_) == 2) ra_ = top[nt] ? auxoff : ra_;                                                                \
            if ((r_) == 0) ra_ = bot[nt] ? auxoff + p.Ws : ra_;                                                         \
            int sl_ = col[nt] + (sx_);                                                                                  \
            if ((sx_) == 2) sl_ = lft[nt] ? p.W + 2 : sl_;                                                              \
            if ((sx_) == 0) sl_ = rgt[nt] ? p.W + 3 : sl_;                                                              \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) bf[set_][nt][pl] = Bb_[pl * 2 * region16 + ra_ + sl_];     \
        }                                                                                                               \
        const u32x4* const Ab_ = As + (slot_) * ASTAGE16 + lhi * 128 + l31;                                             \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                                \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) af[set_][mt][pl] = Ab_[pl * 256 + mt * 32];                \
    }
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};

    // ---- prologue: this wave's share of the first halo and of stages 0..3 ----
    {
        const u32x4* const bb = bsrc0;
#pragma unroll
        for (int k = 0; k < KB; ++k)
            if (k < share && lane < blim[k]) glds16(bb + goff[k], Bs + blds[k]);
    }
    BF6X_COPIES(0, 0, 0)
    BF6X_COPIES(1, 1, 0)
    BF6X_COPIES(2, 2, 0)
    BF6X_COPIES(3, 3, 0)
    BF6X_VMCNT(6 + NBW)                               // (stages 2 and 3 may be in flight) the first halo and stages 0, 1 have landed
    __builtin_amdgcn_s_barrier();                     // ... for all four waves
    int z = 0;
    BF6X_READ(0, 0, 0, 0, 0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                     // every wave holds the fragments of tap 0: slot 0 may be refilled (tap 0 does)
    for (int chunk0 = 0; chunk0 < nchunks; chunk0 += 2) {
        // an opaque zero per iteration: keeps hipcc from hoisting the 18 taps' fragment addresses out of the loop (that costs
        // more registers than the file has: 512 + spills) — they are three VALU instructions each to recompute
#ifndef NEMAR_HOST_EMULATION
        asm volatile("s_mov_b32 %0, 0" : "=s"(z));
#endif
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int chunk = chunk0 + half;
            if (half == 1 && chunk >= nchunks) break;
            const int hb = half;                      // = chunk & 1
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int cur = (tap + half) & 1, nxt = cur ^ 1;          // = T & 1 with T = 9 chunk + tap
                const int T = chunk * 9 + tap;
                // tap T + 1: every fragment into the other register set (after the last tap: a harmless read of stale LDS);
                // stage T + 4 into the slot of stage T (read during tap T - 1): two full taps ahead of its first use.
                // 23 + NBW slots of [<= 1 memory instruction + its scalar / vector arithmetic][2 MFMAs], pinned: hipcc otherwise
                // clumps the reads and copies, and every clump longer than an MFMA's 32-cycle shadow idles the matrix pipe
#ifdef NEMAR_TIMELINE
                const bool xprobe = p.tl != nullptr && blockIdx.x == 0 && lane == 0 && T >= 40 && T < 48;
#define BF6X_STAMP(i_) if (xprobe) p.tl[(wid * 8 + (T - 40)) * 8 + (i_)] = clock64();
#else
#define BF6X_STAMP(i_)
#endif
                BF6X_STAMP(0)
                const int ntap = tap == 8 ? 0 : tap + 1;
                const int nhb = tap == 8 ? hb ^ 1 : hb;
                const int nr = ntap / 3, nsx = ntap % 3;
                const int iti = tap + 4 >= 9 ? tap + 4 - 9 : tap + 4;
                const int ici = tap + 4 >= 9 ? chunk + 1 : chunk;
#define BF6X_MFMAS(beg_, n_)                                                                                            \
                _Pragma("unroll") for (int m_ = (beg_); m_ < (beg_) + (n_) && m_ < 48; ++m_) {                           \
                    const int q_ = m_ >> 3, mt_ = (m_ & 7) >> 1, nt_ = m_ & 1;                                          \
                    acc[mt_][nt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[cur][mt_][PA[q_]]), \
                                                                            __builtin_bit_cast(bf16x8, bf[cur][nt_][PB[q_]]), \
                                                                            acc[mt_][nt_], 0, 0, 0);                    \
                }                                                                                                       \
                __builtin_amdgcn_sched_barrier(0);
                int baddr[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    int ra_ = (row[nt] + nr + z) * p.Ws;
                    if (nr == 2) ra_ = top[nt] ? auxoff : ra_;
                    if (nr == 0) ra_ = bot[nt] ? auxoff + p.Ws : ra_;
                    int sl_ = col[nt] + nsx;
                    if (nsx == 2) sl_ = lft[nt] ? p.W + 2 : sl_;
                    if (nsx == 0) sl_ = rgt[nt] ? p.W + 3 : sl_;
                    baddr[nt] = ra_ + sl_;
                }
                const u32x4* const Bn_ = Bs + nhb * bbuf16 + lhi * region16;
                const u32x4* const An_ = As + ((T + 1) & (RING - 1)) * ASTAGE16 + lhi * 128 + l31;
                BF6X_MFMAS(0, 2)
#pragma unroll
                for (int i = 0; i < 6; ++i) {                 // slots 1..6: the B fragments
                    bf[nxt][i / 3][i % 3] = Bn_[(i % 3) * 2 * region16 + baddr[i / 3]];
                    BF6X_MFMAS(2 + 2 * i, 2)
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) {                // slots 7..18: the A fragments
                    af[nxt][i / 3][i % 3] = An_[(i % 3) * 256 + (i / 3) * 32];
                    BF6X_MFMAS(14 + 2 * i, 2)
                }
                {                                             // slots 19..: the copies of stage T + 4
                    const int st_ = min(T + 4, nstage - 1);   // (tail: harmless re-copies of the last stage)
                    const u32x4* const a_ = wsrc0 + (size_t)st_ * wstage + lane;
                    u32x4* const ad_ = As + (T & (RING - 1)) * ASTAGE16 + wid * 192;
                    glds16(a_, ad_);
                    BF6X_MFMAS(38, 2)
                    glds16(a_ + 64, ad_ + 64);
                    BF6X_MFMAS(40, 2)
                    glds16(a_ + 128, ad_ + 128);
                    BF6X_MFMAS(42, 2)
                    if (iti >= 3) {
                        const int hc_ = min(ici + 1, nchunks - 1);
                        const u32x4* const bb_ = bsrc0 + (size_t)(2 * hc_) * p.HpWs;
                        u32x4* const bd_ = Bs + ((ici + 1) & 1) * bbuf16;
#pragma unroll
                        for (int q = 0; q < NBW; ++q) {
                            const int k_ = (iti - 3) * NBW + q;
                            if (lane < blim[k_]) glds16(bb_ + goff[k_], bd_ + blds[k_]);
                            BF6X_MFMAS(44 + 2 * q, q == NBW - 1 ? 48 : 2)
                        }
                    } else {
                        BF6X_MFMAS(44, 48)
                    }
                }
#undef BF6X_MFMAS
                // this wave's copies of stage T + 2 have landed; those of T + 3 and T + 4 may still be in flight
                const int t3 = tap + 3 >= 9 ? tap + 3 - 9 : tap + 3;
                const int nfl = BF6X_COUNT(t3) + BF6X_COUNT(iti);     // compile-time after unrolling: one of three values
                BF6X_STAMP(1)
                if (nfl == 6) BF6X_VMCNT(6)
                else if (nfl == 6 + NBW) BF6X_VMCNT(6 + NBW)
                else BF6X_VMCNT(6 + 2 * NBW)
                BF6X_STAMP(2)
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): every fragment of tap T + 1 is in registers
                BF6X_STAMP(3)
                __builtin_amdgcn_s_barrier();         // B_T: stage T + 2 complete for all waves; slot of stage T + 1 is free
                BF6X_STAMP(4)
#undef BF6X_STAMP
            }
        }
    }
    wait_vmem();                                      // (the tail's re-copies)
#undef BF6X_READ
#undef BF6X_VMCNT
#undef BF6X_COUNT
#undef BF6X_COPIES

    const size_t HW = (size_t)p.H * p.W;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float* const d0 = p.dst + (size_t)n * p.M * HW + (size_t)(y0 + row[nt]) * p.W + col[nt];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mblk * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float v = acc[mt][nt][r];
                if (p.bias) v += p.bias[m];
                d0[(size_t)m * HW] = v;
            }
        }
    }
}

int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

}  // namespace

bool nemar_bf6_eligible(int N, int H, int W, int M, int Cred, int R, int S, int stride, int pad, int mode) {
    if (R != 3 || S != 3 || stride != 1 || pad != 1) return false;
    if (M % 128 != 0 || Cred % 16 != 0 || M <= 0 || Cred <= 0) return false;
    if (!(W == 32 || W == 64 || W == 128)) return false;
    const int RT = 256 / W;
    if (H % RT != 0 || H < 4) return false;
    if (W == 128 && mode == BF6_DGRAD_REFLECT) return false;      // halo + folded rows of two buffers exceed the LDS
    if ((long long)N * Cred * (H + 4) * (W + 4) >= (1ll << 31)) return false;
    return true;
}

size_t nemar_bf6_scratch_bytes(int N, int Cred, int H, int W) {
    return (size_t)3 * N * (Cred / 8) * (H + 4) * (W + 4) * 16 + 16384;     // + slack for whole-KiB halo reads
}

size_t nemar_bf6_pack_bytes(int M, int Cred) { return (size_t)(Cred / 16) * 9 * (M / 128) * 768 * 16 + 16384; }

void nemar_bf6_pack(const float* w, void* packed, int K, int C, int dgrad, hipStream_t st) {
    const int M = dgrad ? C : K, Cred = dgrad ? K : C;
    const long long total = (long long)(Cred / 16) * 9 * (M / 128) * 256;
    hipLaunchKernelGGL(bf6_pack_kernel, dim3(nemar_stream_grid(total, 256)), dim3(256), 0, st, w, (u32x4*)packed, M, Cred, dgrad);
}

void nemar_bf6_conv(const float* src, const void* packed, const float* bias, float* dst, int N, int H, int W, int M, int Cred,
                    int mode, void* scratch, int xcd_map, int dbg, int variant, int rot, long long* tl, hipStream_t st) {
    const long long total = (long long)N * (Cred / 8) * (H + 4) * (W + 4);
    hipLaunchKernelGGL(split_planes_kernel, dim3(nemar_cdiv(total, 256)), dim3(256), 0, st, src, (u32x4*)scratch, N, Cred, H, W,
                       mode, total);
    Bf6Params p;
    p.planes = (const u32x4*)scratch;
    p.wp = (const u32x4*)packed;
    p.bias = bias;
    p.dst = dst;
    p.N = N; p.H = H; p.W = W; p.M = M; p.Cred = Cred;
    p.Ws = W + 4;
    p.HpWs = (H + 4) * (W + 4);
    p.wshift = ilog2(W);
    p.RT = 256 / W;
    p.tiles_per_img = H / p.RT;
    p.mblks = M / 128;
    p.halo_instr = nemar_cdiv((long long)(p.RT + 2) * p.Ws * 16, 1024);
    p.fold = mode == BF6_DGRAD_REFLECT;
    p.aux_instr = p.fold ? nemar_cdiv((long long)2 * p.Ws * 16, 1024) : 0;
    p.plane16 = total;
    p.dbg = dbg;
    p.tl = tl;
    p.rot = (rot > 1 && (Cred / 16) % rot == 0) ? rot : 0;
    p.halo16 = (p.RT + 2) * p.Ws;
    p.aux16 = p.fold ? 2 * p.Ws : 0;
    const int grid = N * p.tiles_per_img * p.mblks;
    p.xcd = (xcd_map && grid % 8 == 0 && (grid / 8) % p.mblks == 0) ? 1 : 0;
    const int region = p.halo_instr + p.aux_instr;
    const dim3 g(grid), b(384);
    if (variant == 3 && 6 * (p.halo16 + p.aux16) * 2 + 4 * 768 <= 9728) {
        const int ipr = nemar_cdiv(p.halo16, 64) + nemar_cdiv(p.aux16, 64);
        const int nbw = nemar_cdiv(nemar_cdiv(6 * ipr, 4), 6);
        if (nbw <= 2) hipLaunchKernelGGL((igemm_bf6x_kernel<2>), g, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((igemm_bf6x_kernel<3>), g, dim3(256), 0, st, p);
        return;
    }
    if (variant == 2 && 6 * (p.halo16 + p.aux16) * 2 + 4 * 768 <= 9728) {
        const int ipr = nemar_cdiv(p.halo16, 64) + nemar_cdiv(p.aux16, 64);
        const int nbw = nemar_cdiv(nemar_cdiv(6 * ipr, 4), 7);
        if (nbw <= 2) hipLaunchKernelGGL((igemm_bf6m_kernel<2>), g, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((igemm_bf6m_kernel<3>), g, dim3(256), 0, st, p);
        return;
    }
    if (variant == 1 && 6 * (p.halo16 + p.aux16) * 2 + 4 * 768 <= 9728) {
        const int ipr = nemar_cdiv(p.halo16, 64) + nemar_cdiv(p.aux16, 64);
        const int nb = nemar_cdiv(3 * ipr, 7);
        const int abl = (dbg >> 16) & 3;
        if (abl == 1 && nb == 4) hipLaunchKernelGGL((igemm_bf6p_kernel<4, 1>), g, b, 0, st, p);
        else if (abl == 2 && nb == 4) hipLaunchKernelGGL((igemm_bf6p_kernel<4, 2>), g, b, 0, st, p);
        else if (abl == 3 && nb == 4) hipLaunchKernelGGL((igemm_bf6p_kernel<4, 3>), g, b, 0, st, p);
        else if (nb <= 3) hipLaunchKernelGGL((igemm_bf6p_kernel<3>), g, b, 0, st, p);
        else if (nb == 4) hipLaunchKernelGGL((igemm_bf6p_kernel<4>), g, b, 0, st, p);
        else hipLaunchKernelGGL((igemm_bf6p_kernel<5>), g, b, 0, st, p);
        return;
    }
    if (region <= 6) hipLaunchKernelGGL((igemm_bf6_kernel<6, 4>), g, b, 0, st, p);
    else if (region == 7) hipLaunchKernelGGL((igemm_bf6_kernel<7, 4>), g, b, 0, st, p);
    else if (region == 8) hipLaunchKernelGGL((igemm_bf6_kernel<8, 4>), g, b, 0, st, p);
    else if (region == 9) hipLaunchKernelGGL((igemm_bf6_kernel<9, 4>), g, b, 0, st, p);
    else hipLaunchKernelGGL((igemm_bf6_kernel<10, 3>), g, b, 0, st, p);
}
