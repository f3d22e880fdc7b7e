/* nemar_hip_ab.h — what libnemar_hip_ab.so, the MEASUREMENT build of the operator library (the same sources compiled with
 * -DNEMAR_AB, nemar_amd/csrc/build.py), exports on top of nemar_hip.h: process-global switches that re-route shapes to non-default
 * kernels, lower the work thresholds of the routes (so that tests can drive small shapes through them) and hand a timeline buffer to
 * the instrumented kernels.  tools/ and the A/B tests link this library; the product (libnemar_hip.so) has none of these entry points,
 * its switches are compile-time constants at the defaults named below and the non-default kernels are not in it.
 * Not part of the operator contract of the reference's hot path — no reference call site corresponds to these. */
#ifndef NEMAR_HIP_AB_H
#define NEMAR_HIP_AB_H
#include "nemar_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Switches of the convolution family; defaults = measured best = what the product is compiled with.
 *   0  conv tile family for 128x128-capable shapes: 0 wave-specialised (default), 5 same without 16-byte B loads,
 *      6 one barrier per 32 reduction rows, 7 four loader waves, 4 first-generation wave-specialised, 1/2/3 generic
 *   1  extra dynamic LDS per workgroup (occupancy experiments)      2  ablation / experiment bit mask
 *   3  narrow (<= 4 channel) VALU kernels on/off                     4  weight gradient: 0 default, 1 first generation,
 *                                                                       2 wave-specialised without 16-byte source loads
 *   5  weight-gradient workgroup target (default 512)                6  grid-size threshold of the tile choice (384)
 *   7  force the wave-specialised channel tile (1, 2, 4 x 32)        8  3x3 reflect data gradient: border folded into the
 *                                                                       main launch (1, default) / separate ring launch (0)
 *   10 4-deep LDS ring for every 64x64 launch
 *   11 four loader waves for gathered B tiles (on)                   12 reduction splits in data gradients (on)
 *   14 fixed-order split reductions (1, default) / fp32 atomics in the weight + bias gradients (0)
 *   15 XCD-aware workgroup -> tile mapping of the wave-specialised kernels (1, default)
 *   16..19 loader / tile / ring-depth / narrow-kernel variants
 *   20 3x3 stride-1 layers with >= 128 output channels on the bf16 matrix pipe with three-way split operands (1, default;
 *      needs the scratch arena below; 0 = exact-fp32 MFMA kernels).  Packed weight images are per setting.
 *   21 operand split of those kernels: 4 fp16 x 3 (default), 3 bf16 x 6
 *   23 smallest layer (million multiply-adds, default 2000) that takes the split-16 route
 *   24 every other convolution with >= 5 output channels on the 16-bit matrix pipe with the operand split INSIDE the kernel
 *      (csrc/conv_s16g*.hip; 1, default; 0 = exact-fp32 MFMA kernels)      25 its work threshold (million multiply-adds, 30)
 *   26 the wide layers' weight gradient on the in-kernel-split kernel instead of wgrad_split16 (0, default: measured slower)
 *   27 widest channel tile of s16g_kernel (1, 2 = default, 4 x 32)                28 prefer pixel tiles that leave LDS for two workgroups
 *   29 weight gradients of the key-24 layers on s16g_wgrad_kernel (1)     30 stride-1 reflect 3x3 data gradients of those layers on
 *                                                                            the padded domain + reflect_fold_kernel (1)
 *   31 ablation bits of instnorm_planes_kernel (measurement only)
 *   32 3-slot weight ring of the wide-layer kernel for the unfolded 3x3 launches (0, default: no gain)
 *   33 the 7x7 stem / head layers on the 16-bit matrix pipe (csrc/conv_k7.hip; 1)
 *   34 wide weight gradient: one copy of the gy planes + shifted operands built in registers (1) / KS shifted copies in HBM (0)
 *   35 the wide data gradient's split pass also writes the weight gradient's gy planes (1)
 *   36 reduction-split forward layers keep a fused ReLU / LeakyReLU (activation in the sum pass; 1 since round 6c, 0 = such layers are not split)
 *   37 kernel families whose workgroups claim the whole CU's LDS (bit mask: 1 LDS-DMA forms of wgrad_split16_kernel (default), 2 igemm_split16_kernel,
 *      4 s16g_kernel — csrc/common.h)                                38 wide 3x3 weight gradient stages through registers (1, default) / by LDS-DMA (0)
 *   39 most reduction runs per tile of the wide-layer kernel (8; 1 = never split: small test shapes with the fused data-gradient epilogue)
 *   40 most 64-channel blocks one s16g_kernel workgroup runs on one converted halo (4; 1 = one workgroup per channel block, rounds 3-5)
 *   41 ... as long as the grid keeps this many workgroups (256; tests: 0)
 *   42 the four output-parity classes of a stride-2 data gradient / ConvTranspose2d in ONE s16g_kernel workgroup per tile, on one converted
 *      halo (0, default = one workgroup per (tile, class): the fused form measured 10-30 % slower, csrc/conv_s16g.hip; 1 = where the grid
 *      keeps min(192, key 41) workgroups; 2 = 1 and a four-class problem that cannot fuse leaves the route: tests)
 *   43 stride-1 reflect data gradients of tiny maps on the exact-fp32 kernels: one split launch over the padded domain + a sum-and-fold pass
 *      (1, default) / interior + border-ring launches with their sums and a gather (0) */
int nemar_tune(int key, int value);
int nemar_tune_ptr(void* timeline_buffer);   /* device buffer for per-stage cycle stamps (tools/timeline_*.py), NULL = off */
/* grad_input variant for A/B measurements: 0 (default) = gather + fixed point (needs the workspace), 1 = fp32 atomics through an
 * LDS tile per 16x64 output tile, 2 = global fp32 atomics (warp.hip has the numbers). */
int nemar_grid_sample_tune(int variant);

#ifdef __cplusplus
}
#endif
#endif
