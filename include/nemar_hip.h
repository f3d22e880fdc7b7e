/* nemar_hip.h — C ABI of libnemar_hip.so: the gfx950 (MI355X / CDNA4) operator library behind the NeMAR
 * training step, NEMARModel.optimize_parameters() (reference models/nemar_model.py:266-288).
 *
 * The reference has no FFI of its own: its operator boundary is the set of torch call sites listed in
 * SURVEY.md §2.2 (K1..K15).  Each entry point below replaces one of those call sites and cites it.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  Every pointer is DEVICE memory owned by the caller;
 *     tensors are contiguous NCHW float32.  The library never allocates, frees or synchronises.
 *   - every launch goes on `stream` (a hipStream_t passed as void*; NULL = the null stream), so the caller's
 *     stream ordering (torch's current stream under autograd) holds.  Re-entrant per stream: every input of an operator —
 *     including the side inputs of the wide-layer route (scratch arena, max words, planes: nemar_conv_extras) — travels with
 *     the call.  The only process-global state are the recorded weight-pack plans
 *     (the measurement switches of earlier versions live in a separate build of the library: nemar_hip_ab.h).
 *   - return value: 0 on success, negative on error (NEMAR_EINVAL bad shape/pointer/unsupported,
 *     NEMAR_ELAUNCH HIP launch error, NEMAR_EWORKSPACE workspace too small); nemar_last_error() returns the
 *     message of the calling thread's last failure.  Python glue raises on non-zero.
 *   - gradients are WRITTEN unless the matching `accumulate` flag is non-zero (then added in place).
 *   - scratch: ops that need it take (workspace, ws_bytes) and export nemar_<op>_workspace(...) -> bytes.
 */
#ifndef NEMAR_HIP_H
#define NEMAR_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEMAR_OK 0
#define NEMAR_EINVAL (-1)
#define NEMAR_ELAUNCH (-2)
#define NEMAR_EWORKSPACE (-3)

/* library */
int nemar_version(void);              /* major*10000 + minor*100 + patch; 602 = this header (0.4.x exported nemar_tune*) */
const char* nemar_last_error(void);   /* thread-local message of the last failing call */

/* ---- K9/K10/K11: sampling-grid generation fused into bilinear grid_sample ------------------------------
 * F.grid_sample(img, grid, 'bilinear', 'zeros', align_corners=False)
 *     reference models/stn/unet_stn.py:173-174, models/stn/affine_stn.py:129-130
 * grid_mode selects how the (never materialised) grid is synthesised from grid_src:
 *   NEMAR_GRID_EXPLICIT  grid_src = grid [N,Ho,Wo,2] (x,y), normalised coordinates
 *   NEMAR_GRID_UNET      grid_src = offsets [N,2,Ho,Wo] planar; grid = linspace(-1,1) identity + offsets
 *                        (reference models/stn/unet_stn.py:121-129,167; channel 0 = x)
 *   NEMAR_GRID_AFFINE    grid_src = dtheta [N,6]; theta = dtheta + [1,0,0,0,1,0];
 *                        grid = F.affine_grid(theta, align_corners=False) (models/stn/affine_stn.py:122,128)
 * in [N,C,H,W] -> out [N,C,Ho,Wo]. */
#define NEMAR_GRID_EXPLICIT 0
#define NEMAR_GRID_UNET 1
#define NEMAR_GRID_AFFINE 2
int nemar_grid_sample_fwd(const float* in, const float* grid_src, int grid_mode, float* out,
                          int N, int C, int H, int W, int Ho, int Wo, void* stream);
/* gin [N,C,H,W] may be NULL (source is data, e.g. real_A).  ggrid has the layout of grid_src
 * (EXPLICIT [N,Ho,Wo,2]; UNET [N,2,Ho,Wo]; AFFINE [N,6]).
 * With a workspace (and same-size input/output, C <= 4: the training path) grad_input is computed WITHOUT floating-point
 * atomics and is bitwise reproducible: a gather per 64x16 destination tile for pixels that sample within 3 texels of their
 * own position, 64-bit fixed-point atomics (40 bits below max |gout|) for the others; the affine grid gradient is summed
 * in a fixed order.  Workspace contract: nemar_grid_sample_bwd_workspace() bytes, of which the leading
 * nemar_grid_sample_bwd_zeroed_bytes() must be ZERO on entry of the first call and are left zero by every call (the
 * fixed-point accumulator); the rest is scratch.  workspace == NULL: fp32-atomic scatter (any shape; not reproducible). */
size_t nemar_grid_sample_bwd_workspace(int N, int C, int H, int W);
size_t nemar_grid_sample_bwd_zeroed_bytes(int N, int C, int H, int W);
int nemar_grid_sample_bwd(const float* in, const float* grid_src, int grid_mode, const float* gout,
                          float* gin, int accum_gin, float* ggrid, int accum_ggrid,
                          int N, int C, int H, int W, int Ho, int Wo, void* workspace, size_t ws_bytes, void* stream);

/* ---- K12: deformation smoothness / bilateral regulariser -------------------------------------------------
 * smoothness_loss(deformation, img, alpha)   reference models/stn/stn_losses.py:4-30,
 * called from UnetSTN._calculate_regularization_term, models/stn/unet_stn.py:179-201.
 * d [N,2,H,W]; img [N,Ci,H,W] or NULL (alpha<=0 or NULL => unweighted).
 * fwd: loss[0] = (accumulate ? loss[0] : 0) + factor * smoothness(d, img, alpha)
 * bwd: gd (+)= gscale[0] * factor * d smoothness / d d     (gscale: device scalar, the upstream gradient) */
size_t nemar_smoothness_workspace(int N, int H, int W);
int nemar_smoothness_fwd(const float* d, const float* img, int Ci, float alpha, float factor,
                         float* loss, int accumulate, void* workspace, size_t ws_bytes,
                         int N, int H, int W, void* stream);
int nemar_smoothness_bwd(const float* d, const float* img, int Ci, float alpha, const float* gscale,
                         float factor, float* gd, int accumulate, int N, int H, int W, void* stream);

/* ---- K1/K2/K4/K8: convolution family on fp32 MFMA (exact fp32) ---------------------------------------------
 * nn.Conv2d / nn.ConvTranspose2d with the ReflectionPad2d and torch.cat feeding them folded in —
 *     reference models/networks.py:349-377 (ResnetGenerator), :418-439 (ResnetBlock), :576-597
 *     (NLayerDiscriminator); models/stn/layers.py:85 (Conv); models/stn/unet_stn.py:80,97 (cat);
 *     models/stn/affine_stn.py:69-72 (nn.Linear == 1x1 conv on a 1x1 image).
 * x = channel concat of x0 [N,C0,H,W] and x1 [N,C1,H,W] (x1 NULL/C1=0 for a single source), never materialised.
 * w [K,C0+C1,R,S] (torch layout), bias [K] or NULL.  pad_mode: NEMAR_PAD_ZERO | NEMAR_PAD_REFLECT
 * (reflect == nn.ReflectionPad2d(pad) followed by an unpadded conv).  act is applied in the epilogue:
 * NEMAR_ACT_NONE | RELU | LRELU(slope) | TANH.  y [N,K,OH,OW], OH = (H + 2 pad - R) / stride + 1.
 * workspace: packed weights, then the slabs of a split reduction (tiny, deep layers; summed in a fixed order, so the forward
 * pass stays bitwise reproducible) — nemar_conv2d_fwd_workspace bytes.  prepacked != 0: the workspace already holds this
 * weight tensor's packed image from an earlier call with the same (w values, shape, stride, pad): the pack launch is
 * skipped (the caller caches one workspace per weight tensor and invalidates it when the optimizer steps). */
#define NEMAR_PAD_ZERO 0
#define NEMAR_PAD_REFLECT 1
#define NEMAR_ACT_NONE 0
#define NEMAR_ACT_RELU 1
#define NEMAR_ACT_LRELU 2
#define NEMAR_ACT_TANH 3
size_t nemar_conv2d_fwd_workspace(int N, int H, int W, int K, int C, int R, int S, int stride, int pad);
int nemar_conv2d_fwd(const float* x0, int C0, const float* x1, int C1, const float* w, const float* bias,
                     float* y, int N, int H, int W, int K, int R, int S, int stride, int pad, int pad_mode,
                     int act, float slope, void* workspace, size_t ws_bytes, int prepacked, void* stream);
/* Data gradient: gy [N,K,OH,OW] -> gx0 [N,C0,H,W] | gx1 [N,C1,H,W] (gx0 NULL: its channels are skipped, e.g. the
 * real_A half of the discriminator input).  With bias/act it is ALSO the forward of
 * nn.ConvTranspose2d(K -> C, k, stride, pad, output_padding) whose weight is w [K,C,R,S]
 * (reference models/networks.py:369-372): pass gy := input, (H,W) := the transposed conv's output size. */
size_t nemar_conv2d_bwd_data_workspace(int N, int C, int H, int W, int K, int R, int S, int stride, int pad,
                                       int pad_mode);
int nemar_conv2d_bwd_data(const float* gy, const float* w, const float* bias, int act, float slope,
                          float* gx0, int C0, float* gx1, int C1, int N, int H, int W, int K, int OH, int OW,
                          int R, int S, int stride, int pad, int pad_mode, void* workspace, size_t ws_bytes,
                          int prepacked, void* stream);
/* Weight gradient, ACCUMULATED into gw [K,C0+C1,R,S] and, when gb != NULL, the bias gradient ACCUMULATED into
 * gb [K] in the same pass (the caller zero-fills once per optimizer step; the translation net receives two passes
 * per step) — autograd's conv backward + AccumulateGrad under loss.backward(), reference models/nemar_model.py:223,260.
 * The pixel reduction is split across workgroups; every split stores its partial result to its own slab of `workspace`
 * and a second launch adds the slabs in split order.  Split data gradients (few-tile deep layers) do the same, so the
 * whole backward pass of the conv family is BITWISE REPRODUCIBLE run to run, like the forward pass (and like the
 * reference's CPU path). */
size_t nemar_conv2d_bwd_weight_workspace(int N, int C, int H, int W, int K, int OH, int OW, int R, int S, int stride,
                                         int pad);
int nemar_conv2d_bwd_weight(const float* x0, int C0, const float* x1, int C1, const float* gy, float* gw, float* gb,
                            int N, int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad,
                            int pad_mode, void* workspace, size_t ws_bytes, void* stream);
/* The same three entry points with the SIDE INPUTS of the wide-layer fp16 x 3 route (below) passed with the call — the only way to
 * hand them over (rounds 2-3 also had process-wide registrations: nemar_set_scratch / nemar_absmax_hint / nemar_planes_hint; they are
 * gone).  Any member may be NULL / 0 = "not given"; extras == NULL is the plain entry point.  Same kernels, bit-identical results.
 *   scratch / scratch_bytes     transient arena of the wide-layer fp16 x 3 route for THIS call (nemar_conv2d_scratch bytes)
 *   src_max_words / _count      per-sample max |source| words (source = x0 for fwd / bwd_weight, gy for bwd_data); count = N or 1
 *   src2_max_words / _count     bwd_weight only: the same for gy
 *   src_planes                  the source's operand planes written by its producer, scaled by src_max_words (count = N).  fwd: the
 *                               channel-blocked planes of x0 (nemar_instnorm_fwd_planes `planes`); bwd_data: the data-gradient planes of
 *                               gy (nemar_instnorm_bwd_planes `dgrad_planes`, written for THIS layer's pad_mode); bwd_weight: the
 *                               pixel-major X planes of x0 (nemar_instnorm_fwd_planes `wgrad_planes`).  With planes the fp32 tensor of
 *                               that operand is not read: its pointer only has to be non-NULL and distinct
 *   gy_planes_out / _bytes      bwd_data only: a buffer of nemar_conv2d_gy_planes_bytes(...) bytes.  When given together with
 *                               src_max_words (count = N) on a layer of the wide route, the pass that splits gy for the data gradient
 *                               ALSO writes the operand planes the weight gradient of the same layer needs (gy is read once) ...
 *   src2_planes                 ... and bwd_weight takes them here (with the SAME words as src2_max_words) instead of splitting gy again.
 *                               The buffer must stay untouched between the two calls.  (nemar_instnorm_bwd_planes `wgrad_planes`
 *                               are the same planes from the producer of gy.)
 *   addend                      bwd_data only: a tensor of gx0's shape ADDED to the data gradient in the kernel's epilogue (the
 *                               ResnetBlock skip gradient, reference models/networks.py:443-446: out = x + conv_block(x))
 *   out_max_words               bwd_data only: a NEMAR_MAX_WORDS(N) buffer; the epilogue publishes the per-sample max |gx0| (words 0..N-1)
 *                               Both only where nemar_conv2d_bwd_data_fusable(...) says 1 (the addend alone: where
 *                               nemar_conv2d_bwd_data_addend_ok(...) says 1) — elsewhere the call fails with NEMAR_EINVAL.
 * A packed-weight workspace (prepacked = 1) must be reused under the same route conditions it was written under (arena present or
 * not, nemar_config_epoch unchanged). */
typedef struct nemar_conv_extras {
    void* scratch;
    size_t scratch_bytes;
    const void* src_max_words;
    int src_max_count;
    const void* src2_max_words;
    int src2_max_count;
    const void* src_planes;
    void* gy_planes_out;
    size_t gy_planes_bytes;
    const void* src2_planes;
    const float* addend;
    void* out_max_words;
    const float* bias_partials;   /* (ABI 602) bwd_weight only, with gb != NULL on a layer of the wide route: per-plane sums of gy [N, K]
                                   * (nemar_instnorm_bwd_planes `bias_partials`); gb += their sum over the batch, in batch order, inside the
                                   * launch that sums the weight gradient's slabs — instead of a nemar_bias_from_partials call of its own */
} nemar_conv_extras;
/* 1: the layer's bwd_data_ex honours addend / out_max_words and takes src_planes, and its bwd_weight_ex takes both operands as planes */
int nemar_conv2d_bwd_data_fusable(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int pad_mode);
/* (ABI 602) 1: the layer's bwd_data_ex (one destination, no bias, no activation) adds extras.addend to gx0 — the wide route's epilogue, or
 * the fold pass that ends the data gradient of a small stride-1 reflect layer (out_max_words is NOT honoured there: only where _fusable says 1) */
int nemar_conv2d_bwd_data_addend_ok(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int pad_mode);
/* bytes of the pixel-major X planes nemar_instnorm_fwd_planes can write for the weight gradient of a KS x KS / pad-1 reflect layer (0: none) */
size_t nemar_conv2d_x_planes_bytes(int N, int C, int H, int W, int KS);
/* bytes of the gy planes a bwd_data call can leave behind for the bwd_weight call of the same layer (0: this layer does not take them) */
size_t nemar_conv2d_gy_planes_bytes(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int pad_mode);
/* 1 when the last nemar_conv2d_bwd_data_ex call on this thread filled its gy_planes_out buffer (check before passing it on as src2_planes) */
int nemar_last_gy_planes(void);
int nemar_conv2d_fwd_ex(const float* x0, int C0, const float* x1, int C1, const float* w, const float* bias, float* y, int N,
                        int H, int W, int K, int R, int S, int stride, int pad, int pad_mode, int act, float slope,
                        void* workspace, size_t ws_bytes, int prepacked, void* stream, const nemar_conv_extras* extras);
int nemar_conv2d_bwd_data_ex(const float* gy, const float* w, const float* bias, int act, float slope, float* gx0, int C0,
                             float* gx1, int C1, int N, int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad,
                             int pad_mode, void* workspace, size_t ws_bytes, int prepacked, void* stream,
                             const nemar_conv_extras* extras);
int nemar_conv2d_bwd_weight_ex(const float* x0, int C0, const float* x1, int C1, const float* gy, float* gw, float* gb, int N,
                               int H, int W, int K, int OH, int OW, int R, int S, int stride, int pad, int pad_mode,
                               void* workspace, size_t ws_bytes, void* stream, const nemar_conv_extras* extras);
/* Weight-pack plans.  Every convolution family reads its weights from a packed operand image in the `workspace` of the call
 * (prepacked = 0: the call rebuilds it first — a max reduction and a pack launch per weight and direction, ~240 launches of 4-6 us per
 * training step).  A plan batches them: while nemar_pack_plan_record(plan) is in effect on the calling thread, the pack launches the
 * convolution entry points issue are ALSO recorded as jobs of `plan` (record(-1) stops); nemar_pack_plan_commit copies the recorded
 * job arguments into a caller-owned device buffer of nemar_pack_plan_bytes(plan) bytes (once per change: nemar_pack_plan_dirty);
 * nemar_pack_plan_run re-runs every job — the same images, bit for bit — in at most five launches (one multi-job kernel per family,
 * job arguments read from that buffer), after which the owners of those workspaces may call with prepacked = 1.  The caller keeps the
 * weights, the workspaces and the device buffer alive and unmoved while the plan is in use, and uses one plan per group of weights
 * that change together (an optimizer).  nemar_pack_plan_jobs -> pack jobs recorded; nemar_pack_plan_reset forgets the plan. */
int nemar_pack_plan_record(int plan);
int nemar_pack_plan_jobs(int plan);
size_t nemar_pack_plan_bytes(int plan);
int nemar_pack_plan_dirty(int plan);
int nemar_pack_plan_commit(int plan, void* device_buffer, size_t bytes, void* stream);
int nemar_pack_plan_run(int plan, void* stream);
int nemar_pack_plan_reset(int plan);
/* Which kernel family served the calling thread's last nemar_conv2d_* call: 0 exact-fp32 implicit GEMM, 1 narrow (<= 4 channel)
 * VALU kernels, 2 split-16 kernels of the wide residual-block layers, 3 general 16-bit-pipe kernels (tests / tools). */
int nemar_last_route(void);
/* The route a shape takes — and with it the FORMAT of the packed weight image a `prepacked` call finds in its workspace — is a
 * function of the shape and of the library's measurement switches.  This library has none: always 0.  In the measurement build
 * (nemar_hip_ab.h) it is a counter bumped by every nemar_tune call, and a caller that caches packed workspaces keys them with it. */
int nemar_config_epoch(void);
/* Per calling thread, 1 / 0, returns the previous setting (0.6.1).  While on, the two producers of per-sample maxima on the residual blocks'
 * path — nemar_instnorm_fwd_planes (max_words) and nemar_conv2d_bwd_data_ex (nemar_conv_extras.out_max_words) — do NOT launch the reduction
 * of their per-workgroup partial words; word n of the buffer holds a marker (0xFFFF0000 | partials) instead.  Such a buffer may ONLY be
 * passed on as nemar_instnorm_fwd_planes residual_max_words or nemar_instnorm_bwd_planes gy_max_words, which reduce a sample's partials
 * themselves; every other consumer of max words needs finalized words (the default).  One launch less per producer call: 4 us on one
 * stream, 12 us beside a second stream, 25 times per training step on the chain the rest of the step waits for. */
int nemar_set_max_words_lazy(int on);
/* (ABI 602) The reduction a lazy producer skipped, on demand: every sample whose result word still holds the marker gets the maximum of its partial
 * words; finalized words are left alone.  nemar_instnorm_fwd_max / _bwd_max honour the lazy setting too (ABI 602): a binding that publishes maxima
 * "in case the consumer is a wide-route convolution" pays the reduction only for the tensors such a consumer actually reads. */
int nemar_max_words_finalize(void* max_words, int samples, void* stream);
/* Transient scratch arena for nemar_conv2d_fwd / nemar_conv2d_bwd_data / nemar_conv2d_bwd_weight.  The wide stride-1 / pad-1
 * layers (the 3x3 ResnetBlock convolutions, reference models/networks.py:418-439, and the discriminator's 256->512 4x4 layer,
 * :576-597; >= 128 channels, >= 2 G multiply-adds) run on the 16-bit matrix pipe at fp32 accuracy: each fp32 operand is split into
 * two power-of-two-scaled fp16 terms and three partial products are accumulated in fp32 (csrc/conv_split16*.hip; measured error
 * against float64 below that of the exact-fp32 MFMA kernels).  The split copies of the source tensors live in this
 * arena for the duration of the call.  nemar_conv2d_scratch -> bytes the layer wants (0: never uses it); the caller passes a device
 * buffer of at least that size as nemar_conv_extras.scratch (one stream at a time per buffer).  A layer called without an arena, or
 * with one that is too small, runs on the exact-fp32 MFMA kernels instead — same results within fp32 rounding. */
size_t nemar_conv2d_scratch(int N, int H, int W, int K, int C, int R, int S, int stride, int pad);
/* The fp16 form of those kernels scales every SAMPLE of a source tensor by its own power of two, derived from the largest finite
 * magnitude of the sample (so samples of very different magnitude in one batch — the batched real / fake passes — do not share a
 * scale).  Error bound of the fp16 x 3 form, per product v w: |error| <= 3 * 2^-22 |v w| as long as |v| >= 2^-14 max_sample|v|
 * (both fp16 terms normal); below that the absolute error of v is 2^-37 max_sample|v|.  Non-finite elements do not take part in the
 * maximum and propagate as Inf / NaN; an all-zero sample has scale 1.  (The general kernels of tune key 24 go further: their scale
 * is per workgroup tile and 16-channel chunk, or per channel row in the weight gradient — csrc/conv_s16g*.hip.)
 * A caller that feeds one tensor to several calls (x: forward and weight gradient; gy: data and weight gradient) computes the words
 * once with nemar_absmax_samples (out_words: `samples` 4-byte words that are ZERO on entry — the kernel takes an atomic max of the bit
 * patterns into them; nemar_absmax = one sample) and passes them as nemar_conv_extras.src_max_words / src2_max_words (count = the
 * tensor's batch size, or 1 = one word for the whole tensor).  Without them every call runs its own max pass. */
int nemar_absmax(const float* t, long long n, void* out_word, void* stream);
int nemar_absmax_samples(const float* t, int samples, long long per_sample, void* out_words, void* stream);
/* Measurement hook for bench.py's roofline entry: while enabled, HIP events are recorded on the launch stream around the main
 * kernel (igemm_split16_kernel) of every forward / data-gradient call of those layers; read -> summed duration, summed algorithmic
 * (fp32-equivalent) flop 2 N OH OW K C R S of the timed launches, and their count (synchronises on the recorded events, resets). */
int nemar_kernel_timer(int enable);
int nemar_kernel_timer_read(double* total_ms, double* total_flop, int* launches);
/* gb[C] += sum over batch and plane of g [N,C,HW] (bias gradient; two fixed-order stages through `workspace`). */
size_t nemar_bias_grad_workspace(int N, int C, int HW);
int nemar_bias_grad(const float* g, float* gb, int N, int C, int HW, void* workspace, size_t ws_bytes, void* stream);

/* ---- K3 (+K5): InstanceNorm2d(affine=False, track_running_stats=False) with fused activation / residual -------
 * nn.InstanceNorm2d — reference models/networks.py:24 (used :351,358,373,426,439,584,592), models/stn/layers.py:16;
 * the ReLU/LeakyReLU(0.2) after it, and the ResnetBlock skip `x + conv_block(x)` (models/networks.py:445).
 * x,y,residual: [planes = N*C, HW]; stats [planes,2] = (mean, rstd) saved for the backward.
 * fwd: y = (residual ? residual : 0) + act((x - mean) * rstd), biased variance, eps inside the sqrt.
 * bwd: gx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = gy * act'(xhat).  act: NONE | RELU | LRELU. */
int nemar_instnorm_fwd(const float* x, const float* residual, float* y, float* stats, int planes, int HW,
                       float eps, int act, float slope, void* stream);
int nemar_instnorm_bwd(const float* x, const float* stats, const float* gy, float* gx, int planes, int HW,
                       int act, float slope, void* stream);
/* The same, and max |output| (finite elements) of every SAMPLE into max_words[sample]: what nemar_absmax_samples would compute in a
 * pass of its own — the producer has the values in registers.  The words are what nemar_conv_extras.src_max_words takes (the
 * fp16 x 3 convolutions consume them).  max_words is a buffer of NEMAR_MAX_WORDS(samples) 4-byte words, no initialisation needed: the first
 * `samples` words are the result, the rest is scratch for the per-workgroup partials a second tiny launch reduces (no atomics —
 * thousands of workgroups' device-scope atomicMax on a handful of words cost more than the pass they replaced).
 * planes_per_sample <= NEMAR_MAX_PARTIALS. */
#define NEMAR_MAX_PARTIALS 2048
#define NEMAR_MAX_WORDS(samples) ((size_t)(samples) * (1 + NEMAR_MAX_PARTIALS))
int nemar_instnorm_fwd_max(const float* x, const float* residual, float* y, float* stats, int planes, int HW,
                           float eps, int act, float slope, void* max_words, int planes_per_sample, void* stream);
int nemar_instnorm_bwd_max(const float* x, const float* stats, const float* gy, float* gx, int planes, int HW,
                           int act, float slope, void* max_words, int planes_per_sample, void* stream);

/* InstanceNorm (+ activation, + Dropout(p) drawn exactly as nemar_dropout draws it over the [N,C,H,W] tensor, + residual) that ALSO
 * writes its output as the fp16 x 3 planes a following 3x3 / pad-1 REFLECT convolution of the wide-layer route consumes
 * (reference: the conv -> InstanceNorm -> ReLU -> [Dropout] -> ReflectionPad2d(1) -> conv chain of ResnetBlock,
 * models/networks.py:418-446) — the consumer then skips its max and split passes (nemar_conv_extras.src_planes + .src_max_words).
 *   y          fp32 output or NULL (planes only: nothing else reads it)
 *   planes     2 * N * (C/8) * (H+4) * (W+4) 16-byte words (hi plane, lo plane), layout of conv_split16.hip
 *   scale_words[N]  out: the a-priori BOUND each sample was scaled by, sqrt(HW) [/ (1-p)] [+ max |residual| of the sample] —
 *              InstanceNorm's output cannot exceed it, so the scale is known before the first element exists; pass these words as
 *              src_max_words of the consuming convolution (its epilogue unscales with them).  A bound 2^k above the actual
 *              maximum costs nothing for k <= 8: elements above 2^(k-14) x max keep the split's 22 bits, smaller ones are off by
 *              <= 2^(k-36) x max (norm_planes.hip).
 *   residual_max_words[N]  required with a residual: its per-sample maxima (a producer's max words)
 *   max_words  NEMAR_MAX_WORDS(N) buffer or NULL: the ACTUAL per-sample maxima of the output (the next layer's residual bound)
 *   wgrad_planes  NULL, or nemar_conv2d_x_planes_bytes(N, C, H, W, 3) bytes: ALSO the pixel-major X planes the WEIGHT gradient of that
 *              convolution reads (same scale; pass as src_planes + src_max_words of its nemar_conv2d_bwd_weight_ex): the layer then
 *              runs neither of its split passes over this tensor
 * Limits: C % 8 == 0, W % 4 == 0, H * W <= 4096, N <= 256. */
int nemar_instnorm_fwd_planes(const float* x, const float* residual, const void* residual_max_words, float* y, float* stats,
                              int N, int C, int H, int W, float eps, int act, float slope, float dropout_p,
                              unsigned long long seed, unsigned offset, void* planes, void* scale_words, void* max_words,
                              void* wgrad_planes, void* stream);
/* The backward of that producer for a 3x3 / pad-1 convolution of the wide route IN FRONT of it (reference: autograd through
 * conv -> InstanceNorm -> ReLU -> [Dropout] of ResnetBlock, models/networks.py:418-446): gx = InstanceNorm backward of gy through
 * [dropout ->] act -> InstanceNorm (nemar_instnorm_bwd's formula; the dropout mask of (p, seed, offset) regenerated), written as the
 * OPERAND PLANES of the convolution's two gradient calls instead of (gx != NULL: besides) the fp32 tensor:
 *   dgrad_planes   2 * N * (C/8) * (H+4) * (W+4) 16-byte words: gy planes of nemar_conv2d_bwd_data_ex (extras.src_planes) for a layer
 *                  with this pad_mode (0 zero, 1 reflect: the folded border rows of the reflect data gradient)
 *   wgrad_planes   nemar_conv2d_gy_planes_bytes(...) bytes: gy planes of nemar_conv2d_bwd_weight_ex (extras.src2_planes)
 *   scale_words[N] out: the a-priori bound the planes are scaled by, rstd_max(sample) * (2 + sqrt(HW)) [/ (1-p)] * max|gy|(sample);
 *                  pass as src_max_words / src2_max_words of the two calls
 *   gy_max_words[N]  per-sample max |gy| (a producer's words, nemar_conv_extras.out_max_words, or nemar_absmax_samples)
 *   bias_partials  NULL or [N, C]: the sum of gx over each plane (the convolution's bias gradient = their sum over the batch:
 *                  nemar_bias_from_partials)
 * Either planes pointer may be NULL.  Limits as nemar_instnorm_fwd_planes; wgrad_planes: C % 64 == 0, W % 8 == 0. */
int nemar_instnorm_bwd_planes(const float* x, const float* stats, const float* gy, const void* gy_max_words, int N, int C,
                              int H, int W, int act, float slope, float dropout_p, unsigned long long seed, unsigned offset,
                              int pad_mode, float* gx, void* dgrad_planes, void* wgrad_planes, void* scale_words,
                              float* bias_partials, void* stream);
/* gb[C] += sum over the batch of bias_partials [N, C], in batch order (bitwise reproducible) */
int nemar_bias_from_partials(const float* bias_partials, int N, int C, float* gb, void* stream);
/* ---- K5/K6/K7: pointwise, pooling, resize, dropout -----------------------------------------------------------------
 * act_bwd: gx = gy * f'(.) expressed with the activation OUTPUT y (f fused into a conv epilogue):
 *     nn.LeakyReLU / nn.ReLU / nn.Tanh — reference models/networks.py:377,576 ; models/stn/layers.py:61-64. */
int nemar_act_bwd(const float* gy, const float* y, float* gx, long long n, int act, float slope, void* stream);
/* y = act(x) as a stand-alone pass (nn.ReLU / nn.LeakyReLU / nn.Tanh where no producer epilogue can carry it: the U-Net
 * generator's skip path, reference models/networks.py:516-553).  Backward = nemar_act_bwd on the output. */
int nemar_act_fwd(const float* x, float* y, long long n, int act, float slope, void* stream);
/* (ABI 602) dst = [piece 0 | piece 1 | ... | piece k-1] (k <= 8 contiguous pieces of counts[i] floats; `pieces` / `counts` are HOST arrays,
 * read during the call; a NULL piece contributes zeros).  The concatenation along the batch of the model's batched passes
 * (torch.cat of reference models/nemar_model.py:181,220 evaluated once for several images) and, with the incoming gradients as pieces, the
 * backward of slicing such a batch — instead of autograd's cat / zero-fill / copy / add kernels. */
int nemar_concat_pieces(const float* const* pieces, const long long* counts, int k, float* dst, void* stream);
/* (ABI 602) out = a + b (out may alias a or b): the sum of the gradients of a tensor with two consumers (a ResnetBlock's input, a U-Net skip:
 * reference models/networks.py:443-446, models/stn/unet_stn.py:80-97), which autograd would add with an ATen kernel. */
int nemar_add2(const float* a, const float* b, float* out, long long n, void* stream);
/* nn.MaxPool2d(2) — reference models/stn/layers.py:174.  x [planes,H,W] -> y [planes,H/2,W/2].
 * bwd: gx = (addend ? addend : 0) + unpool(gy); the argmax (first maximum, row-major) is recomputed from x. */
int nemar_maxpool2_fwd(const float* x, float* y, int planes, int H, int W, void* stream);
int nemar_maxpool2_bwd(const float* x, const float* gy, const float* addend, float* gx, int planes, int H, int W,
                       void* stream);
/* F.interpolate(x, (Ho,Wo), mode='bilinear', align_corners=False) — reference models/stn/unet_stn.py:96,188-195,
 * models/nemar_model.py:187-188,204-205,226-227,240-241,254-255.  bwd WRITES gx [planes,H,W]. */
int nemar_bilinear_fwd(const float* x, float* y, int planes, int H, int W, int Ho, int Wo, void* stream);
int nemar_bilinear_bwd(const float* gy, float* gx, int planes, int H, int W, int Ho, int Wo, void* stream);
/* nn.Dropout(p) in training mode — reference models/networks.py:427-428.  y = x * mask / (1-p), mask drawn from
 * Philox4x32-10(seed, offset); the backward is the same call on the upstream gradient (mask regenerated). */
int nemar_dropout(const float* x, float* y, long long n, float p, unsigned long long seed, unsigned offset,
                  void* stream);
/* ... with the per-sample maxima of the output (max_words[i]; a NEMAR_MAX_WORDS(samples) buffer as above) for `samples` samples of
 * `per_sample` elements (a multiple of 4); the masks are those of nemar_dropout over all samples * per_sample elements. */
int nemar_dropout_max(const float* x, float* y, int samples, long long per_sample, float p, unsigned long long seed,
                      unsigned offset, void* max_words, void* stream);

/* Replaying a step as a captured hipGraph: launch arguments are frozen at capture time, so what changes from step to step must live in
 * device memory.  nemar_set_dropout_base registers a device word that every dropout-type launch (nemar_dropout, nemar_dropout_max,
 * nemar_instnorm_fwd_planes) adds to its `offset` at run time (NULL = none, the default); the caller rewrites the word before each replay.
 * nemar_adam_step_dev is nemar_adam_step with hyper[0] = lr / (1 - beta1^step) and hyper[1] = sqrt(1 - beta2^step) read from device memory
 * (computed by the caller in double, rounded to float — exactly what nemar_adam_step passes to its kernel). */
int nemar_set_dropout_base(const void* device_word);
/* dst[0..n) = the n <= 8 four-byte words at host_words, stream-ordered: the values travel as KERNEL ARGUMENTS (read from host memory
 * during the call), so the host buffer may be reused at once — unlike an asynchronous copy from pageable memory, whose source may be
 * read later.  How the step's device-resident parameters (dropout base word, Adam's two scalars) are rewritten before a graph replay. */
int nemar_store_words(void* dst, const void* host_words, int n, void* stream);
int nemar_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper, double beta1,
                        double beta2, double eps, void* stream);

/* Input pipeline, on the GPU: random crop + horizontal flip + ToTensor/Normalize(0.5, 0.5) of a pool of images resident in
 * HBM — reference data/base_dataset.py:63-78 (get_params: ONE crop position / flip per A-B pair) and :81-112
 * (get_transform; Normalize :111).  pool [M,C,H,W] with values in [0, 1/scale]; params [B,4] int32 device array
 * (pool index, y0, x0, flip), 16-byte aligned; y [B,C,Hc,Wc] = (pool[...] * scale - 0.5) / 0.5. */
int nemar_crop_flip_normalize(const float* pool, const int* params, float* y, int M, int B, int C, int H, int W,
                              int Hc, int Wc, float scale, void* stream);

/* ---- K13: losses (already multiplied by their lambda `weight`; optionally accumulated into a device scalar) ------
 * l1:  torch.nn.L1Loss — reference models/nemar_model.py:68,179,195; b == NULL gives mean|a|
 *      (affine STN regulariser, reference models/stn/affine_stn.py:136-138).
 * gan: GANLoss.__call__ against a constant target — reference models/networks.py:263-281.
 *      mode NEMAR_GAN_VANILLA (BCEWithLogits) | NEMAR_GAN_LSGAN (MSE) | NEMAR_GAN_WGANGP (+-mean).
 * fwd: loss[0] = (accumulate ? loss[0] : 0) + weight * L ;  bwd: grad = gscale[0] * weight * dL/dx. */
#define NEMAR_GAN_VANILLA 0
#define NEMAR_GAN_LSGAN 1
#define NEMAR_GAN_WGANGP 2
size_t nemar_loss_workspace(void);
int nemar_l1_loss_fwd(const float* a, const float* b, long long n, float weight, float* loss, int accumulate,
                      void* workspace, size_t ws_bytes, void* stream);
int nemar_l1_loss_bwd(const float* a, const float* b, long long n, const float* gscale, float weight, float* ga,
                      int accumulate, void* stream);
int nemar_gan_loss_fwd(const float* x, long long n, int mode, int target_is_real, float weight, float* loss,
                       int accumulate, void* workspace, size_t ws_bytes, void* stream);
int nemar_gan_loss_bwd(const float* x, long long n, int mode, int target_is_real, const float* gscale,
                       float weight, float* gx, void* stream);

/* ---- K15: fused Adam over a flat parameter buffer --------------------------------------------------------------------
 * torch.optim.Adam(lr, betas=(beta1, beta2), eps).step() — reference models/nemar_model.py:128-137 (construction),
 * :274,282-283 (step).  p,g,m,v: length n; step is 1-based.  Hyper-parameters are doubles, as in torch. */
int nemar_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1,
                    double beta2, double eps, int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEMAR_HIP_H */
