"""`-m gpu` tier: the real gfx950 library (nemar_amd/lib/libnemar_hip.so) through its C-ABI, against the
oracle, with the same bodies as the CPU/emulator tier plus hot-path sizes."""
import numpy as np
import pytest

import kernel_cases as K
from backends import HipBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be(hip_lib):
    return HipBackend(hip_lib)


@pytest.mark.parametrize("mode", [K.GRID_UNET, K.GRID_AFFINE, K.GRID_EXPLICIT])
@pytest.mark.parametrize("scale", [0.01, 0.05, 1.5])
def test_grid_sample(be, mode, scale):
    K.case_grid_sample(be, mode, N=2, C=3, H=12, W=16, Ho=12, Wo=16, scale=scale)


def test_grid_sample_ragged_and_resampled(be):
    K.case_grid_sample(be, K.GRID_UNET, N=1, C=1, H=7, W=9, Ho=7, Wo=9, scale=0.1)
    K.case_grid_sample(be, K.GRID_AFFINE, N=3, C=2, H=10, W=6, Ho=5, Wo=8, scale=0.2)
    K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=8, W=8, Ho=8, Wo=8, scale=0.1, need_gin=False)
    K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=8, W=8, Ho=8, Wo=8, scale=0.1, accumulate=True)
    K.case_grid_sample(be, K.GRID_AFFINE, N=2, C=3, H=8, W=8, Ho=8, Wo=8, scale=0.1, accumulate=True)
    # the LDS-tile grad_input variant (nemar_grid_sample_tune(1)) on several 16x64 tiles: halo overlap between neighbours
    # (small offsets), the global-atomic fallback for corners outside a tile's region (large offsets), > 4 channels
    for variant in (1, 2):          # 2 = global fp32 atomics (the round-1 default)
        be.lib.grid_sample_tune(variant)
        try:
            K.case_grid_sample(be, K.GRID_UNET, N=1, C=2, H=40, W=150, Ho=40, Wo=150, scale=0.02, atomic=True)
            K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=40, W=150, Ho=40, Wo=150, scale=0.5, atomic=True)
            K.case_grid_sample(be, K.GRID_AFFINE, N=2, C=1, H=36, W=70, Ho=36, Wo=70, scale=0.3, accumulate=True, atomic=True)
            K.case_grid_sample(be, K.GRID_UNET, N=1, C=6, H=12, W=16, Ho=12, Wo=16, scale=0.1, atomic=True)
        finally:
            be.lib.grid_sample_tune(0)
    K.case_grid_sample(be, K.GRID_UNET, N=1, C=2, H=40, W=150, Ho=40, Wo=150, scale=0.5, workspace=False)   # no workspace: atomics


def test_grid_sample_bwd_gather_and_fixed_point_paths(be):
    """Default grad_input path on several 64x16 destination tiles: near pixels (gather in LDS, halo across tile borders),
    far pixels (64-bit fixed-point atomics + fold), a mix of both, every grid mode, accumulate, tile tails; each case runs
    twice on one workspace and must be bitwise identical with the accumulator returned all-zero (kernel_cases)."""
    be.lib.grid_sample_tune(16)                 # A/B variant: 256-thread workgroups
    try:
        K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=40, W=150, Ho=40, Wo=150, scale=0.02)
    finally:
        be.lib.grid_sample_tune(0)
    be.lib.grid_sample_tune(8)                  # A/B variant: grid gradient in its own pass
    try:
        K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=40, W=150, Ho=40, Wo=150, scale=0.02)
        K.case_grid_sample(be, K.GRID_AFFINE, N=2, C=3, H=36, W=70, Ho=36, Wo=70, scale=0.02, accumulate=True)
    finally:
        be.lib.grid_sample_tune(0)
    K.case_grid_sample(be, K.GRID_UNET, N=1, C=2, H=40, W=150, Ho=40, Wo=150, scale=0.0)       # the linspace zoom only: all near
    K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=40, W=150, Ho=40, Wo=150, scale=0.02)      # ~1.5 px noise: near + a few far
    K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=40, W=150, Ho=40, Wo=150, scale=0.5)       # every pixel far (and many OOB)
    K.case_grid_sample(be, K.GRID_UNET, N=1, C=4, H=33, W=70, Ho=33, Wo=70, scale=0.05, accumulate=True)
    K.case_grid_sample(be, K.GRID_AFFINE, N=2, C=3, H=36, W=70, Ho=36, Wo=70, scale=0.02)
    K.case_grid_sample(be, K.GRID_AFFINE, N=2, C=1, H=36, W=70, Ho=36, Wo=70, scale=0.3, accumulate=True)
    K.case_grid_sample(be, K.GRID_EXPLICIT, N=1, C=3, H=20, W=66, Ho=20, Wo=66, scale=0.03)
    K.case_grid_sample(be, K.GRID_UNET, N=1, C=6, H=12, W=16, Ho=12, Wo=16, scale=0.1, atomic=True)   # C > 4: legacy kernels
    # smooth LARGE deformations: the gather windows follow the field (tile offsets != 0), straddling tiles with different offsets
    K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=40, W=150, Ho=40, Wo=150, scale=0.001, smooth_px=(9.3, -4.6, 2.0))
    K.case_grid_sample(be, K.GRID_UNET, N=1, C=3, H=48, W=200, Ho=48, Wo=200, scale=0.004, smooth_px=(-17.0, 6.2, 6.0), accumulate=True)
    K.case_grid_sample(be, K.GRID_EXPLICIT, N=1, C=2, H=33, W=130, Ho=33, Wo=130, scale=0.0, smooth_px=(30.5, 12.0, 1.0))   # partly out of bounds
    be.lib.grid_sample_tune(32)                 # A/B variant: windows centred on the tiles (round 2): same results
    try:
        K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=40, W=150, Ho=40, Wo=150, scale=0.001, smooth_px=(9.3, -4.6, 2.0))
    finally:
        be.lib.grid_sample_tune(0)


@pytest.mark.parametrize("mode,scale", [(K.GRID_UNET, 0.0), (K.GRID_UNET, 2.0 / 256), (K.GRID_UNET, 0.1),
                                        (K.GRID_AFFINE, 0.02)])
def test_grid_sample_hot_path_size(be, mode, scale):
    # BASELINE config 2 warp: 8 x 3 x 256 x 256 (oracle finishes in seconds)
    K.case_grid_sample(be, mode, N=8, C=3, H=256, W=256, Ho=256, Wo=256, scale=scale)


@pytest.mark.parametrize("Ci,alpha", [(0, 0.0), (3, 0.0), (3, 1.7), (1, 0.5)])
def test_smoothness(be, Ci, alpha):
    K.case_smoothness(be, N=2, H=9, W=13, Ci=Ci, alpha=alpha)


def test_smoothness_accumulate_factor(be):
    K.case_smoothness(be, N=1, H=2, W=2, Ci=3, alpha=0.9, factor=0.5, accumulate=True)
    K.case_smoothness(be, N=3, H=17, W=5, Ci=0, alpha=0.0, factor=0.25)


@pytest.mark.parametrize("alpha", [0.0, 2.0])
def test_smoothness_hot_path_size(be, alpha):
    K.case_smoothness(be, N=8, H=256, W=256, Ci=3, alpha=alpha)


from test_kernels_emu import CONV_CASES  # noqa: E402  (same shapes as the emulator tier)


@pytest.mark.parametrize("C0,C1,Kc,R,stride,pad,pm", CONV_CASES)
def test_conv_fwd(be, C0, C1, Kc, R, stride, pad, pm):
    K.case_conv_fwd(be, 2, C0, C1, 9, 10, Kc, R, stride, pad, pm, act=K.O.ACT_LRELU)


@pytest.mark.parametrize("C0,C1,Kc,R,stride,pad,pm", CONV_CASES)
def test_conv_bwd_data(be, C0, C1, Kc, R, stride, pad, pm):
    if pm == K.PAD_REFLECT and C1:
        pytest.skip("reflect dgrad is single-destination")
    K.case_conv_bwd_data(be, 2, C0, C1, 9, 10, Kc, R, stride, pad, pm)


def test_conv_bwd_data_skip_first_source(be):
    K.case_conv_bwd_data(be, 2, 3, 3, 10, 8, 12, 4, 2, 1, K.PAD_ZERO, skip0=True)


def test_conv_bwd_data_reflect_variants(be):
    K.case_conv_bwd_data(be, 1, 8, 0, 2, 2, 8, 3, 1, 1, K.PAD_REFLECT)
    K.case_conv_bwd_data(be, 1, 4, 0, 4, 5, 6, 7, 1, 3, K.PAD_REFLECT)
    K.case_conv_bwd_data(be, 1, 16, 0, 9, 8, 5, 3, 2, 1, K.PAD_REFLECT)


@pytest.mark.parametrize("C0,C1,Kc,R,stride,pad,pm", CONV_CASES)
def test_conv_bwd_weight(be, C0, C1, Kc, R, stride, pad, pm):
    K.case_conv_bwd_weight(be, 2, C0, C1, 9, 10, Kc, R, stride, pad, pm)


@pytest.mark.parametrize("N,C0,C1,H,W,Kc,R,stride,pad,pm", K.WGRAD_WIDE_CASES)
def test_conv_bwd_weight_wide(be, N, C0, C1, H, W, Kc, R, stride, pad, pm):
    K.case_conv_bwd_weight(be, N, C0, C1, H, W, Kc, R, stride, pad, pm)


@pytest.mark.parametrize("cfg", [0, 7, 6, 5, 4, 1, 2])
def test_conv_forced_128_tiles(be, cfg):
    """The 128x128 workgroup shapes (wave-specialised gen 2 / gen 1, 4-wave, 8-wave) on small problems: nemar_tune key 6
    lowers the grid-size threshold that normally reserves them for large layers."""
    be.lib.tune(6, 1)
    be.lib.tune(0, cfg)
    try:
        K.case_conv_fwd(be, 2, 16, 0, 9, 10, 70, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_LRELU)   # ragged M and P, scalar stores
        K.case_conv_fwd(be, 2, 16, 16, 8, 12, 130, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_RELU)     # concat, 2 M tiles, 16 B stores
        K.case_conv_fwd(be, 3, 32, 0, 8, 8, 128, 4, 2, 1, K.PAD_ZERO)                          # k4 s2
        K.case_conv_fwd(be, 2, 16, 0, 6, 8, 40, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_RELU)      # 9 stages: odd tail of the 2-stage barrier variant
        K.case_conv_bwd_data(be, 2, 70, 0, 8, 8, 32, 3, 1, 1, K.PAD_REFLECT)                   # dgrad: M = 70 source chans
        K.case_conv_bwd_data(be, 2, 130, 0, 8, 8, 16, 4, 2, 1, K.PAD_ZERO)                     # parity classes (strided out)
        K.case_conv_transpose_fwd(be, 2, 16, 72, 5, 6, 3, 1)
    finally:
        be.lib.tune(0, 0)
        be.lib.tune(6, 384)


@pytest.mark.parametrize("mt", [4, 2])
def test_conv_ws2_direct_weight_fragments_variant(be, mt):
    """nemar_tune(16, 1): every MFMA wave fetches its own A fragments (packed weights) straight from global memory instead of
    through LDS — an A/B variant (measured slower on MI355X), kept correct."""
    be.lib.tune(7, mt)
    be.lib.tune(16, 1)
    try:
        K.case_conv_fwd(be, 2, 16, 0, 6, 8, 40, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_RELU)
        K.case_conv_fwd(be, 1, 16, 16, 9, 12, 70, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_LRELU)
        K.case_conv_fwd(be, 3, 32, 0, 8, 8, 128, 4, 2, 1, K.PAD_ZERO)
        K.case_conv_bwd_data(be, 2, 24, 0, 7, 8, 32, 3, 1, 1, K.PAD_REFLECT)
    finally:
        be.lib.tune(7, 0)
        be.lib.tune(16, 0)


@pytest.mark.parametrize("ring", [4, 5])
def test_conv_ws2_deeper_lds_ring(be, ring):
    """nemar_tune(18, .): loaders 3 / 4 stages ahead of the MFMA waves (counted waits over 2 / 3 stages in flight), incl.
    reductions shorter than the ring."""
    be.lib.tune(7, 4)
    be.lib.tune(18, ring)
    try:
        K.case_conv_fwd(be, 2, 16, 0, 6, 8, 40, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_RELU)       # 9 stages
        K.case_conv_fwd(be, 2, 16, 0, 3, 16, 20, 1, 1, 0, K.PAD_ZERO)                           # 1 stage
        K.case_conv_fwd(be, 1, 32, 0, 4, 8, 130, 1, 1, 0, K.PAD_ZERO)                           # 2 stages
        K.case_conv_fwd(be, 1, 48, 0, 4, 8, 130, 1, 1, 0, K.PAD_ZERO)                           # 3 stages
        K.case_conv_fwd(be, 1, 16, 0, 4, 8, 70, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_LRELU)         # 9 stages, M tail
        K.case_conv_bwd_data(be, 2, 24, 0, 7, 8, 32, 3, 1, 1, K.PAD_REFLECT)
    finally:
        be.lib.tune(7, 0)
        be.lib.tune(18, 3)


@pytest.mark.parametrize("nl", [1, 2])
def test_conv_ws2_256_channel_tiles(be, nl):
    """nemar_tune(17, .): 256 channels x 128 pixels per workgroup (8 MFMA waves + 2 / 4 loader waves)."""
    be.lib.tune(7, 4)
    be.lib.tune(17, nl)
    try:
        K.case_conv_fwd(be, 2, 16, 0, 8, 16, 200, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_RELU)      # ragged M = 200, 2 pixel tiles
        K.case_conv_fwd(be, 1, 16, 16, 9, 12, 260, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_LRELU)       # 2 channel tiles (260), concat
        K.case_conv_bwd_data(be, 2, 136, 0, 7, 8, 32, 3, 1, 1, K.PAD_REFLECT)                    # folded reflect border
    finally:
        be.lib.tune(7, 0)
        be.lib.tune(17, 0)


@pytest.mark.parametrize("mt", [1, 2, 4])
def test_conv_ws2_vector_loads(be, mt):
    """Wave-specialised igemm with 16-byte B loads (stride 1, OW % 4 == 0, |dx| <= 1): clamped border groups are patched
    by the MFMA waves — reflect and zero borders, 4-wide rows (one group is both first and last), tile tails."""
    be.lib.tune(7, mt)
    try:
        K.case_conv_fwd(be, 2, 16, 0, 6, 8, 40, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_RELU)
        K.case_conv_fwd(be, 3, 32, 0, 5, 4, 33, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_NONE)      # OW = 4
        K.case_conv_fwd(be, 1, 16, 16, 9, 12, 70, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_LRELU)   # concat, P = 108 (tail)
        K.case_conv_fwd(be, 2, 16, 0, 3, 16, 20, 1, 1, 0, K.PAD_ZERO)                       # 1x1
        K.case_conv_fwd(be, 2, 16, 0, 16, 32, 130, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_RELU) # 8 pixel tiles x 2 channel tiles: XCD-aware mapping
        K.case_conv_bwd_data(be, 2, 24, 0, 7, 8, 32, 3, 1, 1, K.PAD_REFLECT)                # ring + vector main pass
        K.case_conv_bwd_data(be, 2, 24, 0, 7, 8, 16, 3, 1, 1, K.PAD_ZERO)
        K.case_conv_transpose_fwd(be, 2, 16, 40, 5, 8, 3, 1)                                # parity classes, dx in {0, 1}
    finally:
        be.lib.tune(7, 0)


def test_conv_fwd_split_reduction(be):
    """Tiny, deep forward layers (the registration net's 2x2 .. 8x8 maps) split their reduction over grid.z: per-split slabs
    behind the packed weights, summed in split order (bias in slab 0); a fused ReLU / LeakyReLU is applied by the sum pass (nemar_tune(36, 0):
    such layers are not split)."""
    K.case_conv_fwd(be, 1, 128, 0, 2, 2, 128, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_NONE)    # STN bottleneck layer, 72 stages
    K.case_conv_fwd(be, 2, 64, 0, 4, 4, 64, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_NONE, bias=False)
    K.case_conv_fwd(be, 2, 64, 64, 4, 4, 64, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_LRELU)        # decoder conv: split + activation in the sum pass
    K.case_conv_fwd(be, 1, 64, 0, 8, 8, 64, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_RELU)
    be.lib.tune(36, 0)                                                                       # ... the unsplit form of the same layers
    try:
        K.case_conv_fwd(be, 2, 64, 64, 4, 4, 64, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_LRELU)
        K.case_conv_fwd(be, 1, 64, 0, 8, 8, 64, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_RELU)
    finally:
        be.lib.tune(36, 1)
    # tiny stride-1 reflect data gradients: ONE split launch over the padded domain, the sum pass folds the mirrored border (fold_small) ...
    K.case_conv_bwd_data(be, 2, 128, 0, 2, 2, 128, 3, 1, 1, K.PAD_REFLECT)                  # 2x2: every texel is a border texel
    K.case_conv_bwd_data(be, 2, 64, 0, 4, 4, 64, 3, 1, 1, K.PAD_REFLECT, seed=1)
    K.case_conv_bwd_data(be, 1, 64, 0, 6, 10, 64, 3, 1, 1, K.PAD_REFLECT, seed=2)           # non-square
    K.case_conv_bwd_data(be, 1, 40, 0, 16, 16, 72, 3, 1, 1, K.PAD_REFLECT, seed=3)          # ragged channel tiles
    K.case_conv_bwd_data(be, 2, 64, 0, 4, 4, 64, 3, 1, 1, K.PAD_REFLECT, seed=4, addend=True)   # + the skip gradient in the sum-and-fold pass
    K.case_conv_bwd_data(be, 1, 8, 0, 6, 10, 8, 3, 1, 1, K.PAD_REFLECT, seed=5, addend=True)    # unsplit: the addend rides in the plain fold pass
    be.lib.tune(43, 0)                                                                       # ... and the interior + ring form of the same layers
    try:
        K.case_conv_bwd_data(be, 2, 128, 0, 2, 2, 128, 3, 1, 1, K.PAD_REFLECT)              # main pass split + ring split
        K.case_conv_bwd_data(be, 2, 64, 0, 4, 4, 64, 3, 1, 1, K.PAD_REFLECT, seed=1)
    finally:
        be.lib.tune(43, 1)
    K.case_conv_bwd_data(be, 2, 64, 64, 4, 4, 64, 3, 1, 1, K.PAD_ZERO)                      # two destinations: the sum pass parts the rows


def test_conv_bwd_data_split_reduction(be):
    """Few, deep tiles (256 stages): the wave-specialised data gradient splits the reduction over grid.z; every split stores
    its partial gradient to its own slab and the slabs are summed in split order."""
    K.case_conv_bwd_data(be, 1, 70, 0, 6, 6, 256, 4, 1, 1, K.PAD_ZERO)      # D's 256->512 k4 layer in small
    K.case_conv_bwd_data(be, 2, 128, 0, 4, 8, 512, 3, 1, 1, K.PAD_REFLECT)  # + border ring on top of the split main pass


def test_conv_bwd_data_narrow_inputs(be):
    """Data gradient of layers with <= 4 input channels: correlation of gy with flipped/transposed weights on the
    narrow kernel (+ the border-ring launch for reflect padding)."""
    K.case_conv_bwd_data(be, 2, 2, 0, 9, 10, 20, 3, 1, 1, K.PAD_ZERO)
    K.case_conv_bwd_data(be, 2, 4, 0, 9, 10, 40, 4, 1, 1, K.PAD_ZERO)       # k4 p1: output one smaller, padding 2 in the gradient
    K.case_conv_bwd_data(be, 1, 3, 0, 12, 11, 16, 7, 1, 3, K.PAD_REFLECT)   # T stem
    K.case_conv_bwd_data(be, 1, 1, 0, 8, 8, 33, 3, 1, 0, K.PAD_ZERO)        # no padding: gradient padding R-1


def test_conv_narrow_register_tiled_forward(be):
    """Wide images (OW >= 64) take the 4-pixels-per-lane narrow forward: tile tails in x and y, every filter size, reflect and
    zero borders, the channel-split mode, and the data gradient of <= 4-input-channel layers that runs on it."""
    K.case_conv_fwd(be, 1, 18, 0, 9, 70, 3, 7, 1, 3, K.PAD_REFLECT, act=K.O.ACT_TANH)       # T head shape, 5 channel rounds
    K.case_conv_fwd(be, 2, 9, 0, 10, 130, 2, 3, 1, 1, K.PAD_ZERO, act=K.O.ACT_NONE)          # STN offset head, 2 tiles in x
    K.case_conv_fwd(be, 1, 40, 0, 6, 66, 1, 4, 1, 1, K.PAD_ZERO, act=K.O.ACT_NONE)           # k4: output one narrower; split mode
    K.case_conv_fwd(be, 1, 5, 0, 5, 64, 4, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_LRELU)
    K.case_conv_bwd_data(be, 1, 3, 0, 9, 70, 16, 7, 1, 3, K.PAD_REFLECT)                     # T stem data gradient
    be.lib.tune(19, 0)
    try:
        K.case_conv_fwd(be, 1, 18, 0, 9, 70, 3, 7, 1, 3, K.PAD_REFLECT, act=K.O.ACT_TANH)   # the one-pixel-per-lane kernel
    finally:
        be.lib.tune(19, 1)


def test_conv_narrow_register_tiled_weight_gradient(be):
    """<= 4-output-channel weight gradients on images at least 32 wide: thread = (channel, filter row), 4 pixels per step."""
    K.case_conv_bwd_weight(be, 1, 18, 0, 9, 40, 3, 7, 1, 3, K.PAD_REFLECT)       # T head shape: ragged channel chunk, 2 tiles in x/y
    K.case_conv_bwd_weight(be, 2, 40, 0, 6, 36, 1, 4, 1, 1, K.PAD_ZERO)          # D logits shape, 2 channel chunks
    K.case_conv_bwd_weight(be, 1, 33, 0, 10, 34, 2, 3, 1, 1, K.PAD_ZERO)         # STN offset head
    be.lib.tune(19, 0)
    try:
        K.case_conv_bwd_weight(be, 1, 18, 0, 9, 40, 3, 7, 1, 3, K.PAD_REFLECT)   # first-generation kernel, same shape
    finally:
        be.lib.tune(19, 1)


def test_conv_narrow_channel_split(be):
    """Narrow forward with few output tiles and no activation: channel ranges meet in y through atomics."""
    K.case_conv_fwd(be, 2, 40, 0, 9, 10, 1, 4, 1, 1, K.PAD_ZERO, act=K.O.ACT_NONE)              # D logit conv, 3 ranges
    K.case_conv_fwd(be, 1, 70, 0, 6, 6, 2, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_NONE, bias=False)  # ragged last range


def test_conv_fwd_acts_and_linear(be):
    K.case_conv_fwd(be, 2, 16, 0, 6, 6, 3, 7, 1, 3, K.PAD_REFLECT, act=K.O.ACT_TANH)
    K.case_conv_fwd(be, 3, 64, 0, 1, 1, 20, 1, 1, 0, K.PAD_ZERO, act=K.O.ACT_RELU)
    K.case_conv_fwd(be, 1, 8, 0, 2, 2, 8, 3, 1, 1, K.PAD_REFLECT, act=K.O.ACT_NONE, bias=False)


@pytest.mark.parametrize("R,op", [(3, 1), (4, 0)])
def test_conv_transpose_fwd(be, R, op):
    K.case_conv_transpose_fwd(be, 2, 16, 12, 5, 6, R, op)


# hot-path layer shapes at batch 1-2: multi-tile grids, many reduction stages
@pytest.mark.parametrize("C0,C1,Kc,R,stride,pad,pm,HW", [
    (256, 0, 256, 3, 1, 1, K.PAD_REFLECT, 32),   # T resblock conv (spatial reduced so the fp64 oracle stays fast)
    (32, 0, 32, 3, 1, 1, K.PAD_REFLECT, 64),     # STN full-res resblock conv
    (64, 32, 32, 3, 1, 1, K.PAD_ZERO, 48),       # STN up_1 (concat 64+32)
    (3, 3, 64, 4, 2, 1, K.PAD_ZERO, 64),         # D layer 1
    (128, 0, 256, 4, 2, 1, K.PAD_ZERO, 32),      # D layer 3
    (64, 0, 3, 7, 1, 3, K.PAD_REFLECT, 40),      # T head
])
def test_conv_hot_shapes(be, C0, C1, Kc, R, stride, pad, pm, HW):
    K.case_conv_fwd(be, 2, C0, C1, HW, HW, Kc, R, stride, pad, pm, act=K.O.ACT_RELU)
    if not (pm == K.PAD_REFLECT and C1):
        K.case_conv_bwd_data(be, 2, C0, C1, HW, HW, Kc, R, stride, pad, pm)
    K.case_conv_bwd_weight(be, 2, C0, C1, HW, HW, Kc, R, stride, pad, pm)


@pytest.mark.parametrize("H,W", [(2, 2), (4, 4), (9, 7), (31, 31), (64, 64), (72, 96), (128, 128), (144, 192), (200, 200), (224, 224), (244, 252), (256, 256), (300, 300)])
@pytest.mark.parametrize("act", [K.O.ACT_NONE, K.O.ACT_RELU, K.O.ACT_LRELU])
def test_instnorm(be, H, W, act):
    K.case_instnorm(be, 2, 3, H, W, act, residual=(act == K.O.ACT_NONE))


def test_pointwise(be):
    K.case_pointwise(be)


def test_concat_pieces_and_add2(be):
    K.case_concat_and_add(be)


def test_crop_flip_normalize(be):
    K.case_crop_flip_normalize(be)


def test_dropout(be):
    K.case_dropout(be, n=1 << 20)


def test_losses(be):
    K.case_losses(be)


def test_adam(be):
    K.case_adam(be, n=100003)


@pytest.mark.parametrize("variant", [4, 3])
def test_conv_split16_matrix_pipe(be, variant):
    """3x3 stride-1 layers with >= 128 output channels on the bf16 MFMA with three-way split operands (conv_split16.hip): padded,
    channel-blocked split planes; halo staged once per 16-channel chunk; weight-stage ring; zero and reflect padding; one and
    several row tiles per image, both 128-channel halves, 32- and 64-pixel rows."""
    # 4 = fp16 x 3 products (two scaled fp16 planes; the default), 3 = bf16 x 6 products
    be.lib.tune(21, variant)
    try:
        K.case_conv_split16(be, 2, 32, 8, 32, 128, K.PAD_REFLECT, dgrad=False)
        K.case_conv_split16(be, 1, 16, 16, 32, 256, K.PAD_ZERO, dgrad=False)
        K.case_conv_split16(be, 1, 16, 8, 64, 128, K.PAD_REFLECT, dgrad=False)
        K.case_conv_split16(be, 1, 48, 4, 128, 128, K.PAD_REFLECT, dgrad=False)      # 128-pixel rows: two rows per tile, 3 chunks
        K.case_conv_split16(be, 1, 16, 4, 256, 128, K.PAD_REFLECT, dgrad=False)      # 256-pixel rows: one row per tile (fp16 form only)
        K.case_conv_split16(be, 1, 128, 8, 32, 128, K.PAD_ZERO, dgrad=False)         # one tile, 8 chunks: reduction cut into two slabs
    finally:
        be.lib.tune(21, 4)


@pytest.mark.parametrize("variant", [4, 3])
def test_conv_split16_reflect_data_gradient(be, variant):
    """Data gradient of a reflect-padded 3x3 layer on the split-16 kernel: the folded border rows / slots written by the split
    pass are selected by address for (row 1, last filter row), (row H-2, first filter row) and the same in x; tiles that hold
    both special rows, only one, or none; and the zero-padded data gradient."""
    be.lib.tune(21, variant)
    try:
        K.case_conv_split16(be, 1, 128, 8, 32, 16, K.PAD_REFLECT, dgrad=True)
        K.case_conv_split16(be, 2, 128, 16, 32, 32, K.PAD_REFLECT, dgrad=True)
        K.case_conv_split16(be, 1, 128, 12, 64, 16, K.PAD_REFLECT, dgrad=True)
        K.case_conv_split16(be, 1, 256, 8, 32, 16, K.PAD_ZERO, dgrad=True)
        K.case_conv_split16(be, 1, 128, 4, 128, 16, K.PAD_REFLECT, dgrad=True)       # 128-pixel rows (fits the LDS in the fp16 form only)
    finally:
        be.lib.tune(21, 4)


def test_conv_split16_gy_split_once_for_both_gradients(be):
    """nemar_conv_extras.gy_planes_out / .src2_planes: the data-gradient call's split pass writes both operand layouts of gy (reflect
    fold rows and zero padding; plane rows rounded up to 4: two zero rows below a 6-row image; 32-, 64- and 128-pixel rows)."""
    K.case_conv_split16_dual_gy(be, 1, 128, 8, 32, 128, K.PAD_REFLECT)
    K.case_conv_split16_dual_gy(be, 2, 128, 4, 64, 192, K.PAD_ZERO)
    K.case_conv_split16_dual_gy(be, 1, 128, 6, 128, 128, K.PAD_REFLECT)


def test_conv_split16_weight_gradient(be):
    """Weight gradient of the wide 3x3 layers on the 16-bit matrix pipe (conv_split16_wgrad.hip): one copy of the gy planes (the
    shifted operands built in registers: v_permlane32_swap + v_alignbit) and padded x planes in tile order, nine taps per workgroup, pixel slabs summed in order; reflect and zero padding, one and several
    slabs per image, 2 and 3 chunks per row, rectangular channel counts."""
    K.case_conv_split16_wgrad(be, 1, 128, 8, 8, 128, K.PAD_REFLECT)
    K.case_conv_split16_wgrad(be, 2, 128, 8, 16, 192, K.PAD_ZERO)
    K.case_conv_split16_wgrad(be, 1, 192, 4, 24, 128, K.PAD_REFLECT)
    K.case_conv_split16_wgrad(be, 1, 128, 6, 16, 128, K.PAD_ZERO)                  # 6 rows -> 8 plane rows (two of zeros)
    K.case_conv_split16_wgrad(be, 2, 128, 8, 16, 128, K.PAD_ZERO, R=4)             # 4x4: 16 taps, gy 7x15, four shifts
    K.case_conv_split16_wgrad(be, 1, 128, 5, 32, 192, K.PAD_ZERO, R=4)             # gy 4x31
    be.lib.tune(34, 0)             # first generation: KS shifted copies of the gy planes in HBM
    try:
        K.case_conv_split16_wgrad(be, 1, 192, 4, 24, 128, K.PAD_REFLECT)
        K.case_conv_split16_wgrad(be, 2, 128, 8, 16, 128, K.PAD_ZERO, R=4)
    finally:
        be.lib.tune(34, 1)
    be.lib.tune(38, 0)             # the 3x3 kernel staged by LDS-DMA (the default stages through registers: DESIGN.md 4g)
    try:
        K.case_conv_split16_wgrad(be, 1, 128, 8, 8, 128, K.PAD_REFLECT)
        K.case_conv_split16_wgrad(be, 2, 128, 8, 16, 192, K.PAD_ZERO)
        K.case_conv_split16_wgrad(be, 1, 192, 4, 24, 128, K.PAD_REFLECT)
        K.case_conv_split16_wgrad(be, 1, 128, 6, 16, 128, K.PAD_ZERO)
    finally:
        be.lib.tune(38, 1)


def test_absmax_and_hint(be):
    K.case_absmax_and_hint(be)


@pytest.mark.parametrize("variant", [4, 3])
def test_conv_split16_4x4_layers(be, variant):
    """The discriminator's 4x4 / stride 1 / pad 1 layers on the split-16 kernel: computed on the input-sized domain with the last
    output row / column masked (forward), and as a full correlation of the (H-1) x (W-1) gradient with the flipped taps (data
    gradient, source offset 2); 16 taps per 16-channel chunk."""
    be.lib.tune(21, variant)
    try:
        K.case_conv_split16(be, 1, 32, 8, 32, 128, K.PAD_ZERO, dgrad=False, R=4)
        K.case_conv_split16(be, 2, 16, 16, 32, 256, K.PAD_ZERO, dgrad=False, R=4)
        K.case_conv_split16(be, 1, 128, 8, 32, 32, K.PAD_ZERO, dgrad=True, R=4)
        K.case_conv_split16(be, 2, 256, 16, 32, 16, K.PAD_ZERO, dgrad=True, R=4)
    finally:
        be.lib.tune(21, 4)


from test_kernels_emu import S16G_FWD, S16G_DGRAD, S16G_WGRAD


@pytest.mark.gpu
@pytest.mark.parametrize("case", S16G_FWD)
def test_conv_s16g_forward(be, case):
    K.case_conv_s16g_fwd(be, *case)


@pytest.mark.gpu
def test_conv_s16g_forward_act_bias(be):
    K.case_conv_s16g_fwd(be, 1, 16, 0, 4, 32, 32, 3, 1, 1, K.PAD_ZERO, act=2, bias=True)
    K.case_conv_s16g_fwd(be, 1, 16, 0, 4, 32, 32, 3, 1, 1, K.PAD_ZERO, act=0, bias=False)


@pytest.mark.gpu
def test_conv_s16g_dynamic_range(be):
    """Adversarial magnitudes through the C ABI: per-sample 1 : 1e-6 : 1e5, channel blocks 1e7 apart in either order, all-zero."""
    K.case_conv_s16g_fwd(be, 3, 32, 0, 4, 32, 32, 3, 1, 1, K.PAD_ZERO, xscale=[1.0, 1e-6, 1e5])
    chan = np.concatenate([np.full(16, 1e4), np.full(16, 1e-3)])
    K.case_conv_s16g_fwd(be, 1, 32, 0, 4, 32, 32, 3, 1, 1, K.PAD_ZERO, xscale=np.stack([chan]))
    K.case_conv_s16g_fwd(be, 1, 32, 0, 4, 32, 32, 3, 1, 1, K.PAD_ZERO, xscale=np.stack([chan[::-1]]))
    K.case_conv_s16g_fwd(be, 1, 16, 0, 4, 32, 32, 3, 1, 1, K.PAD_ZERO, xscale=[0.0], bias=False)
    K.case_conv_s16g_fwd(be, 8, 64, 0, 64, 64, 128, 3, 2, 1, K.PAD_ZERO, xscale=10.0 ** np.arange(-4, 4))      # a real-sized layer


@pytest.mark.gpu
@pytest.mark.parametrize("case", S16G_DGRAD)
def test_conv_s16g_data_gradient(be, case):
    K.case_conv_s16g_bwd_data(be, *case)


S16G_DGRAD_FUSED = [
    # N, C0, C1, H,  W,  K,  R, stride, pad          (C0 = the output rows: whole 32 / 64-row blocks; even extents)
    (1, 64, 0, 8, 64, 32, 3, 2, 1),                     # 64-row tile, classes of 1 / 2 / 2 / 4 taps, two chunks
    (2, 32, 0, 10, 72, 16, 4, 2, 1),                    # 32-row tile, 4x4: four classes of four taps, ragged tile (5 x 36 class pixels)
    (1, 128, 0, 6, 128, 48, 3, 2, 1),                   # two 64-row blocks, three chunks, tiles of 2 x 64 class pixels (last row of tiles ragged)
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", S16G_DGRAD_FUSED)
def test_conv_s16g_data_gradient_class_fused(be, case):
    """Stride-2 data gradients with the four output-parity classes in ONE workgroup per tile (conv_s16g.hip, CF): one converted halo, four
    accumulator sets, column-parity pairs stored as 8-byte words.  tune(42, 2): a problem that does not fuse fails the route check."""
    K.case_conv_s16g_bwd_data(be, *case, cf=2)
    K.case_conv_s16g_bwd_data(be, *case, cf=0)           # (one workgroup per class, the rounds 3-5 form: still there for odd extents)


@pytest.mark.gpu
@pytest.mark.parametrize("R,op", [(3, 1), (4, 0)])
def test_conv_s16g_transpose_forward_class_fused(be, R, op):
    with K.s16g_route(be, cf=2):
        K.case_conv_transpose_fwd(be, 2, 32, 64, 8, 32, R, op)                       # bias + ReLU in the paired epilogue
        assert be.lib.last_route() == 3
        K.case_conv_transpose_fwd(be, 1, 16, 32, 5, 36, R, op, act=K.O.ACT_LRELU)    # 32-row tile, ragged class tile
        assert be.lib.last_route() == 3


@pytest.mark.gpu
def test_conv_s16g_data_gradient_skip_first_source(be):
    K.case_conv_s16g_bwd_data(be, 1, 16, 16, 4, 32, 24, 3, 1, 1, skip0=True)


@pytest.mark.gpu
@pytest.mark.parametrize("R,op", [(3, 1), (4, 0)])
def test_conv_s16g_transpose_forward(be, R, op):
    with K.s16g_route(be):
        K.case_conv_transpose_fwd(be, 2, 32, 24, 8, 32, R, op)
        assert be.lib.last_route() == 3


@pytest.mark.gpu
@pytest.mark.parametrize("case", S16G_WGRAD)
def test_conv_s16g_weight_gradient(be, case):
    K.case_conv_s16g_bwd_weight(be, *case)


@pytest.mark.gpu
def test_conv_s16g_weight_gradient_row_scales(be):
    K.case_conv_s16g_bwd_weight(be, 1, 64, 0, 4, 32, 64, 3, 1, 1, K.PAD_ZERO, gscale=10.0 ** np.linspace(-6, 6, 64))
    K.case_conv_s16g_bwd_weight(be, 8, 64, 0, 64, 64, 128, 3, 2, 1, K.PAD_ZERO, gscale=10.0 ** np.linspace(-5, 5, 128))     # a real-sized layer


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["samples", "outlier", "zero", "inf", "channels", "subnormal"])
def test_conv_split16_dynamic_range(be, what):
    """The fp16 x 3 route of the wide layers on adversarial magnitudes (per-sample scales): see kernel_cases."""
    K.case_conv_split16_dynamic_range(be, what, N=3, C=128, H=16, W=32, K=128)


@pytest.mark.gpu
@pytest.mark.parametrize("act,res,drop", [(1, False, 0.5), (0, True, 0.0), (2, False, 0.0), (1, True, 0.0)])
def test_instnorm_planes(be, act, res, drop):
    K.case_instnorm_planes(be, act, res, drop)


@pytest.mark.gpu
def test_step_params_in_device_memory(be):
    K.case_step_params_in_device_memory(be)


def test_conv_ex_per_call_side_inputs(be):
    K.case_conv_ex(be)


def test_conv_from_producer_planes(be):
    K.case_conv_from_producer_planes(be)


@pytest.mark.gpu
@pytest.mark.parametrize("pad_mode,act,drop", [(K.PAD_REFLECT, 1, 0.5), (K.PAD_REFLECT, 0, 0.0), (K.PAD_ZERO, 2, 0.0)])
def test_resblock_planes_chain(be, pad_mode, act, drop):
    """Round 6: producer-written operand planes for all three calls of a wide layer + the data gradient's fused epilogue (kernel_cases)."""
    K.case_resblock_planes_chain(be, pad_mode, act, drop)


@pytest.mark.gpu
def test_resblock_planes_chain_xcd_order(be):
    """64 (sample, 8-channel group) units = one whole round of the XCD-aware workgroup order of the two producers (norm_planes.hip
    np_unit_of_block: the eight groups of a 64-channel block on consecutive workgroups of ONE XCD); H == 5 rows: the reflect fold's border
    rows 2 and H - 3 coincide."""
    K.case_resblock_planes_chain(be, K.PAD_REFLECT, 1, 0.5, N=4, C=128, H=8, W=32)
    K.case_resblock_planes_chain(be, K.PAD_REFLECT, 2, 0.0, N=1, C=64, H=5, W=16, producers_only=True)

@pytest.mark.gpu
def test_resblock_planes_chain_bench_shape(be):
    """... at the residual blocks' own plane size (64 x 64: the 1024-thread workgroups are full, the LDS tiles at their largest)"""
    K.case_resblock_planes_chain(be, K.PAD_REFLECT, 1, 0.5, N=2, C=128, H=64, W=64)


@pytest.mark.gpu
@pytest.mark.parametrize("mbl", [1, 4])
def test_conv_s16g_channel_blocks_per_workgroup(be, mbl):
    """Round 6: one s16g_kernel workgroup runs the taps of four 64-channel blocks on one converted halo (two weight regions in LDS,
    four accumulator sets) where the layer has a multiple of 256 output rows and 128-pixel tiles.  Same results as one workgroup per block (mbl = 1) within the kernel's bound: forward with several chunks and
    a rescale between them, 128 and 256 output channels, stride 1 / 2 / 4x4, data gradient (a forward over the transposed weights) and a
    transposed convolution's parity classes."""
    chan = 10.0 ** np.linspace(3, -4, 48)
    K.case_conv_s16g_fwd(be, 1, 48, 0, 4, 32, 256, 3, 1, 1, K.PAD_ZERO, act=2, xscale=np.stack([chan]), mbl=mbl)       # chunks shrink: no rescale
    K.case_conv_s16g_fwd(be, 1, 48, 0, 4, 32, 256, 3, 1, 1, K.PAD_ZERO, xscale=np.stack([chan[::-1]]), mbl=mbl)        # chunks grow: every block's accumulators rescale
    K.case_conv_s16g_fwd(be, 2, 16, 16, 8, 64, 128, 3, 2, 1, K.PAD_ZERO, act=1, mbl=mbl)                               # stride 2, two sources
    K.case_conv_s16g_fwd(be, 1, 32, 0, 8, 64, 128, 4, 2, 1, K.PAD_ZERO, mbl=mbl)                                       # 4x4 stride 2 (discriminator)
    K.case_conv_s16g_fwd(be, 1, 32, 0, 8, 64, 256, 3, 2, 1, K.PAD_ZERO, act=1, mbl=mbl)                                # the translation net's second down-sampling layer (four blocks)
    K.case_conv_s16g_fwd(be, 1, 16, 0, 8, 64, 512, 4, 2, 1, K.PAD_ZERO, mbl=mbl)                                       # eight blocks: two workgroups of four
    K.case_conv_s16g_fwd(be, 1, 32, 0, 6, 40, 192, 3, 1, 1, K.PAD_REFLECT, mbl=mbl)                                    # 3 blocks: falls back to fewer per workgroup
    K.case_conv_s16g_bwd_data(be, 1, 256, 0, 4, 32, 32, 3, 1, 1, mbl=mbl)                                              # data gradient: 256 "output" channels
    K.case_conv_s16g_bwd_data(be, 1, 128, 0, 8, 64, 32, 3, 2, 1, mbl=mbl)                                              # stride-2 data gradient: parity classes
    # ... and at the translation net's own down-sampling shape (64 -> 128, stride 2, four chunks, two blocks)
    K.case_conv_s16g_fwd(be, 2, 64, 0, 64, 64, 128, 3, 2, 1, K.PAD_ZERO, act=1, mbl=mbl)


@pytest.mark.gpu
def test_producer_max_words(be):
    K.case_producer_max_words(be)


@pytest.mark.gpu
def test_conv_s16g_reflect_data_gradient(be):
    """Reflect-padded stride-1 layers outside the wide residual blocks (the registration net's 32 / 64-channel ResnetBlocks): data
    gradient of the padded input on the general 16-bit-pipe kernel, then the fold of the mirrored border."""
    K.case_conv_s16g_bwd_data(be, 2, 32, 0, 8, 32, 32, 3, 1, 1, pad_mode=K.PAD_REFLECT)
    K.case_conv_s16g_bwd_data(be, 1, 64, 0, 6, 40, 48, 3, 1, 1, pad_mode=K.PAD_REFLECT)
    K.case_conv_s16g_bwd_data(be, 2, 32, 0, 8, 32, 32, 3, 1, 1, pad_mode=K.PAD_REFLECT, seed=7, addend=True)      # + a skip gradient in the fold pass


# ---- 7x7 stem / head layers on the 16-bit matrix pipe (csrc/conv_k7.hip) ----
@pytest.mark.parametrize("case", [
    (2, 3, 12, 40, 64, K.PAD_REFLECT), (1, 3, 70, 32, 32, K.PAD_ZERO), (2, 64, 10, 36, 3, K.PAD_REFLECT), (1, 32, 8, 8, 2, K.PAD_ZERO),
    (1, 32, 6, 64, 3, K.PAD_REFLECT), (1, 64, 5, 32, 2, K.PAD_ZERO),
    (2, 3, 128, 128, 64, K.PAD_REFLECT), (2, 64, 128, 96, 3, K.PAD_REFLECT),
])
def test_conv_k7_weight_gradient(be, case):
    N, C, H, W, Kc, pm = case
    K.case_conv_k7_bwd_weight(be, N, C, H, W, Kc, pm)


def test_conv_k7_weight_gradient_scales(be):
    K.case_conv_k7_bwd_weight(be, 3, 3, 9, 32, 64, K.PAD_REFLECT, xscale=[1.0, 1e-4, 1e3])
    K.case_conv_k7_bwd_weight(be, 2, 64, 9, 32, 3, K.PAD_REFLECT, xscale=[1e-3, 1e2])


@pytest.mark.parametrize("case", [
    (2, 3, 9, 70, 64, K.PAD_REFLECT, K.O.ACT_NONE),      # stem: ragged tile rows (9 = 2 x 4 + 1) and columns (70 = 64 + 6)
    (1, 4, 8, 16, 32, K.PAD_ZERO, K.O.ACT_RELU),         # 4 input channels (28 (c, dy) pairs: the 4th operand word), 32 outputs, zero border
    (1, 1, 4, 8, 96, K.PAD_REFLECT, K.O.ACT_LRELU),      # one input channel, 96 outputs (two channel blocks, the second half full)
    (2, 3, 128, 192, 64, K.PAD_REFLECT, K.O.ACT_NONE),
])
def test_conv_k7_forward(be, case):
    N, C, H, W, Kc, pm, act = case
    K.case_conv_k7_fwd(be, N, C, H, W, Kc, pm, act=act)


def test_conv_k7_forward_tile_scales(be):
    K.case_conv_k7_fwd(be, 3, 3, 8, 64, 64, K.PAD_REFLECT, xscale=[1.0, 1e-5, 1e4])


@pytest.mark.parametrize("case", [
    (2, 64, 9, 40, 3, K.PAD_REFLECT), (1, 32, 8, 16, 2, K.PAD_ZERO), (1, 64, 4, 70, 1, K.PAD_REFLECT),
    (1, 32, 8, 64, 3, K.PAD_REFLECT),       # in-kernel fold: first and last row tile, one column tile holding both mirrored borders
    (1, 64, 12, 128, 2, K.PAD_REFLECT),     # ... an interior row tile, the two borders in different column tiles
    (2, 64, 128, 96, 3, K.PAD_REFLECT),
])
def test_conv_k7_data_gradient(be, case):
    N, C, H, W, Kc, pm = case
    K.case_conv_k7_bwd_data(be, N, C, H, W, Kc, pm)


@pytest.mark.parametrize("case", [
    (2, 32, 8, 24, 3, K.PAD_REFLECT, K.O.ACT_TANH),      # the RGB head: 32 -> 3, reflect border, tanh in the shift-sum pass
    (1, 16, 5, 40, 1, K.PAD_ZERO, K.O.ACT_NONE),         # one output channel, zero border
    (1, 48, 6, 28, 4, K.PAD_REFLECT, K.O.ACT_RELU),      # four outputs: all 32 (k, dx) rows in use
    (2, 64, 128, 128, 3, K.PAD_REFLECT, K.O.ACT_TANH),
])
def test_conv_k7_many_to_few_forward(be, case):
    """7x7 layers with <= 4 OUTPUT channels: vertical 7-tap convolution with (k, dx) pseudo-channels on the general 16-bit-pipe kernel +
    horizontal shift-sum (csrc/conv.hip k7_mf_*), against the float64 oracle."""
    N, C, H, W, Kc, pm, act = case
    K.case_conv_fwd(be, N, C, 0, H, W, Kc, 7, 1, 3, pm, act=act)
    assert be.lib.last_route() == K.ROUTE_K7


@pytest.mark.parametrize("case", [
    (2, 3, 9, 20, 32, K.PAD_REFLECT), (1, 2, 6, 28, 16, K.PAD_ZERO), (1, 4, 12, 24, 48, K.PAD_REFLECT),
    (2, 3, 96, 128, 64, K.PAD_REFLECT),
])
def test_conv_k7_many_to_few_data_gradient(be, case):
    """data gradient of a 7x7 layer with <= 4 INPUT channels (the stem): the same two passes with flipped, transposed weights; reflect
    border folded inside the shift-sum pass"""
    N, C, H, W, Kc, pm = case
    K.case_conv_bwd_data(be, N, C, 0, H, W, Kc, 7, 1, 3, pm)
    assert be.lib.last_route() == K.ROUTE_K7
