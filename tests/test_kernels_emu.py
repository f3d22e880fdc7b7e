"""CPU tier: the unmodified HIP kernel sources, compiled for the host SIMT emulator (tests/emu), checked
against the oracle.  This validates index arithmetic, tiling and wave-collective logic without a GPU;
the `-m gpu` tier (tests/test_kernels_gpu.py) runs the same bodies on the real gfx950 library."""
import pytest

import kernel_cases as K
from backends import EmuBackend


@pytest.fixture(scope="module")
def be(emu_lib):
    return EmuBackend(emu_lib)


@pytest.mark.parametrize("mode", [K.GRID_UNET, K.GRID_AFFINE, K.GRID_EXPLICIT])
@pytest.mark.parametrize("scale", [0.0, 0.05, 1.5])
def test_grid_sample(be, mode, scale):
    if mode == K.GRID_AFFINE and scale == 0.0:
        # exact-identity affine grids sample at integer texel centres, where d(out)/d(grid) is discontinuous
        # (left/right derivative of the bilinear kernel) - fp32 vs fp64 rounding picks different sides.
        scale = 0.01
    K.case_grid_sample(be, mode, N=2, C=3, H=12, W=16, Ho=12, Wo=16, scale=scale)


def test_grid_sample_ragged_and_resampled(be):
    K.case_grid_sample(be, K.GRID_UNET, N=1, C=1, H=7, W=9, Ho=7, Wo=9, scale=0.1)          # Wo % 4 != 0
    K.case_grid_sample(be, K.GRID_AFFINE, N=3, C=2, H=10, W=6, Ho=5, Wo=8, scale=0.2)       # Ho,Wo != H,W
    K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=8, W=8, Ho=8, Wo=8, scale=0.1, need_gin=False)
    K.case_grid_sample(be, K.GRID_UNET, N=2, C=3, H=8, W=8, Ho=8, Wo=8, scale=0.1, accumulate=True)
    K.case_grid_sample(be, K.GRID_AFFINE, N=2, C=3, H=8, W=8, Ho=8, Wo=8, scale=0.1, accumulate=True)


@pytest.mark.parametrize("Ci,alpha", [(0, 0.0), (3, 0.0), (3, 1.7), (1, 0.5)])
def test_smoothness(be, Ci, alpha):
    K.case_smoothness(be, N=2, H=9, W=13, Ci=Ci, alpha=alpha)


def test_smoothness_accumulate_factor(be):
    K.case_smoothness(be, N=1, H=2, W=2, Ci=3, alpha=0.9, factor=0.5, accumulate=True)
    K.case_smoothness(be, N=3, H=17, W=5, Ci=0, alpha=0.0, factor=0.25)
