"""`-m gpu` tier: the real gfx950 library (nemar_amd/lib/libnemar_hip.so) through its C-ABI, against the
oracle, with the same bodies as the CPU/emulator tier plus hot-path sizes."""
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
