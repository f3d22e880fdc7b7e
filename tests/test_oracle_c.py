"""Pin the oracle's C restatement (oracle/c/nemar_ref.c) against the numpy oracle (which is itself pinned against torch
and against golden values of the reference)."""
import ctypes

import numpy as np
import pytest

from oracle import build_c
from oracle import ops_np as O


@pytest.fixture(scope="module")
def clib():
    lib = ctypes.CDLL(build_c.build())
    lib.ref_smoothness.restype = ctypes.c_double
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_c_unet_warp(clib):
    rng = np.random.default_rng(0)
    N, C, H, W = 2, 3, 14, 18
    img = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    off = (rng.standard_normal((N, 2, H, W)) * 0.1).astype(np.float32)
    go = rng.standard_normal((N, C, H, W)).astype(np.float32)
    out = np.empty_like(img)
    clib.ref_unet_warp_fwd(_p(img), _p(off), _p(out), N, C, H, W)
    grid = O.unet_grid(off)                                  # float32 coordinates, same floor() decisions
    np.testing.assert_allclose(out, O.grid_sample_fwd(img.astype(np.float64), grid), atol=1e-5)
    gin, goff = np.empty_like(img), np.empty_like(off)
    clib.ref_unet_warp_bwd(_p(img), _p(off), _p(go), _p(gin), _p(goff), N, C, H, W)
    want_gin, want_gg = O.grid_sample_bwd(img.astype(np.float64), grid, go.astype(np.float64))
    np.testing.assert_allclose(gin, want_gin, atol=2e-5)
    np.testing.assert_allclose(goff, want_gg.transpose(0, 3, 1, 2), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("alpha", [0.0, 1.7])
def test_c_smoothness(clib, alpha):
    rng = np.random.default_rng(1)
    N, H, W = 2, 9, 13
    d = (rng.standard_normal((N, 2, H, W)) * 0.1).astype(np.float32)
    img = rng.uniform(-1, 1, (N, 3, H, W)).astype(np.float32)
    gd = np.empty_like(d)
    loss = clib.ref_smoothness(_p(d), _p(img), 3, ctypes.c_float(alpha), _p(gd), N, H, W)
    np.testing.assert_allclose(loss, O.smoothness_fwd(d.astype(np.float64), img.astype(np.float64), alpha), rtol=1e-5)
    np.testing.assert_allclose(gd, O.smoothness_bwd(d.astype(np.float64), img.astype(np.float64), alpha), atol=1e-7,
                               rtol=1e-4)
