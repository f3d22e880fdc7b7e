"""Backend-agnostic kernel-vs-oracle test bodies (see tests/backends.py).  Each case drives the C-ABI
exactly as include/nemar_hip.h declares it and compares with oracle/ops_np.py evaluated in float64."""
import numpy as np

from oracle import ops_np as O

GRID_EXPLICIT, GRID_UNET, GRID_AFFINE = 0, 1, 2


def _assert_close(got, want, atol, rtol=0.0, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want)
    lim = atol + rtol * np.abs(want)
    if not np.all(err <= lim):
        i = np.unravel_index(np.argmax(err - lim), err.shape)
        raise AssertionError("%s: max|err|=%.3e at %s (got %.6g want %.6g), atol=%g rtol=%g" %
                             (what, err.max(), i, got[i], want[i], atol, rtol))


# ------------------------------------------------------------------------------------------------
def case_grid_sample(be, mode, N, C, H, W, Ho, Wo, scale, seed=0, need_gin=True, accumulate=False):
    rng = np.random.default_rng(seed)
    inp = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    if mode == GRID_UNET:
        src = (rng.standard_normal((N, 2, Ho, Wo)) * scale).astype(np.float32)
        grid = O.unet_grid(src.astype(np.float64))
    elif mode == GRID_AFFINE:
        src = (rng.standard_normal((N, 6)) * scale).astype(np.float32)
        grid = O.affine_grid(O.affine_theta(src.astype(np.float64)), Ho, Wo)
    else:
        src = (O.unet_grid((rng.standard_normal((N, 2, Ho, Wo)) * scale))).astype(np.float32)
        grid = src.astype(np.float64)
    gout = rng.standard_normal((N, C, Ho, Wo)).astype(np.float32)
    want_out = O.grid_sample_fwd(inp.astype(np.float64), grid)
    want_gin, want_gg = O.grid_sample_bwd(inp.astype(np.float64), grid, gout.astype(np.float64))
    if mode == GRID_UNET:
        want_gsrc = want_gg.transpose(0, 3, 1, 2)
    elif mode == GRID_AFFINE:
        _, want_gsrc = O.affine_warp_bwd(inp.astype(np.float64), src.astype(np.float64), gout.astype(np.float64), Ho, Wo)
    else:
        want_gsrc = want_gg

    d_in, d_src, d_gout = be.dev(inp), be.dev(src), be.dev(gout)
    d_out = be.full((N, C, Ho, Wo), np.nan)
    be.lib.grid_sample_fwd(be.ptr(d_in), be.ptr(d_src), mode, be.ptr(d_out), N, C, H, W, Ho, Wo, be.stream)
    # coordinates are O(W) in fp32 => ~W*6e-8 px of jitter times the local image slope (<=2)
    _assert_close(be.np(d_out), want_out, atol=4e-6 * max(H, W, 16), what="grid_sample_fwd")

    base = 0.5 if accumulate else 0.0
    d_gin = be.full((N, C, H, W), base if accumulate else np.nan) if need_gin else None
    d_gsrc = be.full(src.shape, base if accumulate else np.nan)
    be.lib.grid_sample_bwd(be.ptr(d_in), be.ptr(d_src), mode, be.ptr(d_gout), be.ptr(d_gin), int(accumulate),
                           be.ptr(d_gsrc), int(accumulate), N, C, H, W, Ho, Wo, be.stream)
    tol = 4e-6 * max(H, W, 16)
    if need_gin:
        _assert_close(be.np(d_gin), want_gin + base, atol=tol * 4, what="grid_sample_bwd gin")
    gs_tol = tol * max(H, W) * C
    if mode == GRID_AFFINE:
        gs_tol *= Ho * Wo / 16.0
    _assert_close(be.np(d_gsrc), want_gsrc + base, atol=gs_tol, rtol=1e-4, what="grid_sample_bwd ggrid")


def case_smoothness(be, N, H, W, Ci, alpha, factor=1.0, seed=0, accumulate=False):
    rng = np.random.default_rng(seed)
    d = (rng.standard_normal((N, 2, H, W)) * 0.1).astype(np.float32)
    # plant exact ties so sign(0) = 0 is exercised
    d[:, :, 0, 0] = d[:, :, 1, 0]
    d[:, :, 0, 1] = d[:, :, 0, 0]
    img = rng.uniform(-1, 1, (N, Ci, H, W)).astype(np.float32) if Ci else None
    want = factor * O.smoothness_fwd(d.astype(np.float64), None if img is None else img.astype(np.float64), alpha)
    want_g = factor * 0.75 * O.smoothness_bwd(d.astype(np.float64), None if img is None else img.astype(np.float64), alpha)
    d_d = be.dev(d)
    d_img = be.dev(img) if img is not None else None
    ws_bytes = be.lib.smoothness_workspace(N, H, W)
    ws = be.bytes_buf(ws_bytes)
    base = 0.25 if accumulate else 0.0
    loss = be.full((1,), base if accumulate else np.nan)
    be.lib.smoothness_fwd(be.ptr(d_d), be.ptr(d_img), Ci, alpha, factor, be.ptr(loss), int(accumulate),
                          be.ptr(ws), ws_bytes, N, H, W, be.stream)
    _assert_close(be.np(loss), [want + base], atol=1e-6, rtol=2e-5, what="smoothness_fwd")
    gscale = be.dev(np.array([0.75], dtype=np.float32))
    gd = be.full((N, 2, H, W), base if accumulate else np.nan)
    be.lib.smoothness_bwd(be.ptr(d_d), be.ptr(d_img), Ci, alpha, be.ptr(gscale), factor, be.ptr(gd), int(accumulate),
                          N, H, W, be.stream)
    _assert_close(be.np(gd), want_g + base, atol=1e-7 + 1e-5 * np.abs(want_g).max(), what="smoothness_bwd")
