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
    # The sampling coordinates are computed in float32 with the kernel's exact operation order, so that the
    # oracle makes the same floor() decisions (d out / d grid is discontinuous at integer coordinates);
    # everything downstream of the coordinates is evaluated in float64.
    f32 = np.float32
    if mode == GRID_UNET:
        src = (rng.standard_normal((N, 2, Ho, Wo)) * scale).astype(f32)
        grid = O.unet_grid(src)                                   # float32, fma-exact linspace + offsets
    elif mode == GRID_AFFINE:
        src = (rng.standard_normal((N, 6)) * scale).astype(f32)
        th = src + np.array([1, 0, 0, 0, 1, 0], dtype=f32)[None]
        xb = ((f32(2) * np.arange(Wo, dtype=f32) + f32(1)) / f32(Wo) - f32(1))[None, None, :]
        yb = ((f32(2) * np.arange(Ho, dtype=f32) + f32(1)) / f32(Ho) - f32(1))[None, :, None]
        T = lambda i: th[:, i][:, None, None]
        grid = np.stack([(T(0) * xb + T(1) * yb) + T(2), (T(3) * xb + T(4) * yb) + T(5)], axis=-1).astype(f32)
    else:
        src = O.unet_grid((rng.standard_normal((N, 2, Ho, Wo)) * scale).astype(f32))
        grid = src
    gout = rng.standard_normal((N, C, Ho, Wo)).astype(np.float32)
    want_out = O.grid_sample_fwd(inp.astype(np.float64), grid)
    want_gin, want_gg = O.grid_sample_bwd(inp.astype(np.float64), grid, gout.astype(np.float64))
    if mode == GRID_UNET:
        want_gsrc = want_gg.transpose(0, 3, 1, 2)
    elif mode == GRID_AFFINE:
        xs = (2.0 * np.arange(Wo) + 1.0) / Wo - 1.0
        ys = (2.0 * np.arange(Ho) + 1.0) / Ho - 1.0
        base = np.stack([np.broadcast_to(xs[None, :], (Ho, Wo)), np.broadcast_to(ys[:, None], (Ho, Wo)),
                         np.ones((Ho, Wo))], axis=-1)
        want_gsrc = np.einsum('nhwi,hwk->nik', want_gg, base).reshape(-1, 6)
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


# ------------------------------------------------------------------------------------------------
PAD_ZERO, PAD_REFLECT = 0, 1
_PM = {PAD_ZERO: 'zeros', PAD_REFLECT: 'reflect'}


def _ws(be, nbytes):
    return be.bytes_buf(nbytes), nbytes


def case_conv_fwd(be, N, C0, C1, H, W, K, R, stride, pad, pad_mode, act=O.ACT_NONE, bias=True, seed=0):
    rng = np.random.default_rng(seed)
    C = C0 + C1
    x = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((K, C, R, R)) / np.sqrt(C * R * R)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32) if bias else None
    want = O.act_fwd(O.conv2d_fwd(x.astype(np.float64), w.astype(np.float64),
                                  None if b is None else b.astype(np.float64), stride, pad, _PM[pad_mode]), act)
    OH, OW = want.shape[2:]
    d_x0 = be.dev(x[:, :C0])
    d_x1 = be.dev(x[:, C0:]) if C1 else None
    d_w, d_b = be.dev(w), (be.dev(b) if bias else None)
    d_y = be.full((N, K, OH, OW), np.nan)
    ws, wsb = _ws(be, be.lib.conv2d_fwd_workspace(K, C, R, R))
    be.lib.conv2d_fwd(be.ptr(d_x0), C0, be.ptr(d_x1), C1, be.ptr(d_w), be.ptr(d_b), be.ptr(d_y), N, H, W, K, R, R,
                      stride, pad, pad_mode, act, 0.2, be.ptr(ws), wsb, be.stream)
    _assert_close(be.np(d_y), want, atol=2e-5, rtol=2e-5, what="conv2d_fwd")


def case_conv_bwd_data(be, N, C0, C1, H, W, K, R, stride, pad, pad_mode, skip0=False, seed=0):
    rng = np.random.default_rng(seed)
    C = C0 + C1
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = np.zeros((N, C, H, W))
    w = (rng.standard_normal((K, C, R, R)) / np.sqrt(K * R * R)).astype(np.float32)
    gy = rng.standard_normal((N, K, OH, OW)).astype(np.float32)
    want, _, _ = O.conv2d_bwd(x, w.astype(np.float64), gy.astype(np.float64), stride, pad, _PM[pad_mode])
    d_gy, d_w = be.dev(gy), be.dev(w)
    d_g0 = None if (skip0 or C0 == 0) else be.full((N, C0, H, W), np.nan)
    d_g1 = be.full((N, C1, H, W), np.nan) if C1 else None
    ws, wsb = _ws(be, be.lib.conv2d_bwd_data_workspace(N, C, H, W, K, R, R, stride, pad, pad_mode))
    be.lib.conv2d_bwd_data(be.ptr(d_gy), be.ptr(d_w), None, 0, 0.0, be.ptr(d_g0), C0, be.ptr(d_g1), C1, N, H, W, K, OH,
                           OW, R, R, stride, pad, pad_mode, be.ptr(ws), wsb, be.stream)
    if d_g0 is not None:
        _assert_close(be.np(d_g0), want[:, :C0], atol=2e-5, rtol=2e-5, what="conv2d_bwd_data gx0")
    if d_g1 is not None:
        _assert_close(be.np(d_g1), want[:, C0:], atol=2e-5, rtol=2e-5, what="conv2d_bwd_data gx1")


def case_conv_transpose_fwd(be, N, Ci, Co, H, W, R, out_pad, act=O.ACT_RELU, seed=0):
    """ConvTranspose2d(Ci->Co, k=R, s=2, p=1, op) forward through the data-gradient entry point."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (N, Ci, H, W)).astype(np.float32)
    w = (rng.standard_normal((Ci, Co, R, R)) / np.sqrt(Ci * R * R)).astype(np.float32)
    b = rng.standard_normal(Co).astype(np.float32)
    want = O.act_fwd(O.conv_transpose2d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 2, 1,
                                            out_pad), act)
    Ho, Wo = want.shape[2:]
    d_x, d_w, d_b = be.dev(x), be.dev(w), be.dev(b)
    d_y = be.full((N, Co, Ho, Wo), np.nan)
    ws, wsb = _ws(be, be.lib.conv2d_bwd_data_workspace(N, Co, Ho, Wo, Ci, R, R, 2, 1, PAD_ZERO))
    be.lib.conv2d_bwd_data(be.ptr(d_x), be.ptr(d_w), be.ptr(d_b), act, 0.2, be.ptr(d_y), Co, None, 0, N, Ho, Wo, Ci, H,
                           W, R, R, 2, 1, PAD_ZERO, be.ptr(ws), wsb, be.stream)
    _assert_close(be.np(d_y), want, atol=2e-5, rtol=2e-5, what="conv_transpose2d_fwd")


def case_conv_bwd_weight(be, N, C0, C1, H, W, K, R, stride, pad, pad_mode, seed=0):
    rng = np.random.default_rng(seed)
    C = C0 + C1
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    gy = (rng.standard_normal((N, K, OH, OW)) / np.sqrt(N * OH * OW)).astype(np.float32)
    w = np.zeros((K, C, R, R))
    _, want_gw, want_gb = O.conv2d_bwd(x.astype(np.float64), w, gy.astype(np.float64), stride, pad, _PM[pad_mode])
    d_x0 = be.dev(x[:, :C0])
    d_x1 = be.dev(x[:, C0:]) if C1 else None
    d_gy = be.dev(gy)
    d_gw = be.full((K, C, R, R), 0.5)          # accumulate semantics
    be.lib.conv2d_bwd_weight(be.ptr(d_x0), C0, be.ptr(d_x1), C1, be.ptr(d_gy), be.ptr(d_gw), N, H, W, K, OH, OW, R, R,
                             stride, pad, pad_mode, be.stream)
    _assert_close(be.np(d_gw), want_gw + 0.5, atol=2e-5, rtol=2e-5, what="conv2d_bwd_weight")
    d_gb = be.full((K,), -0.25)
    be.lib.bias_grad(be.ptr(d_gy), be.ptr(d_gb), N, K, OH * OW, be.stream)
    _assert_close(be.np(d_gb), want_gb - 0.25, atol=2e-5, rtol=2e-5, what="bias_grad")
