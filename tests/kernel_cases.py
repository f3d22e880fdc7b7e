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
def case_grid_sample(be, mode, N, C, H, W, Ho, Wo, scale, seed=0, need_gin=True, accumulate=False, workspace=True,
                     atomic=False, smooth_px=None):
    """smooth_px = (shift_x, shift_y, wave) in pixels (GRID_UNET / GRID_EXPLICIT): a smooth LARGE deformation — a translation plus a
    slow sinusoid — on top of the `scale` noise: the regime in which the gather windows of the backward pass follow the field."""
    rng = np.random.default_rng(seed)
    inp = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    # The sampling coordinates are computed in float32 with the kernel's exact operation order, so that the
    # oracle makes the same floor() decisions (d out / d grid is discontinuous at integer coordinates);
    # everything downstream of the coordinates is evaluated in float64.
    f32 = np.float32
    def offsets():
        o = rng.standard_normal((N, 2, Ho, Wo)) * scale
        if smooth_px is not None:
            sx, sy, wave = smooth_px
            yy, xx = np.meshgrid(np.arange(Ho), np.arange(Wo), indexing='ij')
            o[:, 0] += (sx + wave * np.sin(2 * np.pi * yy / Ho + 0.3) * np.cos(2 * np.pi * xx / Wo)) * 2.0 / Wo
            o[:, 1] += (sy + wave * np.cos(2 * np.pi * xx / Wo + 0.7)) * 2.0 / Ho
        return o.astype(f32)

    if mode == GRID_UNET:
        src = offsets()
        grid = O.unet_grid(src)                                   # float32, fma-exact linspace + offsets
    elif mode == GRID_AFFINE:
        src = (rng.standard_normal((N, 6)) * scale).astype(f32)
        th = src + np.array([1, 0, 0, 0, 1, 0], dtype=f32)[None]
        xb = ((f32(2) * np.arange(Wo, dtype=f32) + f32(1)) / f32(Wo) - f32(1))[None, None, :]
        yb = ((f32(2) * np.arange(Ho, dtype=f32) + f32(1)) / f32(Ho) - f32(1))[None, :, None]
        T = lambda i: th[:, i][:, None, None]
        grid = np.stack([(T(0) * xb + T(1) * yb) + T(2), (T(3) * xb + T(4) * yb) + T(5)], axis=-1).astype(f32)
    else:
        src = O.unet_grid(offsets())
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
    # with the workspace: gather + fixed-point path (same-size, C <= 4), else the fp32-atomic kernels
    wsb = be.lib.grid_sample_bwd_workspace(N, C, H, W) if workspace else 0
    ws = be.bytes_buf(wsb) if workspace else None
    for rep in range(2 if workspace else 1):         # second call on the SAME workspace: it must have been returned all-zero
        if rep:
            d_gin = be.full((N, C, H, W), base if accumulate else np.nan) if need_gin else None
            d_gsrc = be.full(src.shape, base if accumulate else np.nan)
        be.lib.grid_sample_bwd(be.ptr(d_in), be.ptr(d_src), mode, be.ptr(d_gout), be.ptr(d_gin), int(accumulate),
                               be.ptr(d_gsrc), int(accumulate), N, C, H, W, Ho, Wo, be.ptr(ws), wsb, be.stream)
        if workspace:
            zb = be.lib.grid_sample_bwd_zeroed_bytes(N, C, H, W)
            assert not np.any(be.np(ws)[:zb // 4]), "grid_sample_bwd must return the accumulator part of its workspace zero-filled"
            if rep == 0:
                first = (None if d_gin is None else be.np(d_gin).copy(), be.np(d_gsrc).copy())
            elif not atomic and H == Ho and W == Wo and C <= 4:      # bitwise reproducible (gather + fixed-point path)
                assert d_gin is None or np.array_equal(be.np(d_gin), first[0], equal_nan=True)
                assert np.array_equal(be.np(d_gsrc), first[1], equal_nan=True)
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
    ws, wsb = _ws(be, be.lib.conv2d_fwd_workspace(N, H, W, K, C, R, R, stride, pad))
    be.lib.conv2d_fwd(be.ptr(d_x0), C0, be.ptr(d_x1), C1, be.ptr(d_w), be.ptr(d_b), be.ptr(d_y), N, H, W, K, R, R,
                      stride, pad, pad_mode, act, 0.2, be.ptr(ws), wsb, 0, be.stream)
    _assert_close(be.np(d_y), want, atol=2e-5, rtol=2e-5, what="conv2d_fwd")


def case_conv_bwd_data(be, N, C0, C1, H, W, K, R, stride, pad, pad_mode, skip0=False, seed=0, addend=False):
    """addend=True: nemar_conv2d_bwd_data_ex with nemar_conv_extras.addend on a layer whose data gradient ends with a fold pass
    (nemar_conv2d_bwd_data_addend_ok must say 1): gx0 = data gradient + addend."""
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
    if addend:
        import ctypes
        from nemar_amd._lib import ConvExtras
        assert C1 == 0 and not skip0
        assert be.lib.conv2d_bwd_data_addend_ok(N, C, H, W, K, R, R, stride, pad, pad_mode) == 1, "addend_ok"
        add = rng.standard_normal((N, C, H, W)).astype(np.float32)
        d_add = be.dev(add)
        e = ConvExtras()
        e.addend = be.ptr(d_add).value
        be.lib.conv2d_bwd_data_ex(be.ptr(d_gy), be.ptr(d_w), None, 0, 0.0, be.ptr(d_g0), C0, None, 0, N, H, W, K, OH, OW, R, R, stride, pad,
                                  pad_mode, be.ptr(ws), wsb, 0, be.stream, ctypes.byref(e))
        _assert_close(be.np(d_g0), want + add, atol=2e-5, rtol=2e-5, what="conv2d_bwd_data gx0 + addend")
        return
    be.lib.conv2d_bwd_data(be.ptr(d_gy), be.ptr(d_w), None, 0, 0.0, be.ptr(d_g0), C0, be.ptr(d_g1), C1, N, H, W, K, OH,
                           OW, R, R, stride, pad, pad_mode, be.ptr(ws), wsb, 0, be.stream)
    if d_g0 is not None:
        _assert_close(be.np(d_g0), want[:, :C0], atol=2e-5, rtol=2e-5, what="conv2d_bwd_data gx0")
    if d_g1 is not None:
        _assert_close(be.np(d_g1), want[:, C0:], atol=2e-5, rtol=2e-5, what="conv2d_bwd_data gx1")


class scratch_arena:
    """Register a scratch arena with the library for the duration of a block (nemar_set_scratch): the wide 3x3 stride-1
    layers then run on the split-16 matrix-pipe kernels (csrc/conv_split16.hip)."""

    def __init__(self, be, nbytes):
        self.be, self.buf, self.nbytes = be, be.bytes_buf(nbytes), nbytes

    def __enter__(self):
        self.be.lib.tune(23, 0)          # test shapes are far below the work threshold of the route
        self.be.lib.set_scratch(self.be.ptr(self.buf), self.nbytes)
        return self

    def __exit__(self, *a):
        self.be.sync()
        self.be.lib.set_scratch(None, 0)
        self.be.lib.tune(23, 2000)


def split16_scratch(be, *shape):
    """nemar_conv2d_scratch for a (small) test shape: with the work threshold of the route lifted"""
    be.lib.tune(23, 0)
    try:
        return be.lib.conv2d_scratch(*shape)
    finally:
        be.lib.tune(23, 2000)


class s16g_route:
    """Lift the work threshold of the general 16-bit-pipe route (csrc/conv_s16g.hip: in-kernel operand split) for small test shapes;
    `on=False` forces the exact-fp32 kernels instead."""

    def __init__(self, be, on=True, mbl=None, cf=None):
        self.be, self.on, self.mbl, self.cf = be, on, mbl, cf

    def __enter__(self):
        self.be.lib.tune(24, 1 if self.on else 0)
        self.be.lib.tune(25, 0)
        if self.mbl is not None:          # channel blocks per workgroup (round 6): force the grouping on the tests' few-tile shapes
            self.be.lib.tune(40, self.mbl)
            self.be.lib.tune(41, 0)
        if self.cf is not None:           # class-fused stride-2 data gradients (round 6): 0 off, 2 required (whatever the grid size)
            self.be.lib.tune(42, self.cf)
            self.be.lib.tune(41, 0)
        return self

    def __exit__(self, *a):
        self.be.sync()
        self.be.lib.tune(24, 1)
        self.be.lib.tune(25, 30)
        self.be.lib.tune(40, 4)
        self.be.lib.tune(41, 256)
        self.be.lib.tune(42, 0)


def case_conv_s16g_fwd(be, N, C0, C1, H, W, K, R, stride, pad, pad_mode, act=O.ACT_NONE, bias=True, seed=0, xscale=None, mbl=None):
    """Forward through nemar_conv2d_fwd on the general 16-bit-pipe route; `xscale` [N] or [N, C] multiplies the source per sample /
    per channel (dynamic-range cases: the block scale is per tile and per 16-channel chunk)."""
    rng = np.random.default_rng(seed)
    C = C0 + C1
    x = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    if xscale is not None:
        xs = np.asarray(xscale, dtype=np.float32)
        x = (x * xs.reshape(xs.shape + (1,) * (4 - xs.ndim))).astype(np.float32)
    w = (rng.standard_normal((K, C, R, R)) / np.sqrt(C * R * R)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32) if bias else None
    want = O.act_fwd(O.conv2d_fwd(x.astype(np.float64), w.astype(np.float64),
                                  None if b is None else b.astype(np.float64), stride, pad, _PM[pad_mode]), act)
    # error scale of an output = what an fp32 dot product of the same terms may lose: 2^-19 of sum |w| |x| (+ bias)
    mag = O.conv2d_fwd(np.abs(x).astype(np.float64), np.abs(w).astype(np.float64), None, stride, pad, _PM[pad_mode])
    OH, OW = want.shape[2:]
    d_x0 = be.dev(x[:, :C0])
    d_x1 = be.dev(x[:, C0:]) if C1 else None
    d_w, d_b = be.dev(w), (be.dev(b) if bias else None)
    d_y = be.full((N, K, OH, OW), np.nan)
    with s16g_route(be, mbl=mbl):
        ws, wsb = _ws(be, be.lib.conv2d_fwd_workspace(N, H, W, K, C, R, R, stride, pad))
        be.lib.conv2d_fwd(be.ptr(d_x0), C0, be.ptr(d_x1), C1, be.ptr(d_w), be.ptr(d_b), be.ptr(d_y), N, H, W, K, R, R,
                          stride, pad, pad_mode, act, 0.2, be.ptr(ws), wsb, 0, be.stream)
    assert be.lib.last_route() == 3, "the shape did not take the general 16-bit-pipe route"
    got = be.np(d_y)
    err = np.abs(got - want)
    lim = 2e-6 * mag + 1e-6 * (np.abs(want) + (0 if b is None else np.abs(b)[None, :, None, None])) + 1e-30
    if not np.all(err <= lim):
        i = np.unravel_index(np.argmax(err / lim), err.shape)
        raise AssertionError("conv2d_fwd (s16g): err %.3e > %.3e at %s (got %.6g want %.6g)" % (err[i], lim[i], i, got[i], want[i]))


def case_conv_s16g_bwd_data(be, N, C0, C1, H, W, K, R, stride, pad, skip0=False, seed=0, pad_mode=PAD_ZERO, mbl=None, cf=None, addend=False):
    with s16g_route(be, mbl=mbl, cf=cf):
        case_conv_bwd_data(be, N, C0, C1, H, W, K, R, stride, pad, pad_mode, skip0=skip0, seed=seed, addend=addend)
        assert be.lib.last_route() == 3, "the shape did not take the general 16-bit-pipe route"


def case_conv_s16g_bwd_weight(be, N, C0, C1, H, W, K, R, stride, pad, pad_mode, seed=0, gscale=None, route=3, xscale=None):
    """Weight + bias gradient through nemar_conv2d_bwd_weight on the general 16-bit-pipe route (csrc/conv_s16g_wgrad.hip);
    `gscale` [K] multiplies the gradient rows (per-row running scales: rows of very different magnitude keep their accuracy)."""
    rng = np.random.default_rng(seed)
    C = C0 + C1
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    if xscale is not None:                      # per-sample magnitudes of the source
        x = (x * np.asarray(xscale, dtype=np.float32)[:, None, None, None]).astype(np.float32)
    gy = (rng.standard_normal((N, K, OH, OW)) / np.sqrt(N * OH * OW)).astype(np.float32)
    if gscale is not None:
        gy = (gy * np.asarray(gscale, dtype=np.float32)[None, :, None, None]).astype(np.float32)
    w = np.zeros((K, C, R, R))
    _, want_gw, want_gb = O.conv2d_bwd(x.astype(np.float64), w, gy.astype(np.float64), stride, pad, _PM[pad_mode])
    _, mag_gw, mag_gb = O.conv2d_bwd(np.abs(x).astype(np.float64), w, np.abs(gy).astype(np.float64), stride, pad, _PM[pad_mode])
    d_x0 = be.dev(x[:, :C0])
    d_x1 = be.dev(x[:, C0:]) if C1 else None
    d_gy = be.dev(gy)
    outs = []
    with s16g_route(be):
        for rep in range(2):
            d_gw = be.full((K, C, R, R), 0.0)
            d_gb = be.full((K,), 0.0)
            ws, wsb = _ws(be, be.lib.conv2d_bwd_weight_workspace(N, C, H, W, K, OH, OW, R, R, stride, pad))
            be.lib.conv2d_bwd_weight(be.ptr(d_x0), C0, be.ptr(d_x1), C1, be.ptr(d_gy), be.ptr(d_gw), be.ptr(d_gb), N, H, W, K,
                                     OH, OW, R, R, stride, pad, pad_mode, be.ptr(ws), wsb, be.stream)
            assert be.lib.last_route() == route, "the shape took route %d, not %d" % (be.lib.last_route(), route)
            outs.append((be.np(d_gw), be.np(d_gb)))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), "not bitwise reproducible"
    for got, want, mag, what in ((outs[0][0], want_gw, mag_gw, "gw"), (outs[0][1], want_gb, mag_gb, "gb")):
        err = np.abs(got - want)
        lim = 2e-6 * mag + 1e-30
        if not np.all(err <= lim):
            i = np.unravel_index(np.argmax(err / lim), err.shape)
            raise AssertionError("conv2d_bwd_weight (s16g) %s: err %.3e > %.3e at %s (got %.6g want %.6g)" %
                                 (what, err[i], lim[i], i, got[i], want[i]))


ROUTE_K7 = 4


def case_conv_k7_bwd_weight(be, N, C, H, W, K, pad_mode, seed=0, gscale=None, xscale=None):
    """Weight + bias gradient of the 7x7 / pad-3 stem (C <= 4 -> K = 32 n) and head (C = 32 n -> K <= 4) layers on the 16-bit matrix pipe
    (csrc/conv_k7.hip): route 4, bitwise reproducible, 2e-6 of sum |gy| |x| against the float64 oracle."""
    case_conv_s16g_bwd_weight(be, N, C, 0, H, W, K, 7, 1, 3, pad_mode, seed=seed, gscale=gscale, route=ROUTE_K7, xscale=xscale)


def case_conv_k7_fwd(be, N, C, H, W, K, pad_mode, act=O.ACT_NONE, bias=True, seed=0, xscale=None):
    """Forward of the 7x7 / pad-3 stem (C <= 4 -> K = 32 n) on the 16-bit matrix pipe (csrc/conv_k7.hip, few -> many): route 4,
    2e-6 of sum |w| |x| against the float64 oracle.  xscale: per-(sample, channel, row-block) magnitudes — the scale is per tile."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    if xscale is not None:
        x = (x * np.asarray(xscale, dtype=np.float32).reshape(N, 1, 1, 1)).astype(np.float32)
    w = (rng.standard_normal((K, C, 7, 7)) / np.sqrt(C * 49)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32) if bias else None
    want = O.act_fwd(O.conv2d_fwd(x.astype(np.float64), w.astype(np.float64), None if b is None else b.astype(np.float64), 1, 3,
                                  _PM[pad_mode]), act)
    mag = O.conv2d_fwd(np.abs(x).astype(np.float64), np.abs(w).astype(np.float64), None, 1, 3, _PM[pad_mode])
    d_x, d_w, d_b = be.dev(x), be.dev(w), (be.dev(b) if bias else None)
    d_y = be.full((N, K, H, W), np.nan)
    ws, wsb = _ws(be, be.lib.conv2d_fwd_workspace(N, H, W, K, C, 7, 7, 1, 3))
    for prepacked in (0, 1):              # the second call reuses the packed weights
        d_y = be.full((N, K, H, W), np.nan)
        be.lib.conv2d_fwd(be.ptr(d_x), C, None, 0, be.ptr(d_w), be.ptr(d_b), be.ptr(d_y), N, H, W, K, 7, 7, 1, 3, pad_mode, act, 0.2,
                          be.ptr(ws), wsb, prepacked, be.stream)
        assert be.lib.last_route() == ROUTE_K7, "route %d" % be.lib.last_route()
        got = be.np(d_y)
        err = np.abs(got - want)
        lim = 2e-6 * mag + 1e-6 * (np.abs(want) + (0 if b is None else np.abs(b)[None, :, None, None])) + 1e-30
        if not np.all(err <= lim):
            i = np.unravel_index(np.argmax(err / lim), err.shape)
            raise AssertionError("conv2d_fwd (k7, prepacked %d): err %.3e > %.3e at %s (got %.6g want %.6g)" %
                                 (prepacked, err[i], lim[i], i, got[i], want[i]))


def case_conv_k7_bwd_data(be, N, C, H, W, K, pad_mode, seed=0):
    """Data gradient of the 7x7 / pad-3 head (C = 32 n -> K <= 4): few -> many with flipped, transposed weights; reflect border on the
    padded domain + fold."""
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((K, C, 7, 7)) / np.sqrt(K * 49)).astype(np.float32)
    gy = rng.standard_normal((N, K, H, W)).astype(np.float32)
    want, _, _ = O.conv2d_bwd(np.zeros((N, C, H, W)), w.astype(np.float64), gy.astype(np.float64), 1, 3, _PM[pad_mode])
    mag, _, _ = O.conv2d_bwd(np.zeros((N, C, H, W)), np.abs(w).astype(np.float64), np.abs(gy).astype(np.float64), 1, 3, _PM[pad_mode])
    d_gy, d_w = be.dev(gy), be.dev(w)
    ws, wsb = _ws(be, be.lib.conv2d_bwd_data_workspace(N, C, H, W, K, 7, 7, 1, 3, pad_mode))
    for prepacked in (0, 1):
        d_gx = be.full((N, C, H, W), np.nan)
        be.lib.conv2d_bwd_data(be.ptr(d_gy), be.ptr(d_w), None, 0, 0.0, be.ptr(d_gx), C, None, 0, N, H, W, K, H, W, 7, 7, 1, 3, pad_mode,
                               be.ptr(ws), wsb, prepacked, be.stream)
        assert be.lib.last_route() == ROUTE_K7, "route %d" % be.lib.last_route()
        got = be.np(d_gx)
        err = np.abs(got - want)
        lim = 2e-6 * mag + 1e-6 * np.abs(want) + 1e-30
        if not np.all(err <= lim):
            i = np.unravel_index(np.argmax(err / lim), err.shape)
            raise AssertionError("conv2d_bwd_data (k7, prepacked %d): err %.3e > %.3e at %s (got %.6g want %.6g)" %
                                 (prepacked, err[i], lim[i], i, got[i], want[i]))


def case_conv_split16(be, N, C, H, W, K, pad_mode, dgrad, seed=0, R=3):
    """One wide 3x3 / stride 1 / pad 1 layer through nemar_conv2d_fwd (dgrad False) or nemar_conv2d_bwd_data with the scratch
    arena registered: the split-16 route must be eligible for the shape, and obey the same tolerance against the float64 oracle
    as the exact-fp32 kernels."""
    need = split16_scratch(be, N, H, W, K, C, 3, 3, 1, 1)
    assert need > 0, "shape is not eligible for the split-16 kernels"
    with scratch_arena(be, need):
        if dgrad:
            case_conv_bwd_data(be, N, C, 0, H, W, K, 3, 1, 1, pad_mode, seed=seed)
        else:
            case_conv_fwd(be, N, C, 0, H, W, K, 3, 1, 1, pad_mode, act=O.ACT_NONE, seed=seed)


def case_conv_split16_dynamic_range(be, what, N=3, C=128, H=8, W=32, K=128, seed=0):
    """The fp16 x 3 route of the wide layers (csrc/conv_split16*.hip) on adversarial magnitudes, through the C ABI: the scale is per
    SAMPLE, so every sample keeps fp32-class accuracy relative to its own magnitude.  what: 'samples' = per-sample magnitudes
    1 : 1e-6 : 1e4; 'outlier' = one 1e4 outlier in an otherwise O(1) sample; 'zero' = an all-zero sample next to a normal one;
    'inf' = one infinity: only the outputs whose receptive field holds it are non-finite; 'channels' = heavy-tailed CHANNELS (x and gy
    channel magnitudes spread over 1e-3 .. 1e3 inside every sample: the per-sample scale is set by the loudest channel, the quiet ones
    live on the absolute term of the stated bound); 'subnormal' = a sample whose every element but one sits at 2^-27 of its maximum,
    i.e. is an fp16 SUBNORMAL after scaling: the 16-bit MFMA must not flush its inputs (outputs away from the maximum stay non-zero
    and accurate to the subnormal spacing)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    gy = rng.standard_normal((N, K, H, W)).astype(np.float32)
    w = (rng.standard_normal((K, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    if what == 'samples':
        sc = np.array([1.0, 1e-6, 1e4], dtype=np.float32)[:N].reshape(N, 1, 1, 1)
        x, gy = x * sc, gy * sc[::-1]
    elif what == 'outlier':
        x[0, 3, 2, 5] = 1e4
        gy[1, 7, 3, 9] = -1e4
    elif what == 'zero':
        x[1] = 0.0
        gy[0] = 0.0
    elif what == 'inf':
        x[0, 5, 4, 10] = np.inf
    elif what == 'channels':
        x = x * (10.0 ** rng.permutation(np.linspace(-3, 3, C))).reshape(1, C, 1, 1)
        gy = gy * (10.0 ** rng.permutation(np.linspace(-3, 3, K))).reshape(1, K, 1, 1)
    elif what == 'subnormal':
        x[0] = np.float32(2.0 ** -27) * np.sign(x[0])
        x[0, 0, 0, 0] = 1.0
    x, gy = x.astype(np.float32), gy.astype(np.float32)
    need = split16_scratch(be, N, H, W, K, C, 3, 3, 1, 1)
    assert need > 0
    d_x, d_gy, d_w = be.dev(x), be.dev(gy), be.dev(w)
    with scratch_arena(be, need):
        d_y = be.full((N, K, H, W), np.nan)
        ws, wsb = _ws(be, be.lib.conv2d_fwd_workspace(N, H, W, K, C, 3, 3, 1, 1))
        be.lib.conv2d_fwd(be.ptr(d_x), C, None, 0, be.ptr(d_w), None, be.ptr(d_y), N, H, W, K, 3, 3, 1, 1, PAD_REFLECT, 0, 0.2,
                          be.ptr(ws), wsb, 0, be.stream)
        assert be.lib.last_route() == 2
        d_gx = be.full((N, C, H, W), np.nan)
        ws, wsb = _ws(be, be.lib.conv2d_bwd_data_workspace(N, C, H, W, K, 3, 3, 1, 1, PAD_REFLECT))
        be.lib.conv2d_bwd_data(be.ptr(d_gy), be.ptr(d_w), None, 0, 0.0, be.ptr(d_gx), C, None, 0, N, H, W, K, H, W, 3, 3, 1, 1,
                               PAD_REFLECT, be.ptr(ws), wsb, 0, be.stream)
        assert be.lib.last_route() == 2
        d_gw = be.full((K, C, 3, 3), 0.0)
        ws, wsb = _ws(be, be.lib.conv2d_bwd_weight_workspace(N, C, H, W, K, H, W, 3, 3, 1, 1))
        be.lib.tune(26, 0)
        be.lib.conv2d_bwd_weight(be.ptr(d_x), C, None, 0, be.ptr(d_gy), be.ptr(d_gw), None, N, H, W, K, H, W, 3, 3, 1, 1,
                                 PAD_REFLECT, be.ptr(ws), wsb, be.stream)
        assert be.lib.last_route() == 2
    y, gx, gw = be.np(d_y), be.np(d_gx), be.np(d_gw)
    x64, gy64, w64 = x.astype(np.float64), gy.astype(np.float64), w.astype(np.float64)
    if what == 'inf':
        fin = x64.copy()
        fin[0, 5, 4, 10] = 0.0
        want = O.conv2d_fwd(fin, w64, None, 1, 1, 'reflect')
        hit = O.conv2d_fwd((~np.isfinite(x64)).astype(np.float64), np.ones_like(w64), None, 1, 1, 'reflect') > 0
        assert not np.isfinite(y[hit]).any(), "outputs whose receptive field holds the infinity must be non-finite"
        err = np.abs(y[~hit] - want[~hit])
        mag = O.conv2d_fwd(np.abs(fin), np.abs(w64), None, 1, 1, 'reflect')[~hit]
        per = np.abs(fin[0]).max()       # the infinity's own sample is scaled by its finite maximum
        assert np.all(err <= 4e-6 * mag + 2e-11 * per), (err.max(),)
        return
    want = O.conv2d_fwd(x64, w64, None, 1, 1, 'reflect')
    if what == 'subnormal':
        # rows >= 3 of sample 0 see only the 2^-27 elements (2^-16 after scaling: subnormal in fp16, spacing 2^-24): not flushed to
        # zero, and accurate to that spacing per term
        far = want[0, :, 3:, :]
        got = y[0, :, 3:, :]
        assert np.all(far != 0) and np.all(got != 0), "fp16 subnormal operands were flushed"
        spacing = 2.0 ** -24 / 2.0 ** 11                       # one subnormal step in units of x (scale 2^11 for max = 1)
        assert np.all(np.abs(got - far) <= 1.0 * spacing * np.abs(w64).sum(axis=(1, 2, 3)).reshape(K, 1, 1) + 1e-30), \
            float(np.abs(got - far).max())
        return
    want_gx, want_gw, _ = O.conv2d_bwd(x64, w64, gy64, 1, 1, 'reflect')
    mag = O.conv2d_fwd(np.abs(x64), np.abs(w64), None, 1, 1, 'reflect')
    mag_gx, _, _ = O.conv2d_bwd(np.abs(x64), np.abs(w64), np.abs(gy64), 1, 1, 'reflect')
    # bound (include/nemar_hip.h): 3 * 2^-22 relative per product while the fp16 terms are normal, i.e. relative to sum |w||x| of the
    # output's own receptive field; + 2^-36 of the SAMPLE's maximum times sum |w| for elements far below it
    xs = np.abs(x64).reshape(N, -1).max(axis=1).reshape(N, 1, 1, 1)
    gs = np.abs(gy64).reshape(N, -1).max(axis=1).reshape(N, 1, 1, 1)
    wsum = np.abs(w64).sum(axis=(1, 2, 3)).reshape(1, K, 1, 1)
    wsum_t = np.abs(w64).sum(axis=(0, 2, 3)).reshape(1, C, 1, 1)
    # (4e-6 = 2^-18: the fp32 accumulation of a 1152-term dot product, as in the exact-fp32 kernels, on top of the operand split)
    lim = 4e-6 * mag + 3e-11 * xs * wsum + 1e-30
    assert np.all(np.abs(y - want) <= lim), ("fwd", float((np.abs(y - want) / lim).max()))
    lim = 4e-6 * mag_gx + 3e-11 * gs * wsum_t + 1e-30
    assert np.all(np.abs(gx - want_gx) <= lim), ("dgrad", float((np.abs(gx - want_gx) / lim).max()))
    # weight gradient: a sum over samples, each accurate relative to its own magnitudes
    _, mag_gw, _ = O.conv2d_bwd(np.abs(x64), w64 * 0, np.abs(gy64), 1, 1, 'reflect')
    lim = 4e-6 * mag_gw + 3e-11 * float((xs * gs).sum()) * H * W + 1e-30
    assert np.all(np.abs(gw - want_gw) <= lim), ("wgrad", float((np.abs(gw - want_gw) / lim).max()))


def case_absmax_and_hint(be, seed=0):
    """nemar_absmax against numpy, odd sizes and an unaligned view; and a
    split-16 forward with the hint registered gives bit-identical results to one that runs its own max pass."""
    rng = np.random.default_rng(seed)
    for n, off in ((1, 0), (1027, 0), (40003, 1), (300000, 0)):
        a = (rng.standard_normal(n + off) * 10.0 ** rng.uniform(-6, 3)).astype(np.float32)
        d = be.dev(a)
        word = be.bytes_buf(4)
        view = d[off:]
        be.lib.absmax(be.ptr(view), n, be.ptr(word), be.stream)
        got = np.asarray(be.np(word), dtype=np.float32)[:1].view(np.uint32)[0]
        want = np.abs(a[off:]).max().astype(np.float32).view(np.uint32)
        assert int(got) == int(want), (n, off, hex(int(got)), hex(int(want)))
    N, C, H, W, K = 1, 16, 8, 32, 128
    x = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((K, C, 3, 3)) / 12).astype(np.float32)
    d_x, d_w = be.dev(x), be.dev(w)
    outs = []
    with scratch_arena(be, split16_scratch(be, N, H, W, K, C, 3, 3, 1, 1)):
        for hint in (False, True):
            d_y = be.full((N, K, H, W), np.nan)
            wsb = be.lib.conv2d_fwd_workspace(N, H, W, K, C, 3, 3, 1, 1)
            cws = be.bytes_buf(wsb)
            if hint:
                word = be.bytes_buf(4)
                be.lib.absmax(be.ptr(d_x), x.size, be.ptr(word), be.stream)
                be.lib.absmax_hint(be.ptr(d_x), be.ptr(word), 1)
            be.lib.conv2d_fwd(be.ptr(d_x), C, None, 0, be.ptr(d_w), None, be.ptr(d_y), N, H, W, K, 3, 3, 1, 1, PAD_ZERO, 0, 0.2,
                              be.ptr(cws), wsb, 0, be.stream)
            if hint:
                be.lib.absmax_hint(be.ptr(d_x), None, 0)
            outs.append(be.np(d_y))
    assert np.array_equal(outs[0], outs[1])
    # per-sample words: finite maxima only (an infinity / NaN does not take part)
    a = (rng.standard_normal((3, 1000)) * np.array([[1e-5], [1.0], [300.0]])).astype(np.float32)
    a[1, 7] = np.inf
    a[2, 9] = np.nan
    words = be.bytes_buf(12)
    be.lib.absmax_samples(be.ptr(be.dev(a)), 3, 1000, be.ptr(words), be.stream)
    got = np.asarray(be.np(words), dtype=np.float32)[:3].view(np.uint32)
    fin = np.where(np.isfinite(a), np.abs(a), 0).max(axis=1).astype(np.float32).view(np.uint32)
    assert list(map(int, got)) == list(map(int, fin))


def case_conv_split16_wgrad(be, N, C, H, W, K, pad_mode, seed=0, R=3):
    """Weight + bias gradient of a wide 3x3 / stride 1 / pad 1 layer through nemar_conv2d_bwd_weight with the scratch arena registered:
    the fp16 x 3 route (csrc/conv_split16_wgrad.hip) must be eligible and obey the tolerances of the exact-fp32 kernels."""
    need = split16_scratch(be, N, H, W, K, C, R, R, 1, 1)
    assert need > 0, "shape is not eligible for the split-16 kernels"
    with scratch_arena(be, need):
        case_conv_bwd_weight(be, N, C, 0, H, W, K, R, 1, 1, pad_mode, seed=seed)


def case_conv_split16_dual_gy(be, N, C, H, W, K, pad_mode, seed=0):
    """The data-gradient call of a wide 3x3 layer leaves the weight gradient's operand planes of gy behind (nemar_conv_extras.gy_planes_out ->
    .src2_planes, csrc/conv_split16_wgrad.hip split_dual_kernel: gy is read and split once for both calls).  Both gradients must be
    BIT-IDENTICAL to those of the calls that split gy on their own, and the handshake (nemar_last_gy_planes) must say what happened."""
    from nemar_amd._lib import ConvExtras
    import ctypes
    need = split16_scratch(be, N, H, W, K, C, 3, 3, 1, 1)
    assert need > 0, "shape is not eligible for the split-16 kernels"
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    gy = (rng.standard_normal((N, K, H, W)) / np.sqrt(N * H * W)).astype(np.float32)
    w = (rng.standard_normal((K, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    d_x, d_gy, d_w = be.dev(x), be.dev(gy), be.dev(w)
    arena = be.bytes_buf(need)
    lib = be.lib
    lib.tune(23, 0)
    try:
        gbytes = lib.conv2d_gy_planes_bytes(N, C, H, W, K, 3, 3, 1, 1, pad_mode)
        assert gbytes > 0, "layer does not take producer-written gy planes"
        gbuf = be.bytes_buf(gbytes)
        words = be.bytes_buf(4 * N)
        lib.absmax_samples(be.ptr(d_gy), N, K * H * W, be.ptr(words), be.stream)
        wsd, wsdb = _ws(be, lib.conv2d_bwd_data_workspace(N, C, H, W, K, 3, 3, 1, 1, pad_mode))
        wsw, wswb = _ws(be, lib.conv2d_bwd_weight_workspace(N, C, H, W, K, H, W, 3, 3, 1, 1))

        def extras(gy_out=None, src2_planes=None, with_words=True, weight_call=False):
            e = ConvExtras()
            e.scratch, e.scratch_bytes = be.ptr(arena).value, need
            if with_words and not weight_call:
                e.src_max_words, e.src_max_count = be.ptr(words).value, N
            if with_words and weight_call:
                e.src2_max_words, e.src2_max_count = be.ptr(words).value, N
            if gy_out is not None:
                e.gy_planes_out, e.gy_planes_bytes = be.ptr(gy_out).value, gbytes
            if src2_planes is not None:
                e.src2_planes = be.ptr(src2_planes).value
            return ctypes.byref(e)

        def run(dual, with_words):
            d_gx = be.full((N, C, H, W), np.nan)
            d_gw = be.full((K, C, 3, 3), 0.5)
            lib.conv2d_bwd_data_ex(be.ptr(d_gy), be.ptr(d_w), None, 0, 0.0, be.ptr(d_gx), C, None, 0, N, H, W, K, H, W, 3, 3, 1, 1,
                                   pad_mode, be.ptr(wsd), wsdb, 0, be.stream, extras(gy_out=gbuf if dual else None, with_words=with_words))
            assert lib.last_route() == 2
            assert lib.last_gy_planes() == (1 if dual else 0)
            lib.conv2d_bwd_weight_ex(be.ptr(d_x), C, None, 0, be.ptr(d_gy), be.ptr(d_gw), None, N, H, W, K, H, W, 3, 3, 1, 1, pad_mode,
                                     be.ptr(wsw), wswb, be.stream,
                                     extras(src2_planes=gbuf if dual else None, with_words=with_words, weight_call=True))
            assert lib.last_route() == 2
            be.sync()
            return be.raw(d_gx), be.raw(d_gw)

        for with_words in (True, False):
            gx_a, gw_a = run(False, with_words)
            gx_b, gw_b = run(True, with_words)
            assert np.array_equal(gx_a, gx_b), "data gradient differs when its split pass also writes the weight gradient's planes"
            assert np.array_equal(gw_a, gw_b), "weight gradient from the data-gradient call's gy planes differs"
        # and against float64
        want_gx, want_gw, _ = O.conv2d_bwd(x.astype(np.float64), w.astype(np.float64), gy.astype(np.float64), 1, 1, _PM[pad_mode])
        _assert_close(gw_b.view(np.float32).reshape(K, C, 3, 3).astype(np.float64), want_gw + 0.5, atol=2e-5, rtol=2e-5,
                      what="conv2d_bwd_weight (gy planes from the data-gradient call)")
        # a buffer that is too small is not written
        d_gx = be.full((N, C, H, W), np.nan)
        e = ConvExtras()
        e.scratch, e.scratch_bytes = be.ptr(arena).value, need
        e.gy_planes_out, e.gy_planes_bytes = be.ptr(gbuf).value, gbytes - 16
        lib.conv2d_bwd_data_ex(be.ptr(d_gy), be.ptr(d_w), None, 0, 0.0, be.ptr(d_gx), C, None, 0, N, H, W, K, H, W, 3, 3, 1, 1, pad_mode,
                               be.ptr(wsd), wsdb, 0, be.stream, ctypes.byref(e))
        assert lib.last_gy_planes() == 0
        be.sync()
    finally:
        lib.tune(23, 2000)


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
                           W, R, R, 2, 1, PAD_ZERO, be.ptr(ws), wsb, 0, be.stream)
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
    d_gbf = be.full((K,), 0.125)
    ws, wsb = _ws(be, be.lib.conv2d_bwd_weight_workspace(N, C, H, W, K, OH, OW, R, R, stride, pad))
    be.lib.conv2d_bwd_weight(be.ptr(d_x0), C0, be.ptr(d_x1), C1, be.ptr(d_gy), be.ptr(d_gw), be.ptr(d_gbf), N, H, W, K,
                             OH, OW, R, R, stride, pad, pad_mode, be.ptr(ws), wsb, be.stream)
    _assert_close(be.np(d_gw), want_gw + 0.5, atol=2e-5, rtol=2e-5, what="conv2d_bwd_weight")
    _assert_close(be.np(d_gbf), want_gb + 0.125, atol=2e-5, rtol=2e-5, what="conv2d_bwd_weight fused bias grad")
    d_gb = be.full((K,), -0.25)
    ws, wsb = _ws(be, be.lib.bias_grad_workspace(N, K, OH * OW))
    be.lib.bias_grad(be.ptr(d_gy), be.ptr(d_gb), N, K, OH * OW, be.ptr(ws), wsb, be.stream)
    _assert_close(be.np(d_gb), want_gb - 0.25, atol=2e-5, rtol=2e-5, what="bias_grad")


# wave-specialised wide weight gradient (conv_wgrad.hip): K > 32 and OH*OW % 4 == 0
WGRAD_WIDE_CASES = [
    # N, C0, C1, H,  W,  K,   R, stride, pad, pad_mode
    (2, 16, 0, 8, 10, 40, 3, 1, 1, PAD_REFLECT),    # 2 column tiles (J = 144), ragged K, reflect border
    (3, 8, 0, 6, 6, 150, 3, 1, 1, PAD_ZERO),        # 2 channel tiles, P = 108: 16-pixel stage tail, image straddle
    (2, 32, 0, 8, 12, 70, 3, 2, 1, PAD_ZERO),       # stride 2 -> 4x6 outputs
    (2, 16, 16, 8, 8, 48, 3, 1, 1, PAD_ZERO),       # two sources (decoder concat)
    (2, 3, 3, 10, 10, 64, 4, 2, 1, PAD_ZERO),       # D first layer: k4 s2 -> 5x5?  (25 % 4 != 0: falls back)
    (2, 6, 0, 17, 17, 64, 4, 2, 1, PAD_ZERO),       # k4 s2 -> 8x8
    (1, 16, 0, 8, 8, 130, 1, 1, 0, PAD_ZERO),       # 1x1, 3 row tiles... K = 130 -> 2 tiles
    (9, 4, 0, 12, 12, 33, 3, 1, 1, PAD_REFLECT),    # many splits (P = 1296)
    # 16-byte source loads (stride 1, OW % 16 == 0): border chunks patched by the MFMA waves
    (2, 16, 0, 4, 16, 40, 3, 1, 1, PAD_REFLECT),     # every stage is both first and last chunk of a row
    (1, 16, 0, 6, 32, 64, 3, 1, 1, PAD_ZERO),        # 2 stages per row, zero border
    (2, 8, 8, 3, 16, 48, 3, 1, 1, PAD_ZERO),         # two sources, H = 3
    (1, 16, 0, 4, 16, 40, 1, 1, 0, PAD_ZERO),        # 1x1: no border chunks at all
    (3, 20, 0, 5, 48, 70, 3, 1, 1, PAD_REFLECT),     # 3 stages per row, 2 column tiles, ragged K
    # narrower channel tiles: K <= 32 -> 32-row tile (1x4 waves), K <= 64 -> 64-row tile
    (2, 16, 0, 8, 8, 20, 3, 1, 1, PAD_REFLECT),
    (2, 8, 0, 4, 16, 32, 3, 1, 1, PAD_ZERO),         # 32-row tile with 16-byte source loads
    (1, 3, 3, 8, 8, 24, 3, 1, 1, PAD_ZERO),          # STN first conv: 3+3 channels -> J = 54
    (2, 32, 0, 4, 16, 32, 1, 1, 0, PAD_ZERO),        # 1x1
    (2, 24, 0, 6, 32, 64, 3, 1, 1, PAD_REFLECT),     # 64-row tile, vector loads, 2 column tiles
    # odd output planes (OH*OW % 4 != 0: the discriminator's 31x31 / 15x15 maps): gy copied into zero-padded planes
    (2, 16, 0, 8, 8, 40, 4, 1, 1, PAD_ZERO),         # 7x7 outputs -> 8 virtual rows
    (3, 8, 0, 6, 7, 150, 4, 1, 1, PAD_ZERO),         # 5x6 = 30 -> 6 virtual rows (36), 2 channel tiles
    (2, 12, 0, 11, 11, 20, 3, 2, 1, PAD_ZERO),       # stride 2 -> 6x6 (fine) ; K <= 32 tile
    (1, 8, 0, 9, 9, 70, 3, 1, 0, PAD_ZERO),          # 7x7 outputs, no padding
]



# ------------------------------------------------------------------------------------------------
def case_instnorm(be, N, C, H, W, act, residual=False, seed=0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((N, C, H, W)) * 2 + rng.standard_normal((N, C, 1, 1)) * 3).astype(np.float32)
    res = rng.standard_normal((N, C, H, W)).astype(np.float32) if residual else None
    gy = rng.standard_normal((N, C, H, W)).astype(np.float32)
    x64 = x.astype(np.float64)
    xhat, m, rstd = O.instance_norm_fwd(x64)
    want = O.act_fwd(xhat, act) + (res.astype(np.float64) if residual else 0.0)
    g = gy.astype(np.float64) * (O.act_bwd(np.ones_like(xhat), O.act_fwd(xhat, act), act))
    want_gx = O.instance_norm_bwd(x64, g)
    d_x, d_gy = be.dev(x), be.dev(gy)
    d_res = be.dev(res) if residual else None
    d_y = be.full((N, C, H, W), np.nan)
    d_st = be.full((N * C, 2), np.nan)
    be.lib.instnorm_fwd(be.ptr(d_x), be.ptr(d_res), be.ptr(d_y), be.ptr(d_st), N * C, H * W, 1e-5, act, 0.2, be.stream)
    _assert_close(be.np(d_y), want, atol=3e-5, rtol=1e-5, what="instnorm_fwd")
    st = be.np(d_st)
    _assert_close(st[:, 0], m.reshape(-1), atol=1e-5, rtol=1e-5, what="instnorm mean")
    _assert_close(st[:, 1], rstd.reshape(-1), atol=0, rtol=3e-5, what="instnorm rstd")
    d_gx = be.full((N, C, H, W), np.nan)
    be.lib.instnorm_bwd(be.ptr(d_x), be.ptr(d_st), be.ptr(d_gy), be.ptr(d_gx), N * C, H * W, act, 0.2, be.stream)
    _assert_close(be.np(d_gx), want_gx, atol=3e-5 * np.abs(want_gx).max(), rtol=1e-4, what="instnorm_bwd")


def case_producer_max_words(be, seed=0):
    """InstanceNorm forward / backward and dropout with the per-sample maximum of their output as a by-product (the words the fp16 x 3
    convolutions scale by): same outputs as the plain entry points, words == numpy's per-sample finite maximum."""
    rng = np.random.default_rng(seed)
    N, C, H, W = 3, 8, 6, 10
    x = (rng.standard_normal((N, C, H, W)) * np.array([1.0, 1e-3, 50.0]).reshape(N, 1, 1, 1)).astype(np.float32)
    res = rng.standard_normal((N, C, H, W)).astype(np.float32)
    gy = rng.standard_normal((N, C, H, W)).astype(np.float32)
    d_x, d_res, d_gy = be.dev(x), be.dev(res), be.dev(gy)

    def words_of(buf):
        return list(map(int, np.asarray(be.np(buf), dtype=np.float32)[:N].view(np.uint32)))

    def want(a):
        return list(map(int, np.abs(a.reshape(N, -1)).max(axis=1).astype(np.float32).view(np.uint32)))

    for act, r in ((1, None), (0, d_res)):
        y0, y1 = be.full(x.shape, np.nan), be.full(x.shape, np.nan)
        st0, st1 = be.zeros(N * C, 2), be.zeros(N * C, 2)
        w = be.bytes_buf(4 * N * 2049)          # NEMAR_MAX_WORDS(N): results + per-workgroup partials, uninitialised
        be.lib.instnorm_fwd(be.ptr(d_x), be.ptr(r), be.ptr(y0), be.ptr(st0), N * C, H * W, 1e-5, act, 0.2, be.stream)
        be.lib.instnorm_fwd_max(be.ptr(d_x), be.ptr(r), be.ptr(y1), be.ptr(st1), N * C, H * W, 1e-5, act, 0.2, be.ptr(w), C, be.stream)
        a = np.asarray(be.np(y1), dtype=np.float32)
        assert np.array_equal(be.np(y0), be.np(y1)) and words_of(w) == want(a)
        g0, g1 = be.full(x.shape, np.nan), be.full(x.shape, np.nan)
        w = be.bytes_buf(4 * N * 2049)          # NEMAR_MAX_WORDS(N): results + per-workgroup partials, uninitialised
        be.lib.instnorm_bwd(be.ptr(d_x), be.ptr(st0), be.ptr(d_gy), be.ptr(g0), N * C, H * W, act, 0.2, be.stream)
        be.lib.instnorm_bwd_max(be.ptr(d_x), be.ptr(st0), be.ptr(d_gy), be.ptr(g1), N * C, H * W, act, 0.2, be.ptr(w), C, be.stream)
        a = np.asarray(be.np(g1), dtype=np.float32)
        assert np.array_equal(be.np(g0), be.np(g1)) and words_of(w) == want(a)
        # LAZY words (nemar_set_max_words_lazy): the call leaves the marker 0xFFFF0000 | planes-per-sample in each result word, and
        # nemar_max_words_finalize — on demand, idempotent — turns it into the same maxima
        for fwd in (True, False):
            out = be.full(x.shape, np.nan)
            w = be.bytes_buf(4 * N * 2049)
            be.lib.set_max_words_lazy(1)
            try:
                if fwd:
                    be.lib.instnorm_fwd_max(be.ptr(d_x), be.ptr(r), be.ptr(out), be.ptr(st1), N * C, H * W, 1e-5, act, 0.2, be.ptr(w), C, be.stream)
                else:
                    be.lib.instnorm_bwd_max(be.ptr(d_x), be.ptr(st0), be.ptr(d_gy), be.ptr(out), N * C, H * W, act, 0.2, be.ptr(w), C, be.stream)
            finally:
                be.lib.set_max_words_lazy(0)
            assert words_of(w) == [0xFFFF0000 | C] * N, "lazy marker"
            a = np.asarray(be.np(out), dtype=np.float32)
            for _ in range(2):
                be.lib.max_words_finalize(be.ptr(w), N, be.stream)
                assert words_of(w) == want(a), "nemar_max_words_finalize"
    y0, y1 = be.full(x.shape, np.nan), be.full(x.shape, np.nan)
    w = be.bytes_buf(4 * N * 2049)
    be.lib.dropout(be.ptr(d_x), be.ptr(y0), x.size, 0.5, 1234567, 9, be.stream)
    be.lib.dropout_max(be.ptr(d_x), be.ptr(y1), N, x.size // N, 0.5, 1234567, 9, be.ptr(w), be.stream)
    a = np.asarray(be.np(y1), dtype=np.float32)
    assert np.array_equal(be.np(y0), be.np(y1)) and words_of(w) == want(a)


def case_conv_ex(be, N=3, C=128, H=8, W=32, K=128, seed=0):
    """nemar_conv2d_*_ex: arena and max words passed with the call == the registered arena / hints, bit for bit, on the same route."""
    import ctypes as C_
    from nemar_amd._lib import ConvExtras
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    gy = rng.standard_normal((N, K, H, W)).astype(np.float32)
    w = (rng.standard_normal((K, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    need = split16_scratch(be, N, H, W, K, C, 3, 3, 1, 1)
    assert need > 0
    d_x, d_gy, d_w = be.dev(x), be.dev(gy), be.dev(w)

    def run(ex):
        outs = []
        fwd = be.lib.conv2d_fwd_ex if ex is not None else be.lib.conv2d_fwd
        bwd = be.lib.conv2d_bwd_data_ex if ex is not None else be.lib.conv2d_bwd_data
        wgr = be.lib.conv2d_bwd_weight_ex if ex is not None else be.lib.conv2d_bwd_weight
        tail = lambda e: (C_.byref(e),) if ex is not None else ()
        d_y = be.full((N, K, H, W), np.nan)
        ws, wsb = _ws(be, be.lib.conv2d_fwd_workspace(N, H, W, K, C, 3, 3, 1, 1))
        fwd(be.ptr(d_x), C, None, 0, be.ptr(d_w), None, be.ptr(d_y), N, H, W, K, 3, 3, 1, 1, PAD_REFLECT, 0, 0.2, be.ptr(ws), wsb, 0,
            be.stream, *tail(ex and ex[0]))
        assert be.lib.last_route() == 2
        d_gx = be.full((N, C, H, W), np.nan)
        ws, wsb = _ws(be, be.lib.conv2d_bwd_data_workspace(N, C, H, W, K, 3, 3, 1, 1, PAD_REFLECT))
        bwd(be.ptr(d_gy), be.ptr(d_w), None, 0, 0.0, be.ptr(d_gx), C, None, 0, N, H, W, K, H, W, 3, 3, 1, 1, PAD_REFLECT, be.ptr(ws), wsb, 0,
            be.stream, *tail(ex and ex[1]))
        assert be.lib.last_route() == 2
        d_gw = be.full((K, C, 3, 3), 0.0)
        ws, wsb = _ws(be, be.lib.conv2d_bwd_weight_workspace(N, C, H, W, K, H, W, 3, 3, 1, 1))
        wgr(be.ptr(d_x), C, None, 0, be.ptr(d_gy), be.ptr(d_gw), None, N, H, W, K, H, W, 3, 3, 1, 1, PAD_REFLECT, be.ptr(ws), wsb,
            be.stream, *tail(ex and ex[2]))
        assert be.lib.last_route() == 2
        be.sync()
        return [be.np(t) for t in (d_y, d_gx, d_gw)]

    with scratch_arena(be, need):
        want = run(None)
    # per call: an arena of its own, and the max words of x / gy computed once and handed to every call that takes the tensor
    arena = be.bytes_buf(need)
    xw, gw_ = be.bytes_buf(4 * N), be.bytes_buf(4 * N)
    be.lib.absmax_samples(be.ptr(d_x), N, C * H * W, be.ptr(xw), be.stream)
    be.lib.absmax_samples(be.ptr(d_gy), N, K * H * W, be.ptr(gw_), be.stream)
    vp = lambda h: C_.cast(be.ptr(h), C_.c_void_p)
    mk = lambda a, na, b, nb: ConvExtras(vp(arena), need, vp(a) if a is not None else None, na, vp(b) if b is not None else None, nb, None)
    be.lib.tune(23, 0)
    try:
        got = run((mk(xw, N, None, 0), mk(gw_, N, None, 0), mk(xw, N, gw_, N)))
    finally:
        be.lib.tune(23, 2000)
    for a, b, what in zip(want, got, ('fwd', 'dgrad', 'wgrad')):
        assert np.array_equal(a, b), what


def case_step_params_in_device_memory(be, seed=0):
    """What a captured hipGraph needs: nemar_adam_step_dev (the two step-dependent scalars read from device memory) == nemar_adam_step,
    and a dropout launch with base word b and offset k == the launch with offset b + k and no base — bit for bit."""
    rng = np.random.default_rng(seed)
    n = 5003
    p0 = rng.standard_normal(n).astype(np.float32)
    g = (rng.standard_normal(n) * 0.1).astype(np.float32)
    m0 = (rng.standard_normal(n) * 0.01).astype(np.float32)
    v0 = (rng.random(n) * 1e-3).astype(np.float32)
    lr, b1, b2, eps, step = 2e-4, 0.5, 0.999, 1e-8, 7
    outs = []
    for dev in (False, True):
        d_p, d_g, d_m, d_v = be.dev(p0.copy()), be.dev(g.copy()), be.dev(m0.copy()), be.dev(v0.copy())     # (the host backend aliases)
        if dev:
            hyper = be.dev(np.array([lr / (1.0 - b1 ** step), (1.0 - b2 ** step) ** 0.5], dtype=np.float64).astype(np.float32))
            be.lib.adam_step_dev(be.ptr(d_p), be.ptr(d_g), be.ptr(d_m), be.ptr(d_v), n, be.ptr(hyper), b1, b2, eps, be.stream)
        else:
            be.lib.adam_step(be.ptr(d_p), be.ptr(d_g), be.ptr(d_m), be.ptr(d_v), n, lr, b1, b2, eps, step, be.stream)
        be.sync()
        outs.append((be.np(d_p), be.np(d_m), be.np(d_v)))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    x = rng.standard_normal((2, 16, 6, 10)).astype(np.float32)
    d_x = be.dev(x)
    y0, y1 = be.full(x.shape, np.nan), be.full(x.shape, np.nan)
    be.lib.dropout(be.ptr(d_x), be.ptr(y0), x.size, 0.5, 99, 1000 + 7, be.stream)
    base = be.dev_i32(np.array([1000], dtype=np.int32))
    be.lib.set_dropout_base(be.ptr(base))
    try:
        be.lib.dropout(be.ptr(d_x), be.ptr(y1), x.size, 0.5, 99, 7, be.stream)
        be.sync()
    finally:
        be.lib.set_dropout_base(None)
    assert np.array_equal(be.np(y0), be.np(y1)) and (be.np(y0) == 0).any() and (be.np(y0) != 0).any()


def _decode_planes(raw, N, C, H, W):
    """conv_split16.hip's plane layout -> float64 [2 (hi, lo)][N][C][H+4][W+4]"""
    a = raw.view(np.float16).astype(np.float64).reshape(2, N, C // 8, H + 4, W + 4, 8)
    return a.transpose(0, 1, 2, 5, 3, 4).reshape(2, N, C, H + 4, W + 4)


def case_instnorm_planes(be, act, use_residual, drop_p, N=2, C=16, H=6, W=8, seed=0):
    """InstanceNorm (+ activation, + dropout, + residual) that also writes its output as the fp16 x 3 planes of the following 3x3
    reflect convolution (csrc/norm_planes.hip): fp32 output / statistics against float64, the dropout mask against nemar_dropout's,
    the planes decoded word by word (interior = the fp32 output to 22 bits under the a-priori bound's scale, mirrored border,
    zero spare rows / slots), the bound and maximum words."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((N, C, H, W)) * 3 + 1).astype(np.float32)
    res = (rng.standard_normal((N, C, H, W)) * np.array([1.0, 40.0]).reshape(2, 1, 1, 1)[:N]).astype(np.float32) if use_residual else None
    HW = H * W
    mean = x.astype(np.float64).mean(axis=(2, 3), keepdims=True)
    var = x.astype(np.float64).var(axis=(2, 3), keepdims=True)
    want = (x - mean) / np.sqrt(var + 1e-5)
    if act == 1:
        want = np.maximum(want, 0)
    elif act == 2:
        want = np.where(want > 0, want, 0.2 * want)
    keep = np.ones_like(want, dtype=bool)
    if drop_p > 0:
        ones, m = be.dev(np.ones_like(x)), be.full(x.shape, np.nan)
        be.lib.dropout(be.ptr(ones), be.ptr(m), x.size, drop_p, 424242, 5, be.stream)
        keep = be.np(m) != 0
        want = np.where(keep, want / (1 - drop_p), 0.0)
    resmax = None
    if use_residual:
        want = want + res
        resmax = be.dev(np.abs(res).reshape(N, -1).max(axis=1).astype(np.float32))
    d_x, d_res = be.dev(x), (be.dev(res) if use_residual else None)
    d_y, d_st = be.full(x.shape, np.nan), be.full((N * C, 2), np.nan)
    planes = be.bytes_buf(2 * N * (C // 8) * (H + 4) * (W + 4) * 16)
    scale_w, max_w = be.bytes_buf(4 * N), be.bytes_buf(4 * N * 2049)
    be.lib.instnorm_fwd_planes(be.ptr(d_x), be.ptr(d_res), be.ptr(resmax), be.ptr(d_y), be.ptr(d_st), N, C, H, W, 1e-5, act, 0.2,
                               drop_p, 424242, 5, be.ptr(planes), be.ptr(scale_w), be.ptr(max_w), None, be.stream)
    be.sync()
    y = be.np(d_y)
    assert np.array_equal(y != 0, keep & (y != 0)) and np.abs(y - want).max() < 2e-5 * max(1.0, np.abs(want).max()), np.abs(y - want).max()
    if drop_p > 0:
        assert np.all(y[~keep] == 0)
    st = be.np(d_st).reshape(N, C, 2)
    assert np.abs(st[..., 0] - mean[..., 0, 0]).max() < 1e-5 and np.abs(st[..., 1] * np.sqrt(var[..., 0, 0] + 1e-5) - 1).max() < 1e-5
    # the words: bound = sqrt(HW) / (1 - p) + max |residual|;  maximum = max |y| per sample, bit for bit
    bound = be.raw(scale_w)[:4 * N].view(np.float32).astype(np.float64)
    want_bound = np.sqrt(HW) / (1 - drop_p) + (np.abs(res).reshape(N, -1).max(axis=1) if use_residual else 0.0)
    assert np.allclose(bound, want_bound, rtol=1e-6), (bound, want_bound)
    assert np.all(bound >= np.abs(y).reshape(N, -1).max(axis=1))
    got_max = be.raw(max_w)[:4 * N].view(np.uint32)
    assert np.array_equal(got_max, np.abs(y).reshape(N, -1).max(axis=1).astype(np.float32).view(np.uint32))
    # the planes
    scale = 2.0 ** (11 - np.floor(np.log2(bound)))
    pl = _decode_planes(be.raw(planes)[:2 * N * (C // 8) * (H + 4) * (W + 4) * 16], N, C, H, W)
    val = (pl[0] + pl[1]) / scale.reshape(N, 1, 1, 1)
    ypad = np.pad(y, ((0, 0), (0, 0), (1, 1), (1, 1)), mode='reflect')
    err = np.abs(val[:, :, :H + 2, :W + 2] - ypad)
    assert np.all(err <= 2.0 ** -21 * np.abs(ypad) + 2.0 ** -24 / scale.reshape(N, 1, 1, 1)), err.max()
    assert np.all(pl[:, :, :, H + 2:, :] == 0) and np.all(pl[:, :, :, :, W + 2:] == 0)
    hi = pl[0] / scale.reshape(N, 1, 1, 1)                       # the high plane alone is the fp16 rounding of the value
    assert np.all(np.abs(hi[:, :, :H + 2, :W + 2] - ypad) <= 2.0 ** -11 * np.abs(ypad) + 2.0 ** -24 / scale.reshape(N, 1, 1, 1))


def case_conv_from_producer_planes(be, N=3, C=128, H=8, W=32, K=128, seed=0):
    """A wide 3x3 reflect convolution that takes its operand planes from the InstanceNorm pass that produced its input
    (nemar_planes_hint + the bound words as nemar_absmax_hint): same accuracy bound as with its own max + split passes; and the
    planes really are what it read (zeroed planes -> zero output)."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((N, C, H, W)) * np.array([1.0, 1e-3, 30.0]).reshape(3, 1, 1, 1)[:N]).astype(np.float32)
    w = (rng.standard_normal((K, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    need = split16_scratch(be, N, H, W, K, C, 3, 3, 1, 1)
    assert need > 0
    d_x, d_w = be.dev(x), be.dev(w)
    d_h, d_st = be.full(x.shape, np.nan), be.full((N * C, 2), np.nan)
    nb = 2 * N * (C // 8) * (H + 4) * (W + 4) * 16
    planes, scale_w = be.bytes_buf(nb), be.bytes_buf(4 * N)
    be.lib.instnorm_fwd_planes(be.ptr(d_x), None, None, be.ptr(d_h), be.ptr(d_st), N, C, H, W, 1e-5, 1, 0.2, 0.0, 0, 0, be.ptr(planes),
                               be.ptr(scale_w), None, None, be.stream)
    outs = []
    with scratch_arena(be, need):
        for mode in ('planes', 'own passes', 'zeroed planes'):
            d_y = be.full((N, K, H, W), np.nan)
            ws, wsb = _ws(be, be.lib.conv2d_fwd_workspace(N, H, W, K, C, 3, 3, 1, 1))
            src = planes if mode == 'planes' else be.bytes_buf(nb)
            if mode != 'own passes':
                be.lib.planes_hint(be.ptr(d_h), be.ptr(src), N, C, H, W)
                be.lib.absmax_hint(be.ptr(d_h), be.ptr(scale_w), N)
            be.lib.conv2d_fwd(be.ptr(d_h), C, None, 0, be.ptr(d_w), None, be.ptr(d_y), N, H, W, K, 3, 3, 1, 1, PAD_REFLECT, 0, 0.2,
                              be.ptr(ws), wsb, 0, be.stream)
            assert be.lib.last_route() == 2
            be.lib.planes_hint(be.ptr(d_h), None, 0, 0, 0, 0)
            be.lib.absmax_hint(be.ptr(d_h), None, 0)
            be.sync()
            outs.append(be.np(d_y))
    h64, w64 = be.np(d_h), w.astype(np.float64)
    want = O.conv2d_fwd(h64, w64, None, 1, 1, 'reflect')
    mag = O.conv2d_fwd(np.abs(h64), np.abs(w64), None, 1, 1, 'reflect')
    bound = be.raw(scale_w)[:4 * N].view(np.float32).astype(np.float64).reshape(N, 1, 1, 1)
    wsum = np.abs(w64).sum(axis=(1, 2, 3)).reshape(1, K, 1, 1)
    for y, who in zip(outs[:2], ('planes', 'own passes')):
        lim = 4e-6 * mag + 3e-11 * bound * wsum + 1e-30
        assert np.all(np.abs(y - want) <= lim), (who, float((np.abs(y - want) / lim).max()))
    assert np.all(outs[2] == 0), "the convolution did not read the hinted planes"


def _decode_pixel_planes(raw, N, C, rows, CPR):
    """conv_split16_wgrad.hip's pixel-major layout (X or G_0 planes) -> float64 [2 (hi, lo)][N][C][rows][8 CPR]"""
    a = raw.view(np.float16).astype(np.float64).reshape(2, N, C // 64, rows, CPR, 64, 8)
    return a.transpose(0, 1, 2, 5, 3, 4, 6).reshape(2, N, C, rows, CPR * 8)


def _dgrad_plane_content(g, reflect):
    """what conv_split16.hip's plane_value makes of g [N, C, H, W]: [N, C, H+4, W+4] (zero padding, or the reflect data gradient's
    folded border rows / slots)"""
    N, C, H, W = g.shape
    out = np.zeros((N, C, H + 4, W + 4), dtype=np.float64)
    rows = {r: [r - 1] for r in range(1, H + 1)}
    cols = {c: [c - 1] for c in range(1, W + 1)}
    if reflect:
        rows[H + 2], rows[H + 3] = [0, 2], [H - 3, H - 1]
        cols[W + 2], cols[W + 3] = [0, 2], [W - 3, W - 1]
    for r, ys in rows.items():
        for c, xs in cols.items():
            out[:, :, r, c] = sum(g[:, :, y, x] for y in ys for x in xs)
    return out


def case_resblock_planes_chain(be, pad_mode, act, drop_p, N=2, C=128, H=8, W=32, seed=0, producers_only=False):
    """The producer-fused chain of a ResnetBlock convolution (round 6): the InstanceNorm pass in front writes the weight gradient's X
    planes too (nemar_instnorm_fwd_planes wgrad_planes), the InstanceNorm BACKWARD behind writes the operand planes of both gradient
    calls instead of the fp32 tensor (nemar_instnorm_bwd_planes), the data gradient adds the skip gradient and publishes the per-sample
    maximum of its result in the epilogue (nemar_conv_extras.addend / .out_max_words) — every piece against float64, the planes decoded
    word by word."""
    from nemar_amd._lib import ConvExtras
    import ctypes
    lib = be.lib
    rng = np.random.default_rng(seed)
    K = C
    reflect = pad_mode == PAD_REFLECT
    mags = np.resize(np.array([1.0, 30.0, 1e-2]), N).reshape(N, 1, 1, 1)
    # ---- forward producer: y = dropout(act(IN(x0))) with both plane sets ----
    x0 = (rng.standard_normal((N, C, H, W)) * 3 + 1).astype(np.float32)
    d_x0 = be.dev(x0)
    d_y, d_st0 = be.full(x0.shape, np.nan), be.full((N * C, 2), np.nan)
    CPR, Hg = (W + 2 + 7) // 8, (H + 3) // 4 * 4
    Hx = Hg + 2
    xbytes = lib.conv2d_x_planes_bytes(N, C, H, W, 3)
    assert xbytes == 2 * N * C * Hx * CPR * 16
    cplanes = be.bytes_buf(2 * N * (C // 8) * (H + 4) * (W + 4) * 16)
    xplanes = be.bytes_buf(xbytes)
    fscale, fmax = be.bytes_buf(4 * N), be.bytes_buf(4 * N * 2049)
    lib.instnorm_fwd_planes(be.ptr(d_x0), None, None, be.ptr(d_y), be.ptr(d_st0), N, C, H, W, 1e-5, act, 0.2, drop_p, 777, 3,
                            be.ptr(cplanes), be.ptr(fscale), be.ptr(fmax), be.ptr(xplanes), be.stream)
    be.sync()
    y = be.np(d_y).astype(np.float64)
    fb = be.raw(fscale)[:4 * N].view(np.float32).astype(np.float64)
    fs = (2.0 ** (11 - np.floor(np.log2(fb)))).reshape(N, 1, 1, 1)
    xp = _decode_pixel_planes(be.raw(xplanes)[:xbytes], N, C, Hx, CPR)
    val = (xp[0] + xp[1]) / fs
    ypad = np.pad(y, ((0, 0), (0, 0), (1, 1), (1, 1)), mode='reflect')
    err = np.abs(val[:, :, :H + 2, :W + 2] - ypad)
    assert np.all(err <= 2.0 ** -21 * np.abs(ypad) + 2.0 ** -24 / fs), ("X planes", err.max())
    assert np.all(xp[:, :, :, H + 2:, :] == 0) and np.all(xp[:, :, :, :, W + 2:] == 0), "X planes: spare rows / columns must be zero"
    # the channel-blocked planes of the same call hold the same values
    cp = _decode_planes(be.raw(cplanes)[:2 * N * (C // 8) * (H + 4) * (W + 4) * 16], N, C, H, W)
    assert np.array_equal(cp[:, :, :, :H + 2, :W + 2], xp[:, :, :, :H + 2, :W + 2]), "the two layouts of one producer differ"
    # ---- lazy maximum words (nemar_set_max_words_lazy): the same call leaves the marker instead of launching the reduction; the next producer
    # (IN + skip: residual_max_words) reduces the partial words itself and must come out bit for bit as with finalized words ----
    fmax_lazy = be.bytes_buf(4 * N * 2049)
    cplanes2, xplanes2, fscale2 = be.bytes_buf(2 * N * (C // 8) * (H + 4) * (W + 4) * 16), be.bytes_buf(xbytes), be.bytes_buf(4 * N)
    d_y2 = be.full(x0.shape, np.nan)
    assert lib.set_max_words_lazy(1) == 0
    try:
        lib.instnorm_fwd_planes(be.ptr(d_x0), None, None, be.ptr(d_y2), be.ptr(d_st0), N, C, H, W, 1e-5, act, 0.2, drop_p, 777, 3,
                                be.ptr(cplanes2), be.ptr(fscale2), be.ptr(fmax_lazy), be.ptr(xplanes2), be.stream)
    finally:
        assert lib.set_max_words_lazy(0) == 1
    be.sync()
    wl, we = be.raw(fmax_lazy)[:4 * N * 2049].view(np.uint32), be.raw(fmax)[:4 * N * 2049].view(np.uint32)
    assert np.all(wl[:N] == (0xFFFF0000 | (C // 8))), [hex(int(v)) for v in wl[:N]]
    assert np.array_equal(wl[N:N + N * (C // 8)], we[N:N + N * (C // 8)]), "lazy / eager partial words differ"
    assert np.array_equal(we[:N], we[N:N + N * (C // 8)].reshape(N, C // 8).max(axis=1)), "finalized words are not the maxima of the partials"
    x9 = (rng.standard_normal((N, C, H, W)) * 2).astype(np.float32)
    d_x9 = be.dev(x9)
    outs = []
    for words in (fmax, fmax_lazy):
        o9, st9, pl9, sc9 = be.full(x0.shape, np.nan), be.full((N * C, 2), np.nan), be.bytes_buf(2 * N * (C // 8) * (H + 4) * (W + 4) * 16), be.bytes_buf(4 * N)
        lib.instnorm_fwd_planes(be.ptr(d_x9), be.ptr(d_y), be.ptr(words), be.ptr(o9), be.ptr(st9), N, C, H, W, 1e-5, 0, 0.2, 0.0, 0, 0,
                                be.ptr(pl9), be.ptr(sc9), None, None, be.stream)
        be.sync()
        outs.append((be.np(o9).copy(), be.raw(sc9)[:4 * N].copy(), be.raw(pl9)[:2 * N * (C // 8) * (H + 4) * (W + 4) * 16].copy()))
    assert all(np.array_equal(a, b) for a, b in zip(*outs)), "a consumer of lazy residual words differs from one of finalized words"

    # ---- backward producer: the InstanceNorm in front of which the convolution sits (x1 = the convolution's output) ----
    x1 = (rng.standard_normal((N, C, H, W)) * np.linspace(0.5, 4.0, C).reshape(1, C, 1, 1) - 0.5).astype(np.float32)
    gy = (rng.standard_normal((N, C, H, W)) * mags).astype(np.float32)
    mean = x1.astype(np.float64).mean(axis=(2, 3), keepdims=True)
    var = x1.astype(np.float64).var(axis=(2, 3), keepdims=True)
    rstd = 1.0 / np.sqrt(var + 1e-5)
    st1 = np.stack([mean[..., 0, 0], rstd[..., 0, 0]], axis=-1).astype(np.float32)        # [N, C, 2] as the forward would have saved it
    xhat = (x1 - st1[..., 0].reshape(N, C, 1, 1).astype(np.float64)) * st1[..., 1].reshape(N, C, 1, 1).astype(np.float64)
    g = gy.astype(np.float64)
    if drop_p > 0:
        ones, m = be.dev(np.ones_like(x1)), be.full(x1.shape, np.nan)
        lib.dropout(be.ptr(ones), be.ptr(m), x1.size, drop_p, 4242, 9, be.stream)
        g = np.where(be.np(m) != 0, g / (1 - drop_p), 0.0)
    if act == 1:
        g = g * (xhat > 0)
    elif act == 2:
        g = g * np.where(xhat > 0, 1.0, 0.2)
    r64 = st1[..., 1].reshape(N, C, 1, 1).astype(np.float64)
    want_gx = r64 * (g - g.mean(axis=(2, 3), keepdims=True) - xhat * (g * xhat).mean(axis=(2, 3), keepdims=True))
    d_x1, d_st1, d_gy = be.dev(x1), be.dev(st1.reshape(N * C, 2)), be.dev(gy)
    gymax = be.dev(np.abs(gy).reshape(N, -1).max(axis=1).astype(np.float32))
    d_gx = be.full(x1.shape, np.nan)
    dbytes = 2 * N * (C // 8) * (H + 4) * (W + 4) * 16
    lib.tune(23, 0)
    lib.tune(39, 1)                  # (a few-tile test shape would split its reduction over workgroups: no fused epilogue there)
    try:
        gbytes = 2 * N * K * Hg * CPR * 16
        if not producers_only:        # (producers_only: a plane size the wide convolution kernels do not take — the producers themselves do)
            assert lib.conv2d_gy_planes_bytes(N, C, H, W, K, 3, 3, 1, 1, pad_mode) == gbytes
            assert lib.conv2d_bwd_data_fusable(N, C, H, W, K, 3, 3, 1, 1, pad_mode) == 1
    finally:
        lib.tune(23, 2000)
        lib.tune(39, 8)
    dplanes, gplanes = be.bytes_buf(dbytes), be.bytes_buf(gbytes)
    bscale, bsum = be.bytes_buf(4 * N), be.full((N, C), np.nan)
    lib.instnorm_bwd_planes(be.ptr(d_x1), be.ptr(d_st1), be.ptr(d_gy), be.ptr(gymax), N, C, H, W, act, 0.2, drop_p, 4242, 9,
                            1 if reflect else 0, be.ptr(d_gx), be.ptr(dplanes), be.ptr(gplanes), be.ptr(bscale), be.ptr(bsum), be.stream)
    be.sync()
    gx = be.np(d_gx).astype(np.float64)
    smax = np.abs(want_gx).reshape(N, -1).max(axis=1).reshape(N, 1, 1, 1)
    assert np.all(np.abs(gx - want_gx) <= 2e-5 * smax), ("gx", float((np.abs(gx - want_gx) / smax).max()))
    bb = be.raw(bscale)[:4 * N].view(np.float32).astype(np.float64)
    want_bb = st1[..., 1].max(axis=1).astype(np.float64) * (2 + np.sqrt(H * W)) / (1 - drop_p) * np.abs(gy).reshape(N, -1).max(axis=1)
    assert np.allclose(bb, want_bb, rtol=1e-6), (bb, want_bb)
    assert np.all(bb >= np.abs(gx).reshape(N, -1).max(axis=1)), "the bound is not a bound"
    bs = (2.0 ** (11 - np.floor(np.log2(bb)))).reshape(N, 1, 1, 1)
    dp = _decode_planes(be.raw(dplanes)[:dbytes], N, C, H, W)
    want_dp = _dgrad_plane_content(gx, reflect)
    err = np.abs((dp[0] + dp[1]) / bs - want_dp)
    mag_dp = _dgrad_plane_content(np.abs(gx), reflect)           # (the folded rows / slots are fp32 sums of up to four terms)
    assert np.all(err <= 2.0 ** -21 * mag_dp + 2.0 ** -23 / bs), ("data-gradient planes", float((err / (2.0 ** -21 * mag_dp + 2.0 ** -23 / bs)).max()))
    gp = _decode_pixel_planes(be.raw(gplanes)[:gbytes], N, C, Hg, CPR)
    want_gp = np.zeros((N, C, Hg, CPR * 8))
    want_gp[:, :, :H, :W] = gx
    err = np.abs((gp[0] + gp[1]) / bs - want_gp)
    assert np.all(err <= 2.0 ** -21 * np.abs(want_gp) + 2.0 ** -24 / bs), ("weight-gradient planes", float(err.max()))
    got_bsum = be.np(bsum).astype(np.float64)
    assert np.all(np.abs(got_bsum - gx.sum(axis=(2, 3))) <= 1e-5 * np.abs(gx).sum(axis=(2, 3)) + 1e-30), "bias partials"
    if producers_only:
        return

    # ---- the convolution's two gradient calls on those planes: w [K, C, 3, 3], layer input = y (the forward producer's output) ----
    w = (rng.standard_normal((K, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    skip = (rng.standard_normal((N, C, H, W)) * mags * 0.1).astype(np.float32)
    d_w, d_skip = be.dev(w), be.dev(skip)
    need = split16_scratch(be, N, H, W, K, C, 3, 3, 1, 1)
    arena = be.bytes_buf(need)
    lib.tune(23, 0)
    lib.tune(39, 1)
    try:
        wsd, wsdb = _ws(be, lib.conv2d_bwd_data_workspace(N, C, H, W, K, 3, 3, 1, 1, pad_mode))
        wsw, wswb = _ws(be, lib.conv2d_bwd_weight_workspace(N, C, H, W, K, H, W, 3, 3, 1, 1))
        d_gin = be.full((N, C, H, W), np.nan)
        omax = be.bytes_buf(4 * N * 2049)
        ghost = be.full((N, C, H, W), np.nan)            # the fp32 gx the planes stand for: must not be read
        e = ConvExtras()
        e.scratch, e.scratch_bytes = be.ptr(arena).value, need
        e.src_max_words, e.src_max_count = be.ptr(bscale).value, N
        e.src_planes = be.ptr(dplanes).value
        e.addend = be.ptr(d_skip).value
        e.out_max_words = be.ptr(omax).value
        lib.conv2d_bwd_data_ex(be.ptr(ghost), be.ptr(d_w), None, 0, 0.0, be.ptr(d_gin), C, None, 0, N, H, W, K, H, W, 3, 3, 1, 1,
                               pad_mode, be.ptr(wsd), wsdb, 0, be.stream, ctypes.byref(e))
        assert lib.last_route() == 2
        be.sync()
        gin = be.np(d_gin)
        want_gin, want_gw, _ = O.conv2d_bwd(y, w.astype(np.float64), gx, 1, 1, _PM[pad_mode])
        mag = O.conv2d_bwd(np.abs(y), np.abs(w.astype(np.float64)), np.abs(gx), 1, 1, _PM[pad_mode])[0]
        wsum = np.abs(w.astype(np.float64)).sum(axis=(0, 2, 3)).reshape(1, C, 1, 1)
        lim = 4e-6 * mag + 3e-11 * bb.reshape(N, 1, 1, 1) * wsum + 2e-7 * np.abs(skip) + 1e-30
        assert np.all(np.abs(gin - (want_gin + skip)) <= lim), ("data gradient from planes + skip", float((np.abs(gin - (want_gin + skip)) / lim).max()))
        got_omax = be.raw(omax)[:4 * N].view(np.uint32)
        assert np.array_equal(got_omax, np.abs(gin).reshape(N, -1).max(axis=1).astype(np.float32).view(np.uint32)), "epilogue max words"
        # ... and lazily: marker words, the same partial words, the same result; the backward producer fed with either comes out the same
        omax_lazy, d_gin2 = be.bytes_buf(4 * N * 2049), be.full((N, C, H, W), np.nan)
        e.out_max_words = be.ptr(omax_lazy).value
        lib.set_max_words_lazy(1)
        try:
            lib.conv2d_bwd_data_ex(be.ptr(ghost), be.ptr(d_w), None, 0, 0.0, be.ptr(d_gin2), C, None, 0, N, H, W, K, H, W, 3, 3, 1, 1,
                                   pad_mode, be.ptr(wsd), wsdb, 1, be.stream, ctypes.byref(e))
        finally:
            lib.set_max_words_lazy(0)
        be.sync()
        parts = (H // (256 // W)) * (C // 128)
        ol, oe = be.raw(omax_lazy)[:4 * N * 2049].view(np.uint32), be.raw(omax)[:4 * N * 2049].view(np.uint32)
        assert np.array_equal(be.np(d_gin2), gin)
        assert np.all(ol[:N] == (0xFFFF0000 | parts)), [hex(int(v)) for v in ol[:N]]
        assert np.array_equal(ol[N:N + N * parts], oe[N:N + N * parts]) and np.array_equal(oe[:N], oe[N:N + N * parts].reshape(N, parts).max(axis=1))
        res = []
        for words in (omax, omax_lazy):
            dpl, gpl, bsc = be.bytes_buf(dbytes), be.bytes_buf(gbytes), be.bytes_buf(4 * N)
            lib.instnorm_bwd_planes(be.ptr(d_x1), be.ptr(d_st1), be.ptr(d_gin), be.ptr(words), N, C, H, W, act, 0.2, 0.0, 0, 0,
                                    1 if reflect else 0, None, be.ptr(dpl), be.ptr(gpl), be.ptr(bsc), None, be.stream)
            be.sync()
            res.append((be.raw(bsc)[:4 * N].copy(), be.raw(dpl)[:dbytes].copy(), be.raw(gpl)[:gbytes].copy()))
        assert all(np.array_equal(a, b) for a, b in zip(*res)), "the backward producer on lazy gradient words differs from the one on finalized words"
        # weight gradient: both operands as planes
        d_gw = be.full((K, C, 3, 3), 0.25)
        ghost_x = be.full((N, C, H, W), np.nan)
        e2 = ConvExtras()
        e2.scratch, e2.scratch_bytes = be.ptr(arena).value, need
        e2.src_max_words, e2.src_max_count = be.ptr(fscale).value, N
        e2.src2_max_words, e2.src2_max_count = be.ptr(bscale).value, N
        e2.src2_planes = be.ptr(gplanes).value
        if reflect:
            e2.src_planes = be.ptr(xplanes).value
            xsrc = ghost_x
        else:
            xsrc = d_y                                    # (the X planes carry the reflect border: a zero-padded layer splits x itself)
        # ... and the bias gradient from the backward producer's per-plane sums, reduced inside the same call's slab-sum launch
        # (nemar_conv_extras.bias_partials) == nemar_bias_from_partials of the same sums, bit for bit
        d_gb, d_gb2 = be.full((K,), 0.5), be.full((K,), 0.5)
        e2.bias_partials = be.ptr(bsum).value
        lib.conv2d_bwd_weight_ex(be.ptr(xsrc), C, None, 0, be.ptr(ghost), be.ptr(d_gw), be.ptr(d_gb), N, H, W, K, H, W, 3, 3, 1, 1, pad_mode,
                                 be.ptr(wsw), wswb, be.stream, ctypes.byref(e2))
        assert lib.last_route() == 2
        lib.bias_from_partials(be.ptr(bsum), N, K, be.ptr(d_gb2), be.stream)
        be.sync()
        assert np.array_equal(be.np(d_gb), be.np(d_gb2)), "bias gradient riding in the slab-sum launch"
        gw = be.np(d_gw).astype(np.float64) - 0.25
        wmag = O.conv2d_bwd(np.abs(y), np.abs(w.astype(np.float64)), np.abs(gx), 1, 1, _PM[pad_mode])[1]
        lim = 6e-6 * wmag + 1e-6 * np.abs(want_gw).max() + 1e-30
        assert np.all(np.abs(gw - want_gw) <= lim), ("weight gradient from planes", float((np.abs(gw - want_gw) / lim).max()))
    finally:
        lib.tune(23, 2000)
        lib.tune(39, 8)


def case_pointwise(be, seed=0):
    rng = np.random.default_rng(seed)
    # act_bwd
    for act in (O.ACT_RELU, O.ACT_LRELU, O.ACT_TANH):
        n = 1027
        pre = rng.standard_normal(n).astype(np.float32)
        y = O.act_fwd(pre.astype(np.float64), act).astype(np.float32)
        gy = rng.standard_normal(n).astype(np.float32)
        want = O.act_bwd(gy.astype(np.float64), y.astype(np.float64), act)
        d_g = be.full((n,), np.nan)
        d_gy, d_y = be.dev(gy), be.dev(y)          # keep the device buffers alive across the call
        be.lib.act_bwd(be.ptr(d_gy), be.ptr(d_y), be.ptr(d_g), n, act, 0.2, be.stream)
        _assert_close(be.np(d_g), want, atol=1e-6, rtol=1e-6, what="act_bwd")
        d_pre, d_out = be.dev(pre), be.full((n,), np.nan)                 # stand-alone activation (U-Net skip path)
        be.lib.act_fwd(be.ptr(d_pre), be.ptr(d_out), n, act, 0.2, be.stream)
        _assert_close(be.np(d_out), O.act_fwd(pre.astype(np.float64), act), atol=1e-6, rtol=1e-6, what="act_fwd")
    # maxpool (even, odd sizes; ties)
    for (H, W) in ((8, 10), (7, 9), (2, 2)):
        x = rng.integers(-3, 4, (2, 3, H, W)).astype(np.float32)     # many exact ties
        want, _ = O.maxpool2_fwd(x.astype(np.float64))
        Ho, Wo = H // 2, W // 2
        d_x = be.dev(x)
        d_y = be.full((2, 3, Ho, Wo), np.nan)
        be.lib.maxpool2_fwd(be.ptr(d_x), be.ptr(d_y), 6, H, W, be.stream)
        _assert_close(be.np(d_y), want, atol=0, what="maxpool2_fwd")
        gy = rng.standard_normal((2, 3, Ho, Wo)).astype(np.float32)
        add = rng.standard_normal((2, 3, H, W)).astype(np.float32)
        want_g = O.maxpool2_bwd(x.astype(np.float64), gy.astype(np.float64))
        d_gx = be.full((2, 3, H, W), np.nan)
        d_gy, d_add = be.dev(gy), be.dev(add)
        be.lib.maxpool2_bwd(be.ptr(d_x), be.ptr(d_gy), None, be.ptr(d_gx), 6, H, W, be.stream)
        _assert_close(be.np(d_gx), want_g, atol=0, what="maxpool2_bwd")
        be.lib.maxpool2_bwd(be.ptr(d_x), be.ptr(d_gy), be.ptr(d_add), be.ptr(d_gx), 6, H, W, be.stream)
        _assert_close(be.np(d_gx), want_g + add, atol=1e-6, what="maxpool2_bwd+addend")
    # bilinear: 2x up (gather backward), /2 and /4 down, arbitrary
    for (H, W, Ho, Wo) in ((4, 5, 8, 10), (2, 2, 4, 4), (1, 3, 2, 6), (8, 12, 4, 6), (8, 12, 2, 3), (5, 7, 9, 4), (9, 12, 3, 2),
                           (6, 8, 3, 2)):
        x = rng.standard_normal((2, 2, H, W)).astype(np.float32)
        want = O.bilinear_resize_fwd(x.astype(np.float64), Ho, Wo)
        d_y = be.full((2, 2, Ho, Wo), np.nan)
        d_x = be.dev(x)
        be.lib.bilinear_fwd(be.ptr(d_x), be.ptr(d_y), 4, H, W, Ho, Wo, be.stream)
        _assert_close(be.np(d_y), want, atol=2e-6, what="bilinear_fwd %s" % ((H, W, Ho, Wo),))
        gy = rng.standard_normal((2, 2, Ho, Wo)).astype(np.float32)
        want_g = O.bilinear_resize_bwd(gy.astype(np.float64), H, W)
        d_gx = be.full((2, 2, H, W), np.nan)
        d_gy = be.dev(gy)
        be.lib.bilinear_bwd(be.ptr(d_gy), be.ptr(d_gx), 4, H, W, Ho, Wo, be.stream)
        _assert_close(be.np(d_gx), want_g, atol=5e-6, what="bilinear_bwd %s" % ((H, W, Ho, Wo),))


def case_concat_and_add(be, seed=0):
    """nemar_concat_pieces (batch concatenation / the gradient of batch slices, a NULL piece = zeros; 16-byte and scalar forms) and
    nemar_add2 (the sum of two consumers' gradients; aliasing the output) against numpy — bit for bit: copies and one fp32 add."""
    import ctypes
    rng = np.random.default_rng(seed)
    for counts, null in (((7200, 7200, 7200), None), ((7200, 7200, 7200), 1), ((1027, 5, 4096), None), ((48,), None), ((1030,), 0),
                         ((64, 64, 64, 64, 64, 64, 64, 64), 7), ((3, 1), 0)):
        pieces = [rng.standard_normal(c).astype(np.float32) for c in counts]
        devs = [None if i == null else be.dev(q) for i, q in enumerate(pieces)]
        want = np.concatenate([np.zeros(c, np.float32) if i == null else q for i, (q, c) in enumerate(zip(pieces, counts))])
        d_out = be.full((sum(counts),), np.nan)
        ptrs = (ctypes.c_void_p * len(counts))(*[None if d is None else be.ptr(d).value for d in devs])
        cnts = (ctypes.c_longlong * len(counts))(*counts)
        be.lib.concat_pieces(ptrs, cnts, len(counts), be.ptr(d_out), be.stream)
        got = be.np(d_out)
        assert got.shape == want.shape and (got == want.astype(np.float64)).all(), "concat_pieces %r null=%r" % (counts, null)
    for n in (4, 48, 1027, 65536 + 3):
        a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
        want = (a + b).astype(np.float64)
        d_a, d_b, d_o = be.dev(a.copy()), be.dev(b), be.full((n,), np.nan)
        be.lib.add2(be.ptr(d_a), be.ptr(d_b), be.ptr(d_o), n, be.stream)
        assert (be.np(d_o) == want).all(), "add2 n=%d" % n
        be.lib.add2(be.ptr(d_a), be.ptr(d_b), be.ptr(d_a), n, be.stream)          # in place
        assert (be.np(d_a) == want).all(), "add2 in place n=%d" % n


def case_dropout(be, n=40003, p=0.5):
    x = np.ones(n, dtype=np.float32) * 3
    d_x = be.dev(x)
    d_y = be.full((n,), np.nan)
    be.lib.dropout(be.ptr(d_x), be.ptr(d_y), n, p, 1234567, 7, be.stream)
    y = be.np(d_y)
    keep = y != 0
    assert np.allclose(y[keep], 3 / (1 - p))
    frac = keep.mean()
    assert abs(frac - (1 - p)) < 5 * np.sqrt(p * (1 - p) / n), frac
    d_y2 = be.full((n,), np.nan)
    be.lib.dropout(be.ptr(d_x), be.ptr(d_y2), n, p, 1234567, 7, be.stream)
    assert np.array_equal(be.np(d_y2), y)                 # same (seed, offset) -> same mask (backward relies on it)
    be.lib.dropout(be.ptr(d_x), be.ptr(d_y2), n, p, 1234567, 8, be.stream)
    assert not np.array_equal(be.np(d_y2), y)             # different offset -> different mask
    # lag-1 independence of neighbouring elements
    k = keep.astype(np.float64)
    assert abs(np.corrcoef(k[:-1], k[1:])[0, 1]) < 0.03


def case_losses(be, seed=0):
    rng = np.random.default_rng(seed)
    wsb = be.lib.loss_workspace()
    ws = be.bytes_buf(wsb)
    gs = be.dev(np.array([0.5], dtype=np.float32))
    for n in (7, 5000):
        a = rng.standard_normal(n).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        b[:3] = a[:3]                                  # sign(0) = 0
        for bb in (b, None):
            b64 = bb.astype(np.float64) if bb is not None else np.zeros(n)
            loss = be.full((1,), 2.0)
            d_a, d_b = be.dev(a), (be.dev(bb) if bb is not None else None)
            be.lib.l1_loss_fwd(be.ptr(d_a), be.ptr(d_b), n, 100.0, be.ptr(loss), 1, be.ptr(ws), wsb, 0, be.stream)
            _assert_close(be.np(loss), [2.0 + 100.0 * O.l1_loss_fwd(a.astype(np.float64), b64)], atol=1e-5, rtol=1e-5,
                          what="l1_loss_fwd")
            ga = be.full((n,), np.nan)
            be.lib.l1_loss_bwd(be.ptr(d_a), be.ptr(d_b), n, be.ptr(gs), 100.0, be.ptr(ga), 0, be.stream)
            _assert_close(be.np(ga), 50.0 * O.l1_loss_bwd(a.astype(np.float64), b64), atol=1e-9, rtol=1e-6, what="l1_loss_bwd")
        x = (rng.standard_normal(n) * 4).astype(np.float32)
        d_x = be.dev(x)
        for mode, name in ((0, 'vanilla'), (1, 'lsgan'), (2, 'wgangp')):
            for real in (1, 0):
                loss = be.full((1,), np.nan)
                be.lib.gan_loss_fwd(be.ptr(d_x), n, mode, real, 0.5, be.ptr(loss), 0, be.ptr(ws), wsb, 0, be.stream)
                _assert_close(be.np(loss), [0.5 * O.gan_loss_fwd(x.astype(np.float64), bool(real), name)], atol=1e-6,
                              rtol=1e-5, what="gan_loss_fwd " + name)
                gx = be.full((n,), np.nan)
                be.lib.gan_loss_bwd(be.ptr(d_x), n, mode, real, be.ptr(gs), 0.5, be.ptr(gx), be.stream)
                _assert_close(be.np(gx), 0.25 * O.gan_loss_bwd(x.astype(np.float64), bool(real), name), atol=1e-9,
                              rtol=2e-5, what="gan_loss_bwd " + name)


def case_adam(be, n=3001, steps=3, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, dtype=np.float32)
    v = np.zeros(n, dtype=np.float32)
    d_p, d_m, d_v = be.dev(p), be.dev(m), be.dev(v)
    p64, m64, v64 = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for step in range(1, steps + 1):
        g = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 1, n)).astype(np.float32)
        d_g = be.dev(g)
        be.lib.adam_step(be.ptr(d_p), be.ptr(d_g), be.ptr(d_m), be.ptr(d_v), n, 2e-4, 0.5, 0.999, 1e-8, step, be.stream)
        p64, m64, v64 = O.adam_step(p64, g.astype(np.float64), m64, v64, step)
    _assert_close(be.np(d_p), p64, atol=2e-7, rtol=2e-7, what="adam p")
    # g spans six decades, so m = lerp(m, g) cancels: fp32 rounding is relative to the larger operand
    _assert_close(be.np(d_m), m64, atol=1e-7 * np.abs(m64).max(), rtol=1e-6, what="adam m")
    _assert_close(be.np(d_v), v64, atol=1e-7 * np.abs(v64).max(), rtol=1e-6, what="adam v")


def case_crop_flip_normalize(be, seed=0):
    """Input-pipeline augmentation: crop + flip + Normalize(0.5, 0.5) of a resident pool (numpy restatement of the reference's
    get_transform chain, data/base_dataset.py:81-112)."""
    rng = np.random.default_rng(seed)
    M, C, H, W, Hc, Wc, B = 5, 3, 11, 14, 8, 9, 4
    pool = rng.uniform(0, 1, (M, C, H, W)).astype(np.float32)
    params = np.array([[3, 0, 0, 0], [0, 3, 5, 1], [4, 2, 1, 1], [1, 1, 4, 0]], dtype=np.int32)
    want = np.zeros((B, C, Hc, Wc))
    for b, (i, y0, x0, flip) in enumerate(params):
        crop = pool[i, :, y0:y0 + Hc, x0:x0 + Wc].astype(np.float64)
        if flip:
            crop = crop[:, :, ::-1]
        want[b] = (crop - 0.5) / 0.5
    d_pool, d_par = be.dev(pool), be.dev_i32(params)
    d_y = be.full((B, C, Hc, Wc), np.nan)
    be.lib.crop_flip_normalize(be.ptr(d_pool), be.ptr(d_par), be.ptr(d_y), M, B, C, H, W, Hc, Wc, 1.0, be.stream)
    _assert_close(be.np(d_y), want, atol=1e-6, what="crop_flip_normalize")
