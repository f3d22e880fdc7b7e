"""`-m gpu`: the convolution entry points at the REAL shapes of the bench step (BASELINE config 2, batch 8), with the
library's DEFAULT tile selection — the dispatch that the small kernel cases never reach (igemm_ws2_kernel<4,true> on
512 workgroups, wgrad2_kernel<true,2,2,2> one-round splits, the reflect data gradient at 64x64, the four parity
classes of the stride-2 data gradients, the narrow 7x7 kernels at 256x256).

Oracle: torch CPU convolutions (the reference's own arithmetic).  Two tiers per case:
  * fp64 on a SUBSET of channels that touches every 32-channel MFMA row tile / every weight-gradient tile (an exact
    value; the subset keeps the CPU time to a second or two): tolerance atol 2e-5 + rtol 2e-5 (fwd/dgrad), 5e-5 (wgrad);
  * fp32 on the FULL tensor (MKLDNN, a different summation order than ours): max-abs error <= 2e-4 of the tensor's max.
All through the C ABI (include/nemar_hip.h), exactly as nemar_amd/ops.py calls it."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from backends import HipBackend

pytestmark = pytest.mark.gpu
PAD_ZERO, PAD_REFLECT = 0, 1


@pytest.fixture(scope="module")
def be(hip_lib):
    return HipBackend(hip_lib)


def _subset(n, k=12):
    """channel subset hitting both ends of every 32-wide tile boundary region"""
    if n <= k:
        return list(range(n))
    pts = {0, 1, n - 1, n - 2}
    for b in range(32, n, 32):
        pts.update((b - 1, b))
    pts = sorted(p for p in pts if 0 <= p < n)
    if len(pts) > k:
        idx = np.linspace(0, len(pts) - 1, k).round().astype(int)
        pts = [pts[i] for i in idx]
    return pts


def _conv_cpu(x, w, b, stride, pad, pm):
    if pm == PAD_REFLECT and pad:
        x = F.pad(x, (pad, pad, pad, pad), mode='reflect')
        pad = 0
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def _close(got, want, atol, rtol, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want)
    lim = atol + rtol * np.abs(want)
    if not np.all(err <= lim):
        i = np.unravel_index(np.argmax(err - lim), err.shape)
        raise AssertionError("%s: max|err|=%.3e at %s (got %.6g want %.6g)" % (what, err.max(), i, got[i], want[i]))


# name, N, C0, C1, H, W, K, R, stride, pad, pad_mode
SHAPES = [
    ("T resblock 256->256 k3 reflect 64x64", 8, 256, 0, 64, 64, 256, 3, 1, 1, PAD_REFLECT),
    ("D layer4 256->512 k4 s1 32x32->31x31", 8, 256, 0, 32, 32, 512, 4, 1, 1, PAD_ZERO),
    ("R up_1 64+32->32 k3 256x256", 8, 64, 32, 256, 256, 32, 3, 1, 1, PAD_ZERO),
    ("T down1 64->128 k3 s2 256x256", 8, 64, 0, 256, 256, 128, 3, 2, 1, PAD_ZERO),
    ("T down2 128->256 k3 s2 128x128", 8, 128, 0, 128, 128, 256, 3, 2, 1, PAD_ZERO),
    ("D layer2 64->128 k4 s2 128x128", 8, 64, 0, 128, 128, 128, 4, 2, 1, PAD_ZERO),
    ("D layer3 128->256 k4 s2 64x64", 8, 128, 0, 64, 64, 256, 4, 2, 1, PAD_ZERO),
    ("D layer1 3+3->64 k4 s2 256x256", 8, 3, 3, 256, 256, 64, 4, 2, 1, PAD_ZERO),
    ("D logits 512->1 k4 s1 31x31", 8, 512, 0, 31, 31, 1, 4, 1, 1, PAD_ZERO),
    ("T stem 3->64 k7 reflect 256x256", 8, 3, 0, 256, 256, 64, 7, 1, 3, PAD_REFLECT),
    ("T head 64->3 k7 reflect 256x256", 8, 64, 0, 256, 256, 3, 7, 1, 3, PAD_REFLECT),
    ("R resblock 32->32 k3 reflect 256x256", 8, 32, 0, 256, 256, 32, 3, 1, 1, PAD_REFLECT),
    ("R resblock 64->64 k3 reflect 128x128", 8, 64, 0, 128, 128, 64, 3, 1, 1, PAD_REFLECT),
    ("R down_1 3+3->32 k3 256x256", 8, 3, 3, 256, 256, 32, 3, 1, 1, PAD_ZERO),
    ("R bottleneck 128->128 k3 reflect 2x2", 8, 128, 0, 2, 2, 128, 3, 1, 1, PAD_REFLECT),
    ("R output 32->2 k3 256x256", 8, 32, 0, 256, 256, 2, 3, 1, 1, PAD_ZERO),
]


@pytest.mark.parametrize("route", ["default", "exact"])
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_conv_real_shape(be, shape, route):
    """No scratch arena registered.  default: the library's own dispatch — the general 16-bit-pipe kernels with the in-kernel operand
    split (csrc/conv_s16g*.hip) for every layer they take; exact: nemar_tune(24, 0), every layer on the exact-fp32 MFMA / VALU
    kernels.  Same tolerances."""
    be.lib.tune(24, 0 if route == "exact" else 1)
    try:
        _run_shape(be, shape)
    finally:
        be.lib.tune(24, 1)


SPLIT16_SHAPES = [
    ("T resblock 256->256 k3 reflect 64x64", 8, 256, 0, 64, 64, 256, 3, 1, 1, PAD_REFLECT),
    ("T resblock 256->256 k3 reflect 128x128 (512^2 input)", 2, 256, 0, 128, 128, 256, 3, 1, 1, PAD_REFLECT),
    ("wide zero-padded 128->128 k3 32x32", 8, 128, 0, 32, 32, 128, 3, 1, 1, PAD_ZERO),
    ("D layer4 256->512 k4 s1 32x32->31x31", 8, 256, 0, 32, 32, 512, 4, 1, 1, PAD_ZERO),
]


@pytest.mark.parametrize("shape", SPLIT16_SHAPES, ids=[s[0] for s in SPLIT16_SHAPES])
def test_conv_real_shape_split16(be, shape):
    """Scratch arena registered, as nemar_amd/ops.py does: forward and data gradient of the wide 3x3 layers run on the
    split-16 kernels (csrc/conv_split16.hip) and must obey the SAME tolerances as the exact-fp32 kernels."""
    from kernel_cases import scratch_arena, split16_scratch
    name, N, C0, C1, H, W, K, R, stride, pad, pm = shape
    need = split16_scratch(be, N, H, W, K, C0 + C1, R, R, stride, pad)
    assert need > 0
    with scratch_arena(be, need):
        _run_shape(be, shape)


def test_split16_error_is_fp32_class(be):
    """The accuracy claim of csrc/conv_split16.hip, measured: resblock forward and reflect data gradient at the bench shape, both
    routes against the same float64 reference on a channel subset.  The split-16 route must not be worse than 1.5x the
    exact-fp32 route's own max error (+ one fp32 ulp of the result scale), and repeated calls must agree bitwise."""
    from kernel_cases import scratch_arena
    N, C, H, W, K = 8, 256, 64, 64, 256
    g = torch.Generator().manual_seed(20260927)
    x = torch.rand((N, C, H, W), generator=g) * 2 - 1
    w = torch.randn((K, C, 3, 3), generator=g) / np.sqrt(C * 9)
    gy = torch.randn((N, K, H, W), generator=g)
    lib, P = be.lib, be.ptr
    d_x, d_w, d_gy = be.dev(x.numpy()), be.dev(w.numpy()), be.dev(gy.numpy())
    ks = _subset(K, 16)
    want_f = _conv_cpu(x.double(), w[ks].double(), None, 1, 1, PAD_REFLECT).numpy()
    xs = torch.zeros((N, len(ks), H, W), dtype=torch.float64, requires_grad=True)
    _conv_cpu(xs, w[:, ks].double(), None, 1, 1, PAD_REFLECT).backward(gy.double())
    want_d = xs.grad.numpy()

    def run():
        d_y, d_g = be.full((N, K, H, W), np.nan), be.full((N, C, H, W), np.nan)
        wsb = lib.conv2d_fwd_workspace(N, H, W, K, C, 3, 3, 1, 1)
        ws = be.bytes_buf(wsb)
        lib.conv2d_fwd(P(d_x), C, None, 0, P(d_w), None, P(d_y), N, H, W, K, 3, 3, 1, 1, PAD_REFLECT, 0, 0.2, P(ws), wsb, 0,
                       be.stream)
        wsb = lib.conv2d_bwd_data_workspace(N, C, H, W, K, 3, 3, 1, 1, PAD_REFLECT)
        ws = be.bytes_buf(wsb)
        lib.conv2d_bwd_data(P(d_gy), P(d_w), None, 0, 0.0, P(d_g), C, None, 0, N, H, W, K, H, W, 3, 3, 1, 1, PAD_REFLECT,
                            P(ws), wsb, 0, be.stream)
        d_gw = be.full((K, C, 3, 3), 0.0)
        wsb = lib.conv2d_bwd_weight_workspace(N, C, H, W, K, H, W, 3, 3, 1, 1)
        ws = be.bytes_buf(wsb)
        lib.conv2d_bwd_weight(P(d_x), C, None, 0, P(d_gys), P(d_gw), None, N, H, W, K, H, W, 3, 3, 1, 1, PAD_REFLECT, P(ws), wsb,
                              be.stream)
        return be.np(d_y), be.np(d_g), be.np(d_gw)

    gys = gy / np.sqrt(N * H * W)
    d_gys = be.dev(gys.numpy())
    wz = torch.zeros((len(ks), C, 3, 3), dtype=torch.float64, requires_grad=True)
    _conv_cpu(x.double(), wz, None, 1, 1, PAD_REFLECT).backward(gys[:, ks].double())
    want_w = wz.grad.numpy()
    y32, g32, w32 = run()
    with scratch_arena(be, lib.conv2d_scratch(N, H, W, K, C, 3, 3, 1, 1)):
        y6, g6, w6 = run()
        y6b, g6b, w6b = run()
    assert np.array_equal(y6, y6b) and np.array_equal(g6, g6b) and np.array_equal(w6, w6b), "split-16 route is not reproducible"
    assert not np.array_equal(y6, y32) and not np.array_equal(w6, w32), "the arena did not switch the route"
    rows = []
    for what, a32, a6, want, sel in (("fwd", y32, y6, want_f, ks), ("dgrad", g32, g6, want_d, ks), ("wgrad", w32, w6, want_w, None)):
        if sel is None:
            a32, a6 = a32[ks][None], a6[ks][None]
            sel = slice(None)
        e32 = np.abs(a32[:, sel] - want)
        e6 = np.abs(a6[:, sel] - want)
        scale = np.abs(want).max()
        rows.append((what, e32.max(), e6.max(), np.sqrt((e32 ** 2).mean()), np.sqrt((e6 ** 2).mean()), scale))
    import os
    rep = os.environ.get('NEMAR_SPLIT16_REPORT')
    if rep:
        with open(rep, 'a') as f:
            for r in rows:
                f.write("%-6s max|err| exact-fp32 %.3e  split-16 %.3e   rms exact-fp32 %.3e  split-16 %.3e   (result scale %.3g)\n" % r)
    for what, m32, m6, r32, r6, scale in rows:
        assert m6 <= 1.5 * m32 + 1.2e-7 * scale, (what, m32, m6)
        assert r6 <= 1.5 * r32, (what, r32, r6)


def _run_shape(be, shape):
    name, N, C0, C1, H, W, K, R, stride, pad, pm = shape
    C = C0 + C1
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    x = torch.rand((N, C, H, W), generator=g) * 2 - 1
    w = torch.randn((K, C, R, R), generator=g) / np.sqrt(C * R * R)
    b = torch.randn((K,), generator=g)
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    gy = torch.randn((N, K, OH, OW), generator=g)
    lib, P = be.lib, be.ptr
    d_x0, d_x1 = be.dev(x[:, :C0].numpy()), (be.dev(x[:, C0:].numpy()) if C1 else None)
    d_w, d_b, d_gy = be.dev(w.numpy()), be.dev(b.numpy()), be.dev(gy.numpy())

    # ---- forward ---------------------------------------------------------------------------------------------
    d_y = be.full((N, K, OH, OW), np.nan)
    wsb = lib.conv2d_fwd_workspace(N, H, W, K, C, R, R, stride, pad)
    ws = be.bytes_buf(wsb)
    lib.conv2d_fwd(P(d_x0), C0, P(d_x1), C1, P(d_w), P(d_b), P(d_y), N, H, W, K, R, R, stride, pad, pm, 0, 0.2, P(ws),
                   wsb, 0, be.stream)
    y = be.np(d_y)
    ks = _subset(K)
    want = _conv_cpu(x.double(), w[ks].double(), b[ks].double(), stride, pad, pm).numpy()
    _close(y[:, ks], want, 2e-5, 2e-5, name + " fwd (fp64 subset)")
    full = _conv_cpu(x, w, b, stride, pad, pm).numpy()
    assert np.abs(y - full).max() <= 2e-4 * np.abs(full).max(), (name, "fwd full", np.abs(y - full).max())

    # ---- data gradient ---------------------------------------------------------------------------------------
    if not (pm == PAD_REFLECT and C1):
        d_g0 = be.full((N, C0, H, W), np.nan)
        d_g1 = be.full((N, C1, H, W), np.nan) if C1 else None
        wsb = lib.conv2d_bwd_data_workspace(N, C, H, W, K, R, R, stride, pad, pm)
        ws = be.bytes_buf(wsb)
        lib.conv2d_bwd_data(P(d_gy), P(d_w), None, 0, 0.0, P(d_g0), C0, P(d_g1), C1, N, H, W, K, OH, OW, R, R, stride,
                            pad, pm, P(ws), wsb, 0, be.stream)
        gx = np.concatenate([be.np(d_g0)] + ([be.np(d_g1)] if C1 else []), axis=1)
        cs = _subset(C)
        xs = torch.zeros((N, len(cs), H, W), dtype=torch.float64, requires_grad=True)
        _conv_cpu(xs, w[:, cs].double(), None, stride, pad, pm).backward(gy.double())
        _close(gx[:, cs], xs.grad.numpy(), 3e-5, 2e-5, name + " dgrad (fp64 subset)")
        xf = torch.zeros((N, C, H, W), requires_grad=True)
        _conv_cpu(xf, w, None, stride, pad, pm).backward(gy)
        ref = xf.grad.numpy()
        assert np.abs(gx - ref).max() <= 2e-4 * np.abs(ref).max(), (name, "dgrad full", np.abs(gx - ref).max())

    # ---- weight + bias gradient (accumulating) -----------------------------------------------------------------
    gys = gy / np.sqrt(N * OH * OW)
    d_gys = be.dev(gys.numpy())
    d_gw, d_gb = be.full((K, C, R, R), 0.5), be.full((K,), 0.125)
    wsb = lib.conv2d_bwd_weight_workspace(N, C, H, W, K, OH, OW, R, R, stride, pad)
    ws = be.bytes_buf(wsb)
    lib.conv2d_bwd_weight(P(d_x0), C0, P(d_x1), C1, P(d_gys), P(d_gw), P(d_gb), N, H, W, K, OH, OW, R, R, stride, pad,
                          pm, P(ws), wsb, be.stream)
    gw, gb = be.np(d_gw) - 0.5, be.np(d_gb) - 0.125
    ks = _subset(K)
    wz = torch.zeros((len(ks), C, R, R), dtype=torch.float64, requires_grad=True)
    bz = torch.zeros((len(ks),), dtype=torch.float64, requires_grad=True)
    _conv_cpu(x.double(), wz, bz, stride, pad, pm).backward(gys[:, ks].double())
    _close(gw[ks], wz.grad.numpy(), 5e-5, 5e-5, name + " wgrad (fp64 subset)")
    _close(gb[ks], bz.grad.numpy(), 5e-5, 5e-5, name + " bias grad (fp64 subset)")
    wf = torch.zeros((K, C, R, R), requires_grad=True)
    _conv_cpu(x, wf, None, stride, pad, pm).backward(gys)
    ref = wf.grad.numpy()
    assert np.abs(gw - ref).max() <= 2e-4 * np.abs(ref).max() + 2e-6, (name, "wgrad full", np.abs(gw - ref).max())


CONVT = [("T up1 convT 256->128 k3 s2 64x64->128x128", 8, 256, 128, 64, 64, 3, 1),
         ("T up2 convT 128->64 k3 s2 128x128->256x256", 8, 128, 64, 128, 128, 3, 1)]


@pytest.mark.parametrize("route", ["default", "exact"])
@pytest.mark.parametrize("shape", CONVT, ids=[s[0] for s in CONVT])
def test_conv_transpose_real_shape(be, shape, route):
    be.lib.tune(24, 0 if route == "exact" else 1)
    try:
        _run_convt(be, shape)
    finally:
        be.lib.tune(24, 1)


def _run_convt(be, shape):
    name, N, Ci, Co, H, W, R, op = shape
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    x = torch.rand((N, Ci, H, W), generator=g) * 2 - 1
    w = torch.randn((Ci, Co, R, R), generator=g) / np.sqrt(Ci * R * R)
    b = torch.randn((Co,), generator=g)
    lib, P = be.lib, be.ptr
    Ho, Wo = (H - 1) * 2 - 2 + R + op, (W - 1) * 2 - 2 + R + op
    d_x, d_w, d_b = be.dev(x.numpy()), be.dev(w.numpy()), be.dev(b.numpy())
    d_y = be.full((N, Co, Ho, Wo), np.nan)
    wsb = lib.conv2d_bwd_data_workspace(N, Co, Ho, Wo, Ci, R, R, 2, 1, PAD_ZERO)
    ws = be.bytes_buf(wsb)
    lib.conv2d_bwd_data(P(d_x), P(d_w), P(d_b), 0, 0.0, P(d_y), Co, None, 0, N, Ho, Wo, Ci, H, W, R, R, 2, 1, PAD_ZERO,
                        P(ws), wsb, 0, be.stream)
    y = be.np(d_y)
    cs = _subset(Co)
    want = F.conv_transpose2d(x.double(), w[:, cs].double(), b[cs].double(), stride=2, padding=1, output_padding=op).numpy()
    _close(y[:, cs], want, 2e-5, 2e-5, name + " fwd (fp64 subset)")
    full = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=op).numpy()
    assert np.abs(y - full).max() <= 2e-4 * np.abs(full).max()
    # weight gradient of the transposed conv = conv weight gradient with the roles of input and gradient swapped
    gy = torch.randn((N, Co, Ho, Wo), generator=g) / np.sqrt(N * Ho * Wo)
    d_gy = be.dev(gy.numpy())
    d_gw = be.full((Ci, Co, R, R), 0.0)
    wsb = lib.conv2d_bwd_weight_workspace(N, Co, Ho, Wo, Ci, H, W, R, R, 2, 1)
    ws = be.bytes_buf(wsb)
    lib.conv2d_bwd_weight(P(d_gy), Co, None, 0, P(d_x), P(d_gw), None, N, Ho, Wo, Ci, H, W, R, R, 2, 1, PAD_ZERO, P(ws), wsb,
                          be.stream)
    wf = w.clone().requires_grad_(True)
    F.conv_transpose2d(x, wf, None, stride=2, padding=1, output_padding=op).backward(gy)
    ref = wf.grad.numpy()
    assert np.abs(be.np(d_gw) - ref).max() <= 2e-4 * np.abs(ref).max() + 2e-6, name
