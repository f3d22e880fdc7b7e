"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy, dtype-generic: run it in float64 for a tight reference or float32 to mimic
the reference's arithmetic) of every operator on the NeMAR training-step hot path,
`NEMARModel.optimize_parameters()` (reference models/nemar_model.py:266-288).

The reference contains no arithmetic of its own: every number is produced by PyTorch ATen operators
called from its Python modules.  PyTorch is a third-party dependency that is NOT
vendored under /root/reference and is NOT version-pinned by it (scripts/conda_deps.sh:3 installs an
unpinned `pytorch`); the oracle of record is "reference source + torch 2.10.0 CPU fp32".  Each function
below restates the published algorithm of one ATen operator and cites the reference call site(s) it
stands in for.  Pinning: `tests/golden/make_golden.py` imports the reference in the build container
and stores its outputs as fixtures; `tests/test_oracle_golden.py` checks these functions against those
fixtures (and, where torch is importable, against torch autograd directly).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import numpy as np


# ------------------------------------------------------------------------------------------------
# K9 / K10: sampling grids
# ------------------------------------------------------------------------------------------------
def linspace_m1_p1(n, dtype=np.float64):
    """torch.linspace(-1, 1, n) — reference models/stn/unet_stn.py:123-124.
    ATen fills symmetrically from both ends with a fused multiply-add: fma(step, i, start) for i < n//2,
    else fma(-step, n-1-i, end); step is rounded to the working precision first (bit-exact vs torch CPU)."""
    dtype = np.dtype(dtype).type
    if n == 1:
        return np.array([-1.0], dtype=dtype)
    step = np.longdouble(dtype(2.0) / dtype(n - 1))      # product of two working-precision values is exact here
    i = np.arange(n)
    lo = (np.longdouble(-1.0) + step * i).astype(dtype)
    hi = (np.longdouble(1.0) - step * (n - 1 - i)).astype(dtype)
    return np.where(i < n // 2, lo, hi).astype(dtype)


def unet_identity_grid(H, W, dtype=np.float64):
    """UnetSTN.get_identity_grid — reference models/stn/unet_stn.py:121-129.
    Returns [1,2,H,W]; channel 0 = x (width) coordinate, channel 1 = y."""
    x = linspace_m1_p1(W, dtype)
    y = linspace_m1_p1(H, dtype)
    g = np.empty((1, 2, H, W), dtype=dtype)
    g[0, 0] = x[None, :]
    g[0, 1] = y[:, None]
    return g


def unet_grid(offsets):
    """(identity.repeat(B) + deformation).permute(0,2,3,1) — reference models/stn/unet_stn.py:167.
    offsets [N,2,H,W] -> grid [N,H,W,2]."""
    N, _, H, W = offsets.shape
    g = unet_identity_grid(H, W, offsets.dtype) + offsets
    return np.ascontiguousarray(g.transpose(0, 2, 3, 1))


def affine_grid(theta, H, W):
    """F.affine_grid(theta.view(-1,2,3), size) with align_corners=False —
    reference models/stn/affine_stn.py:105,128.  theta [N,2,3] -> grid [N,H,W,2]."""
    dt = theta.dtype
    xs = (2.0 * np.arange(W, dtype=dt) + 1.0) / dt.type(W) - 1.0
    ys = (2.0 * np.arange(H, dtype=dt) + 1.0) / dt.type(H) - 1.0
    base = np.stack([np.broadcast_to(xs[None, :], (H, W)), np.broadcast_to(ys[:, None], (H, W)),
                     np.ones((H, W), dtype=dt)], axis=-1)          # [H,W,3]
    return np.einsum('hwk,nik->nhwi', base, theta).astype(dt)


def affine_theta(dtheta):
    """theta = dtheta + identity — reference models/stn/affine_stn.py:96,122.  [N,6] -> [N,2,3]."""
    ident = np.array([1, 0, 0, 0, 1, 0], dtype=dtype_of(dtheta))
    return (dtheta + ident[None, :]).reshape(-1, 2, 3)


def dtype_of(a):
    return a.dtype.type


# ------------------------------------------------------------------------------------------------
# K11: grid_sample (bilinear, zeros padding, align_corners=False)
# ------------------------------------------------------------------------------------------------
def _locate(grid, H, W):
    dt = grid.dtype.type
    ix = ((grid[..., 0] + dt(1)) * dt(W) - dt(1)) / dt(2)
    iy = ((grid[..., 1] + dt(1)) * dt(H) - dt(1)) / dt(2)
    x0f = np.floor(ix)
    y0f = np.floor(iy)
    tx = ix - x0f
    ty = iy - y0f
    x0 = np.clip(x0f, -2, W + 1).astype(np.int64)
    y0 = np.clip(y0f, -2, H + 1).astype(np.int64)
    return x0, y0, tx, ty


def _gather(inp, n_idx, yy, xx):
    """inp [N,C,H,W]; yy,xx [N,Ho,Wo] int -> values [N,C,Ho,Wo] with zeros out of bounds."""
    N, C, H, W = inp.shape
    valid = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
    xc = np.clip(xx, 0, W - 1)
    yc = np.clip(yy, 0, H - 1)
    v = inp[n_idx[:, None, None], :, yc, xc]          # [N,Ho,Wo,C]
    v = v * valid[..., None]
    return np.ascontiguousarray(v.transpose(0, 3, 1, 2)), valid


def grid_sample_fwd(inp, grid):
    """F.grid_sample(inp, grid, mode='bilinear', padding_mode='zeros', align_corners=False) —
    reference models/stn/unet_stn.py:173-174 and models/stn/affine_stn.py:129-130.
    ATen grid_sampler_2d: ix=((gx+1)W-1)/2; corners (x0,y0)..(x0+1,y0+1); weights
    nw=(1-tx)(1-ty), ne=tx(1-ty), sw=(1-tx)ty, se=tx*ty; out-of-bounds corners contribute 0."""
    N, C, H, W = inp.shape
    x0, y0, tx, ty = _locate(grid, H, W)
    n_idx = np.arange(N)
    one = inp.dtype.type(1)
    ex, ey = one - tx, one - ty
    a, _ = _gather(inp, n_idx, y0, x0)
    b, _ = _gather(inp, n_idx, y0, x0 + 1)
    c, _ = _gather(inp, n_idx, y0 + 1, x0)
    d, _ = _gather(inp, n_idx, y0 + 1, x0 + 1)
    return (a * (ex * ey)[:, None] + b * (tx * ey)[:, None] + c * (ex * ty)[:, None] + d * (tx * ty)[:, None])


def grid_sample_bwd(inp, grid, gout):
    """Backward of grid_sample_fwd (ATen grid_sampler_2d_backward).
    Returns (grad_input [N,C,H,W], grad_grid [N,Ho,Wo,2])."""
    N, C, H, W = inp.shape
    x0, y0, tx, ty = _locate(grid, H, W)
    n_idx = np.arange(N)
    one = inp.dtype.type(1)
    ex, ey = one - tx, one - ty
    a, va = _gather(inp, n_idx, y0, x0)
    b, vb = _gather(inp, n_idx, y0, x0 + 1)
    c, vc = _gather(inp, n_idx, y0 + 1, x0)
    d, vd = _gather(inp, n_idx, y0 + 1, x0 + 1)
    gix = np.sum(gout * ((b - a) * ey[:, None] + (d - c) * ty[:, None]), axis=1)
    giy = np.sum(gout * ((c - a) * ex[:, None] + (d - b) * tx[:, None]), axis=1)
    ggrid = np.stack([gix * inp.dtype.type(W) / 2, giy * inp.dtype.type(H) / 2], axis=-1)
    gin = np.zeros_like(inp)
    nn = np.broadcast_to(n_idx[:, None, None], x0.shape)
    for (yy, xx, wgt, valid) in ((y0, x0, ex * ey, va), (y0, x0 + 1, tx * ey, vb),
                                 (y0 + 1, x0, ex * ty, vc), (y0 + 1, x0 + 1, tx * ty, vd)):
        m = valid
        for ch in range(C):
            np.add.at(gin[:, ch], (nn[m], yy[m], xx[m]), (gout[:, ch] * wgt)[m])
    return gin, ggrid


def unet_warp_bwd(inp, offsets, gout):
    """grad wrt the planar offsets [N,2,H,W] of grid_sample(inp, unet_grid(offsets))."""
    gin, ggrid = grid_sample_bwd(inp, unet_grid(offsets), gout)
    return gin, np.ascontiguousarray(ggrid.transpose(0, 3, 1, 2))


def affine_warp_bwd(inp, dtheta, gout, Ho, Wo):
    """grad wrt dtheta [N,6] of grid_sample(inp, affine_grid(dtheta + I))."""
    theta = affine_theta(dtheta)
    grid = affine_grid(theta, Ho, Wo)
    gin, ggrid = grid_sample_bwd(inp, grid, gout)
    dt = inp.dtype
    xs = (2.0 * np.arange(Wo, dtype=dt) + 1.0) / dt.type(Wo) - 1.0
    ys = (2.0 * np.arange(Ho, dtype=dt) + 1.0) / dt.type(Ho) - 1.0
    base = np.stack([np.broadcast_to(xs[None, :], (Ho, Wo)), np.broadcast_to(ys[:, None], (Ho, Wo)),
                     np.ones((Ho, Wo), dtype=dt)], axis=-1)
    gtheta = np.einsum('nhwi,hwk->nik', ggrid, base).reshape(-1, 6)
    return gin, gtheta


# ------------------------------------------------------------------------------------------------
# K12: smoothness / bilateral regulariser
# ------------------------------------------------------------------------------------------------
def _pairs(x):
    """The four index pairs (a, b) of reference models/stn/stn_losses.py:12-15, diff = a - b."""
    return ((x[:, :, 1:, :], x[:, :, :-1, :]),
            (x[:, :, :, 1:], x[:, :, :, :-1]),
            (x[:, :, :-1, :-1], x[:, :, 1:, 1:]),
            (x[:, :, :-1, 1:], x[:, :, 1:, :-1]))


def smoothness_fwd(d, img=None, alpha=0.0):
    """smoothness_loss — reference models/stn/stn_losses.py:4-30 (four separate means)."""
    loss = d.dtype.type(0)
    bil = img is not None and alpha > 0.0
    ipairs = _pairs(img) if bil else [None] * 4
    for (a, b), ip in zip(_pairs(d), ipairs):
        diff = np.abs(a - b)
        if bil:
            w = np.mean(np.exp(-d.dtype.type(alpha) * np.abs(ip[0] - ip[1])), axis=1, keepdims=True)
            diff = w * diff
        loss = loss + np.mean(diff)
    return loss


def smoothness_bwd(d, img=None, alpha=0.0):
    """d loss / d d for smoothness_fwd (weights carry no gradient: image detached, unet_stn.py:183)."""
    g = np.zeros_like(d)
    N, C, H, W = d.shape
    bil = img is not None and alpha > 0.0
    ipairs = _pairs(img) if bil else [None] * 4
    views = ((g[:, :, 1:, :], g[:, :, :-1, :]),
             (g[:, :, :, 1:], g[:, :, :, :-1]),
             (g[:, :, :-1, :-1], g[:, :, 1:, 1:]),
             (g[:, :, :-1, 1:], g[:, :, 1:, :-1]))
    for (a, b), ip, (ga, gb) in zip(_pairs(d), ipairs, views):
        s = np.sign(a - b)
        if bil:
            w = np.mean(np.exp(-d.dtype.type(alpha) * np.abs(ip[0] - ip[1])), axis=1, keepdims=True)
            s = w * s
        s = s / d.dtype.type(a.size)
        ga += s
        gb -= s
    return g


# ------------------------------------------------------------------------------------------------
# K4 + K1/K2: padding and convolution (direct definition; small shapes only)
# ------------------------------------------------------------------------------------------------
def reflect_index(i, n):
    """nn.ReflectionPad2d index map — reference models/networks.py:349,375,418,432."""
    i = np.where(i < 0, -i, i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def pad2d(x, p, mode):
    if p == 0:
        return x
    if mode == 'reflect':
        return np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)), mode='reflect')
    return np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))


def conv2d_fwd(x, w, b=None, stride=1, pad=0, pad_mode='zeros'):
    """nn.Conv2d — reference models/networks.py:350,357,426,439,576,583,591,597; models/stn/layers.py:85.
    x [N,C,H,W], w [K,C,R,S] -> [N,K,Ho,Wo].  pad_mode 'reflect' folds the preceding ReflectionPad2d."""
    xp = pad2d(x, pad, pad_mode)
    N, C, Hp, Wp = xp.shape
    K, _, R, S = w.shape
    Ho = (Hp - R) // stride + 1
    Wo = (Wp - S) // stride + 1
    out = np.zeros((N, K, Ho, Wo), dtype=x.dtype)
    for r in range(R):
        for s in range(S):
            patch = xp[:, :, r:r + stride * (Ho - 1) + 1:stride, s:s + stride * (Wo - 1) + 1:stride]
            out += np.einsum('nchw,kc->nkhw', patch, w[:, :, r, s])
    if b is not None:
        out += b[None, :, None, None]
    return out


def conv2d_bwd(x, w, gout, stride=1, pad=0, pad_mode='zeros'):
    """Returns (grad_x, grad_w, grad_b) of conv2d_fwd."""
    xp = pad2d(x, pad, pad_mode)
    N, C, Hp, Wp = xp.shape
    K, _, R, S = w.shape
    _, _, Ho, Wo = gout.shape
    gxp = np.zeros_like(xp)
    gw = np.zeros_like(w)
    for r in range(R):
        for s in range(S):
            sl = (slice(None), slice(None), slice(r, r + stride * (Ho - 1) + 1, stride),
                  slice(s, s + stride * (Wo - 1) + 1, stride))
            gw[:, :, r, s] = np.einsum('nkhw,nchw->kc', gout, xp[sl])
            gxp[sl] += np.einsum('nkhw,kc->nchw', gout, w[:, :, r, s])
    gb = gout.sum(axis=(0, 2, 3))
    H, W = x.shape[2:]
    if pad == 0:
        gx = gxp
    elif pad_mode == 'reflect':
        gx = np.zeros_like(x)
        hh = reflect_index(np.arange(-pad, H + pad), H)
        ww = reflect_index(np.arange(-pad, W + pad), W)
        np.add.at(gx, (slice(None), slice(None), hh[:, None], ww[None, :]), gxp)
    else:
        gx = gxp[:, :, pad:pad + H, pad:pad + W]
    return gx, gw, gb


def conv_transpose2d_fwd(x, w, b=None, stride=2, pad=1, out_pad=1):
    """nn.ConvTranspose2d — reference models/networks.py:369-372 (k3 s2 p1 op1) and :522-538 (k4 s2 p1).
    x [N,Ci,H,W], w [Ci,Co,R,S] -> [N,Co,(H-1)s-2p+R+op, ...]."""
    N, Ci, H, W = x.shape
    _, Co, R, S = w.shape
    Ho = (H - 1) * stride - 2 * pad + R + out_pad
    Wo = (W - 1) * stride - 2 * pad + S + out_pad
    full = np.zeros((N, Co, (H - 1) * stride + R + out_pad, (W - 1) * stride + S + out_pad), dtype=x.dtype)
    for r in range(R):
        for s in range(S):
            full[:, :, r:r + stride * (H - 1) + 1:stride, s:s + stride * (W - 1) + 1:stride] += \
                np.einsum('nchw,ck->nkhw', x, w[:, :, r, s])
    out = full[:, :, pad:pad + Ho, pad:pad + Wo].copy()
    if b is not None:
        out += b[None, :, None, None]
    return out


def conv_transpose2d_bwd(x, w, gout, stride=2, pad=1, out_pad=1):
    """Returns (grad_x, grad_w, grad_b) of conv_transpose2d_fwd."""
    N, Ci, H, W = x.shape
    _, Co, R, S = w.shape
    gfull = np.zeros((N, Co, (H - 1) * stride + R + out_pad, (W - 1) * stride + S + out_pad), dtype=x.dtype)
    gfull[:, :, pad:pad + gout.shape[2], pad:pad + gout.shape[3]] = gout
    gx = np.zeros_like(x)
    gw = np.zeros_like(w)
    for r in range(R):
        for s in range(S):
            g = gfull[:, :, r:r + stride * (H - 1) + 1:stride, s:s + stride * (W - 1) + 1:stride]
            gx += np.einsum('nkhw,ck->nchw', g, w[:, :, r, s])
            gw[:, :, r, s] = np.einsum('nchw,nkhw->ck', x, g)
    return gx, gw, gout.sum(axis=(0, 2, 3))


# ------------------------------------------------------------------------------------------------
# K3 + K5: InstanceNorm and pointwise activations
# ------------------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3


def act_fwd(x, act, slope=0.2):
    """ReLU / LeakyReLU(0.2) / Tanh — reference models/networks.py:352,377,576; models/stn/layers.py:61-64."""
    if act == ACT_RELU:
        return np.maximum(x, 0)
    if act == ACT_LRELU:
        return np.where(x > 0, x, x * x.dtype.type(slope))
    if act == ACT_TANH:
        return np.tanh(x)
    return x


def act_bwd(gout, out, act, slope=0.2):
    """Gradient through the activation expressed with its OUTPUT (sign-preserving for relu/lrelu)."""
    if act == ACT_RELU:
        return gout * (out > 0)
    if act == ACT_LRELU:
        return np.where(out > 0, gout, gout * gout.dtype.type(slope))
    if act == ACT_TANH:
        return gout * (1 - out * out)
    return gout


def instance_norm_fwd(x, eps=1e-5):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False) — reference models/networks.py:24,
    models/stn/layers.py:16.  Biased variance over H*W per (n,c)."""
    m = x.mean(axis=(2, 3), keepdims=True)
    v = ((x - m) ** 2).mean(axis=(2, 3), keepdims=True)
    rstd = 1.0 / np.sqrt(v + x.dtype.type(eps))
    return (x - m) * rstd, m, rstd


def instance_norm_bwd(x, gy, eps=1e-5):
    xhat, m, rstd = instance_norm_fwd(x, eps)
    g1 = gy.mean(axis=(2, 3), keepdims=True)
    g2 = (gy * xhat).mean(axis=(2, 3), keepdims=True)
    return rstd * (gy - g1 - xhat * g2)


# ------------------------------------------------------------------------------------------------
# K6 + K7: pooling and bilinear resize
# ------------------------------------------------------------------------------------------------
def maxpool2_fwd(x):
    """nn.MaxPool2d(2) — reference models/stn/layers.py:174.  Returns (out, argmax in 0..3 row-major)."""
    N, C, H, W = x.shape
    Ho, Wo = H // 2, W // 2
    win = x[:, :, :Ho * 2, :Wo * 2].reshape(N, C, Ho, 2, Wo, 2).transpose(0, 1, 2, 4, 3, 5).reshape(N, C, Ho, Wo, 4)
    idx = win.argmax(axis=-1)
    return np.take_along_axis(win, idx[..., None], -1)[..., 0], idx


def maxpool2_bwd(x, gout):
    N, C, H, W = x.shape
    _, idx = maxpool2_fwd(x)
    Ho, Wo = H // 2, W // 2
    g = np.zeros((N, C, Ho, Wo, 4), dtype=x.dtype)
    np.put_along_axis(g, idx[..., None], gout[..., None], -1)
    gx = np.zeros_like(x)
    gx[:, :, :Ho * 2, :Wo * 2] = g.reshape(N, C, Ho, Wo, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(N, C, Ho * 2, Wo * 2)
    return gx


def _bilinear_taps(n_in, n_out, dtype):
    """align_corners=False source index: s = max((d+0.5)*in/out - 0.5, 0); i0=floor(s); i1=min(i0+1,in-1)."""
    scale = dtype(n_in) / dtype(n_out)
    s = np.maximum((np.arange(n_out, dtype=dtype) + dtype(0.5)) * scale - dtype(0.5), dtype(0))
    i0 = np.floor(s).astype(np.int64)
    i0 = np.minimum(i0, n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (s - i0.astype(dtype)).astype(dtype)
    return i0, i1, (dtype(1) - l1), l1


def bilinear_resize_fwd(x, Ho, Wo):
    """F.interpolate(x, (Ho,Wo), mode='bilinear', align_corners=False) — reference
    models/stn/unet_stn.py:96,140,166,188-195; models/nemar_model.py:187-188,204-205,226-227,240-241,254-255."""
    dt = x.dtype.type
    h0, h1, hl0, hl1 = _bilinear_taps(x.shape[2], Ho, dt)
    w0, w1, wl0, wl1 = _bilinear_taps(x.shape[3], Wo, dt)
    top = x[:, :, h0][:, :, :, w0] * wl0 + x[:, :, h0][:, :, :, w1] * wl1
    bot = x[:, :, h1][:, :, :, w0] * wl0 + x[:, :, h1][:, :, :, w1] * wl1
    return top * hl0[:, None] + bot * hl1[:, None]


def bilinear_resize_bwd(gout, H, W):
    dt = gout.dtype.type
    N, C, Ho, Wo = gout.shape
    h0, h1, hl0, hl1 = _bilinear_taps(H, Ho, dt)
    w0, w1, wl0, wl1 = _bilinear_taps(W, Wo, dt)
    gx = np.zeros((N, C, H, W), dtype=gout.dtype)
    for (hi, hl) in ((h0, hl0), (h1, hl1)):
        for (wi, wl) in ((w0, wl0), (w1, wl1)):
            np.add.at(gx, (slice(None), slice(None), hi[:, None], wi[None, :]), gout * hl[:, None] * wl[None, :])
    return gx


# ------------------------------------------------------------------------------------------------
# K13: losses
# ------------------------------------------------------------------------------------------------
def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def gan_loss_fwd(x, target_is_real, mode='vanilla'):
    """GANLoss.__call__ — reference models/networks.py:263-281 (BCEWithLogits / MSE / wgangp mean)
    against a constant target (1.0 real, 0.0 fake)."""
    if mode == 'vanilla':
        return np.mean(softplus(-x if target_is_real else x))
    if mode == 'lsgan':
        t = 1.0 if target_is_real else 0.0
        return np.mean((x - x.dtype.type(t)) ** 2)
    if mode == 'wgangp':
        return -np.mean(x) if target_is_real else np.mean(x)
    raise NotImplementedError(mode)


def gan_loss_bwd(x, target_is_real, mode='vanilla'):
    n = x.dtype.type(x.size)
    if mode == 'vanilla':
        sig = 1.0 / (1.0 + np.exp(-x))
        return (sig - (1.0 if target_is_real else 0.0)) / n
    if mode == 'lsgan':
        return 2 * (x - (1.0 if target_is_real else 0.0)) / n
    if mode == 'wgangp':
        return np.full_like(x, (-1.0 if target_is_real else 1.0)) / n
    raise NotImplementedError(mode)


def l1_loss_fwd(a, b):
    """torch.nn.L1Loss() — reference models/nemar_model.py:68,179,195."""
    return np.mean(np.abs(a - b))


def l1_loss_bwd(a, b):
    return np.sign(a - b) / a.dtype.type(a.size)


# ------------------------------------------------------------------------------------------------
# K15: Adam
# ------------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8):
    """torch.optim.Adam(lr, betas=(beta1, 0.999)) single-tensor update — reference
    models/nemar_model.py:128-137 (construction), :274,282-283 (step).  `step` is 1-based."""
    dt = p.dtype.type
    m = dt(beta1) * m + dt(1 - beta1) * g
    v = dt(beta2) * v + dt(1 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / dt(np.sqrt(bc2)) + dt(eps)
    p = p - dt(lr / bc1) * (m / denom)
    return p, m, v
