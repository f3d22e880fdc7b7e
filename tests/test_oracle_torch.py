"""Pin the numpy oracle (oracle/ops_np.py) against torch CPU float64 autograd, op by op.
torch is the third-party library whose ATen operators produce every number on the reference's hot path
(SURVEY.md §8c); the oracle restates those operators' published algorithms."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops_np as O


def T(a):
    return torch.tensor(a, dtype=torch.float64, requires_grad=True)


def close(a, b, tol=1e-11):
    b = b.detach().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(np.asarray(a), b, rtol=0, atol=tol)


@pytest.fixture
def rng():
    return np.random.default_rng(0)


def test_linspace_bit_exact_fp32():
    for n in (2, 5, 8, 255, 256, 257, 1024):
        assert np.array_equal(O.linspace_m1_p1(n, np.float32), torch.linspace(-1, 1, n).numpy())


@pytest.mark.parametrize("scale", [0.0, 0.3, 3.0])
def test_grid_sample_unet(rng, scale):
    N, C, H, W = 2, 3, 9, 11
    inp = rng.standard_normal((N, C, H, W))
    off = rng.standard_normal((N, 2, H, W)) * scale
    grid = O.unet_grid(off)
    ti, tg = T(inp), T(grid)
    out = F.grid_sample(ti, tg, mode='bilinear', padding_mode='zeros', align_corners=False)
    close(O.grid_sample_fwd(inp, grid), out)
    go = rng.standard_normal(out.shape)
    out.backward(torch.tensor(go))
    gin, gg = O.grid_sample_bwd(inp, grid, go)
    close(gin, ti.grad)
    close(gg, tg.grad)


def test_affine_grid_and_warp(rng):
    N, C, H, W = 2, 3, 9, 11
    inp = rng.standard_normal((N, C, H, W))
    dth = rng.standard_normal((N, 6)) * 0.1
    tt = T(dth)
    ident = torch.tensor([1, 0, 0, 0, 1, 0], dtype=torch.float64)
    g = F.affine_grid((tt + ident).view(-1, 2, 3), (N, C, H, W), align_corners=False)
    close(O.affine_grid(O.affine_theta(dth), H, W), g)
    ti = T(inp)
    o = F.grid_sample(ti, g, align_corners=False)
    go = rng.standard_normal(o.shape)
    o.backward(torch.tensor(go))
    gin, gth = O.affine_warp_bwd(inp, dth, go, H, W)
    close(gin, ti.grad)
    close(gth, tt.grad, 1e-10)


@pytest.mark.parametrize("k,st,pd,pm", [(3, 1, 1, 'zeros'), (3, 2, 1, 'zeros'), (3, 1, 1, 'reflect'),
                                         (4, 2, 1, 'zeros'), (4, 1, 1, 'zeros'), (7, 1, 3, 'reflect'),
                                         (1, 1, 0, 'zeros')])
def test_conv2d(rng, k, st, pd, pm):
    x = rng.standard_normal((2, 5, 8, 9))
    w = rng.standard_normal((4, 5, k, k))
    b = rng.standard_normal(4)
    tx, tw, tb = T(x), T(w), T(b)
    xin = F.pad(tx, (pd, pd, pd, pd), mode='reflect') if pm == 'reflect' else tx
    o = F.conv2d(xin, tw, tb, stride=st, padding=0 if pm == 'reflect' else pd)
    close(O.conv2d_fwd(x, w, b, st, pd, pm), o)
    go = rng.standard_normal(o.shape)
    o.backward(torch.tensor(go))
    gx, gw, gb = O.conv2d_bwd(x, w, go, st, pd, pm)
    close(gx, tx.grad)
    close(gw, tw.grad, 1e-10)
    close(gb, tb.grad, 1e-10)


@pytest.mark.parametrize("k,op", [(3, 1), (4, 0)])
def test_conv_transpose2d(rng, k, op):
    x = rng.standard_normal((2, 5, 8, 9))
    w = rng.standard_normal((5, 4, k, k))
    b = rng.standard_normal(4)
    tx, tw, tb = T(x), T(w), T(b)
    o = F.conv_transpose2d(tx, tw, tb, stride=2, padding=1, output_padding=op)
    close(O.conv_transpose2d_fwd(x, w, b, 2, 1, op), o)
    go = rng.standard_normal(o.shape)
    o.backward(torch.tensor(go))
    gx, gw, gb = O.conv_transpose2d_bwd(x, w, go, 2, 1, op)
    close(gx, tx.grad)
    close(gw, tw.grad, 1e-10)
    close(gb, tb.grad, 1e-10)


def test_instance_norm(rng):
    x = rng.standard_normal((2, 5, 8, 9)) * 3 + 1
    tx = T(x)
    o = F.instance_norm(tx, eps=1e-5)
    close(O.instance_norm_fwd(x)[0], o)
    go = rng.standard_normal(o.shape)
    o.backward(torch.tensor(go))
    close(O.instance_norm_bwd(x, go), tx.grad, 1e-10)


def test_activations(rng):
    x = rng.standard_normal((2, 3, 4, 5))
    go = rng.standard_normal(x.shape)
    for act, fn in ((O.ACT_RELU, F.relu), (O.ACT_LRELU, lambda t: F.leaky_relu(t, 0.2)), (O.ACT_TANH, torch.tanh)):
        tx = T(x)
        o = fn(tx)
        close(O.act_fwd(x, act), o)
        o.backward(torch.tensor(go))
        close(O.act_bwd(go, O.act_fwd(x, act), act), tx.grad)


def test_maxpool(rng):
    x = rng.standard_normal((2, 3, 8, 10))
    tx = T(x)
    o = F.max_pool2d(tx, 2)
    close(O.maxpool2_fwd(x)[0], o)
    go = rng.standard_normal(o.shape)
    o.backward(torch.tensor(go))
    close(O.maxpool2_bwd(x, go), tx.grad)


@pytest.mark.parametrize("Ho,Wo", [(16, 20), (4, 5), (2, 2), (13, 7)])
def test_bilinear(rng, Ho, Wo):
    x = rng.standard_normal((2, 3, 8, 10))
    tx = T(x)
    o = F.interpolate(tx, (Ho, Wo), mode='bilinear', align_corners=False)
    close(O.bilinear_resize_fwd(x, Ho, Wo), o)
    go = rng.standard_normal(o.shape)
    o.backward(torch.tensor(go))
    close(O.bilinear_resize_bwd(go, 8, 10), tx.grad)


def test_l1_and_gan_losses(rng):
    a = rng.standard_normal((2, 3, 4, 4))
    b = rng.standard_normal((2, 3, 4, 4))
    ta = T(a)
    l = torch.nn.L1Loss()(ta, torch.tensor(b))
    l.backward()
    close(O.l1_loss_fwd(a, b), l)
    close(O.l1_loss_bwd(a, b), ta.grad)
    lg = rng.standard_normal((2, 1, 6, 6)) * 3
    for real in (True, False):
        tl = T(lg)
        tgt = torch.full_like(tl, 1.0 if real else 0.0)
        l = F.binary_cross_entropy_with_logits(tl, tgt)
        l.backward()
        close(O.gan_loss_fwd(lg, real, 'vanilla'), l)
        close(O.gan_loss_bwd(lg, real, 'vanilla'), tl.grad)
        tl = T(lg)
        l = F.mse_loss(tl, tgt.detach())
        l.backward()
        close(O.gan_loss_fwd(lg, real, 'lsgan'), l)
        close(O.gan_loss_bwd(lg, real, 'lsgan'), tl.grad)


def test_adam(rng):
    p = rng.standard_normal(50)
    tp = torch.nn.Parameter(torch.tensor(p))
    opt = torch.optim.Adam([tp], lr=2e-4, betas=(0.5, 0.999))
    m = np.zeros(50)
    v = np.zeros(50)
    pp = p.copy()
    for step in range(1, 5):
        g = rng.standard_normal(50)
        tp.grad = torch.tensor(g)
        opt.step()
        pp, m, v = O.adam_step(pp, g, m, v, step)
    close(pp, tp, 1e-13)
