"""`-m gpu`: each network of the hot path (translation generator, PatchGAN discriminator, UNet STN, affine STN) on
the MI355X kernels vs the CPU oracle ON IDENTICAL INPUTS with a smooth (linear) objective: outputs, gradients of
every parameter tensor and of the inputs.  This is the tight gradient-parity gate for the backward kernels in
their real network context; tests/test_step_gpu.py then checks the assembled training step."""
import numpy as np
import pytest
import torch

import seeded
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu


def _load(net, seed, ov=None):
    sd = net.state_dict()
    new = seeded.seeded_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed, ov)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})
    return {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in new.items()}


def _check_grads(net, P, tol):
    gmax = max(float(p.grad.abs().max()) for p in P.values() if p.grad is not None)
    for k, p in net.named_parameters():
        r = P[k].grad
        if r is None or float(r.abs().max()) < 1e-5 * gmax:
            continue        # exactly-null gradients (conv bias in front of InstanceNorm) are rounding noise on both sides
        e = float((p.grad.detach().cpu() - r).abs().max()) / float(r.abs().max())
        assert e < tol, (k, e)


def _opt(stn_type, size):
    from step_configs import make_opt
    return make_opt(dict(stn_type=stn_type, netG='resnet_3blocks', ngf=8, ndf=8, size=size, batch=2, seed=1), [0])


def test_translation_generator():
    from nemar_amd import ops
    from nemar_amd.models import networks
    dev = torch.device('cuda:0')
    net = networks.define_G(3, 3, 8, 'resnet_3blocks', 'instance', False, 'normal', 0.02, [0])
    P = _load(net, 1)
    ops.FlatAdam(net.parameters())
    A, _ = seeded.seeded_images(2, 3, 64, 64, 5)
    x = torch.from_numpy(A).to(dev).requires_grad_(True)
    xr = torch.from_numpy(A).clone().requires_grad_(True)
    out, ref = net(x), R.resnet_generator(P, xr, 3)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 2e-5
    w = torch.from_numpy(seeded.uniform(tuple(out.shape), 78))
    torch.autograd.backward([out], [w.to(dev)])
    (ref * w).sum().backward()
    _check_grads(net, P, 1e-3)
    assert float((x.grad.cpu() - xr.grad).abs().max() / xr.grad.abs().max()) < 2e-3


def test_discriminator_three_passes():
    from nemar_amd import ops
    from nemar_amd.models import networks
    dev = torch.device('cuda:0')
    net = networks.define_D(6, 8, 'basic', 3, 'instance', 'normal', 0.02, [0])
    P = _load(net, 3)
    ops.FlatAdam(net.parameters())
    A, B = seeded.seeded_images(2, 3, 128, 128, 5)
    f1, f2 = seeded.seeded_images(2, 3, 128, 128, 9)
    tA, tB, t1, t2 = [torch.from_numpy(v) for v in (A, B, f1, f2)]
    crit = networks.GANLoss('vanilla')
    one = torch.ones((), device=dev)
    x2 = t1.to(dev).requires_grad_(True)
    imgs = [(tB.to(dev), tB, True), (x2, t1, False), (t2.to(dev), t2, False)]
    terms = [crit(net(tA.to(dev), d), real, 0.5) for d, _, real in imgs]
    torch.autograd.backward(terms, [one] * 3)
    t1r = t1.clone().requires_grad_(True)
    loss = sum(0.5 * R.gan_loss(R.nlayer_discriminator(P, torch.cat([tA, t], 1)), real)
               for t, real in ((tB, True), (t1r, False), (t2, False)))
    loss.backward()
    assert abs(sum(float(t) for t in terms) - float(loss)) < 1e-5
    _check_grads(net, P, 1e-4)
    assert float((x2.grad.cpu() - t1r.grad).abs().max() / t1r.grad.abs().max()) < 1e-3   # dgrad into the image half


@pytest.mark.parametrize("stn_type,size", [("unet", 256), ("affine", 128)])
def test_registration_network(stn_type, size):
    from nemar_amd import ops
    from nemar_amd.models import stn
    dev = torch.device('cuda:0')
    opt = _opt(stn_type, size)
    opt.stn_bilateral_alpha, opt.stn_multires_reg = 1.5, 2
    net = stn.define_stn(opt, stn_type)
    ov = {'offset_map.output.conv2d.weight': 0.02} if stn_type == 'unet' else {'net.local.2.weight': 0.02,
                                                                                  'net.local.2.bias': 0.05}
    P = _load(net, 2, ov)
    ops.FlatAdam(net.parameters())
    A, B = seeded.seeded_images(1, 3, size, size, 5)
    F_, _ = seeded.seeded_images(1, 3, size, size, 6)
    tA, tB, tF = [torch.from_numpy(v) for v in (A, B, F_)]
    xf = tF.to(dev).requires_grad_(True)
    xfr = tF.clone().requires_grad_(True)
    warped, reg = net(tA.to(dev), tB.to(dev), apply_on=[tA.to(dev), xf])
    if stn_type == 'unet':
        wr, regr, off = R.unet_stn(P, tA, tB, [tA, xfr], 1.5, 2)
    else:
        wr, regr, off = R.affine_stn(P, tA, tB, [tA, xfr])
    for a, b in zip(warped, wr):
        assert float((a.detach().cpu() - b.detach()).abs().max()) < 5e-4
    assert abs(float(reg) - float(regr)) < 1e-5 * max(1.0, abs(float(regr)))
    w0 = torch.from_numpy(seeded.uniform(tuple(wr[0].shape), 70))
    w1 = torch.from_numpy(seeded.uniform(tuple(wr[1].shape), 71))
    three = torch.full((), 3.0, device=dev)
    torch.autograd.backward([warped[0], warped[1], reg], [w0.to(dev), w1.to(dev), three])
    ((wr[0] * w0).sum() + (wr[1] * w1).sum() + 3.0 * regr).backward()
    # d(warp)/d(offset) of an image carrying 10% white noise is discontinuous at texel boundaries: percent-level
    # per-tensor deviations are the fp32 conditioning of the reference's own algorithm (see tests/step_parity.py)
    _check_grads(net, P, 3e-2)
    assert float((xf.grad.cpu() - xfr.grad).abs().max() / xfr.grad.abs().max()) < 1e-3   # grid_sample grad_input


def test_unet_generator_matches_reference_fixture():
    """`--netG unet_128` (reference models/networks.py:449-553) on the MI355X kernels against outputs and gradients recorded
    from the reference itself (tests/golden/unet_generator.npz, seeded weights / inputs)."""
    import os
    import numpy as np
    import seeded
    from nemar_amd.models import networks
    from step_parity import load_seeded_into
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'unet_generator.npz'))
    net = networks.define_G(3, 3, 4, 'unet_128', 'instance', False, 'normal', 0.02, [0])
    load_seeded_into(net, 77, None)
    x = torch.from_numpy(seeded.seeded_images(2, 3, 128, 128, 78)[0]).cuda().requires_grad_(True)
    r = torch.from_numpy(seeded.seeded_images(2, 3, 128, 128, 79)[1]).cuda()
    params = dict(net.named_parameters())
    for p in params.values():
        p.grad = torch.zeros_like(p)          # weight-gradient kernels accumulate
    y = net(x)
    (y * r).sum().backward()
    assert (y.detach().cpu().numpy() - g['plain/y']).__abs__().max() < 2e-5
    gx = x.grad.cpu().numpy()
    assert np.abs(gx - g['plain/gx']).max() <= 2e-4 * np.abs(g['plain/gx']).max()
    gmax = max(float(g['plain/gnorm/' + k]) for k in params)
    for k, p in params.items():
        want = float(g['plain/gnorm/' + k])
        got = float(p.grad.norm())
        if want < 1e-4 * gmax:                 # conv biases in front of InstanceNorm: exactly-zero gradient + rounding noise
            assert got < 1e-3 * gmax, (k, got, want, gmax)
            continue
        assert abs(got - want) <= 2e-4 * want, (k, got, want)
        head = p.grad.reshape(-1)[:64].cpu().numpy()
        assert np.abs(head - g['plain/ghead/' + k]).max() <= 5e-4 * max(np.abs(g['plain/ghead/' + k]).max(), want / 10), k
