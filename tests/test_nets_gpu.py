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


@pytest.mark.parametrize("drop", [False, True])
def test_resnet_block_one_node_vs_composed(drop):
    """Round 6: two ResnetBlocks of the wide route (256 channels, 64 x 64) as ONE autograd node each (ops._ResBlock: producer-written
    operand planes for all six convolution calls, no fp32 gradient tensors inside the block, skip gradient added in a data gradient's
    epilogue) against the same blocks composed from conv2d / instance_norm nodes: the forward pass is the same kernels on the same
    bits; the gradients differ by the planes' scale (an a-priori bound instead of the measured maximum) — both are held to the
    float64 gradient of the block (torch CPU, reference models/networks.py:418-446) with the tolerance of the wide-layer kernels."""
    import torch.nn.functional as F
    from nemar_amd import ops
    from nemar_amd.models import networks
    dev = torch.device('cuda:0')
    N, C, H, W = 6, 256, 64, 64          # (below 5 samples the data gradient splits its reduction over workgroups: no fused epilogue, composed blocks)
    blocks = [networks.ResnetBlock(C, 'reflect', 'instance', drop, True).to(dev) for _ in range(2)]
    blocks[0].feeds_block = True
    g = torch.Generator().manual_seed(7)
    for b in blocks:
        for p in b.parameters():
            p.data.copy_((torch.randn(p.shape, generator=g) * (0.03 if p.dim() == 4 else 0.1)).to(dev))
    ops.invalidate_packed_weights()
    params = [p for b in blocks for p in b.parameters()]
    opt = ops.FlatAdam(params)
    x0 = (torch.randn((N, C, H, W), generator=g) * torch.tensor([1.0, 0.05, 7.0, 1.0, 1.0, 2.0]).view(N, 1, 1, 1))
    gout = torch.randn((N, C, H, W), generator=g) * torch.tensor([1.0, 3.0, 1e-3, 1.0, 0.3, 1.0]).view(N, 1, 1, 1)

    def run(fused):
        prev = ops.fused_blocks(fused)
        try:
            ops.manual_seed(1234)
            opt.zero_grad()
            x = x0.to(dev).requires_grad_(True)
            # a producer in front, as the down-sampling stage's InstanceNorm in the generator: writes the first block's operand planes
            h = ops.instance_norm(x, act=ops.ACT_RELU, planes=True)
            for b in blocks:
                h = b(h)
            torch.autograd.backward([h], [gout.to(dev)])
            ops.join_side()
            torch.cuda.synchronize()
            return h.detach().cpu(), x.grad.detach().cpu(), [p.grad.detach().cpu().clone() for p in params]
        finally:
            ops.fused_blocks(prev)

    out_c, gx_c, gp_c = run(False)
    out_f, gx_f, gp_f = run(True)
    out_f2, gx_f2, gp_f2 = run(True)
    assert not torch.equal(gx_f, gx_c), "the one-node block did not run (its gradients are bit-identical to the composed block's)"
    assert torch.equal(out_f, out_c), "the forward pass of the one-node block must be bit-identical to the composed block"
    assert torch.equal(gx_f, gx_f2) and all(torch.equal(a, b) for a, b in zip(gp_f, gp_f2)), "not reproducible run to run"
    # float64 truth (dropout masks taken from the build's own forward: zero pattern of the block's intermediate is not observable, so
    # the reference is only drawn for the mask-free configuration; with dropout the two builds are compared with each other)
    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())
    if not drop:
        xr = x0.double().requires_grad_(True)
        pr = [p.detach().cpu().double().requires_grad_(True) for p in params]
        h = F.relu(F.instance_norm(xr))
        for k in range(2):
            w1, b1, w2, b2 = pr[4 * k:4 * k + 4]
            t = F.instance_norm(F.conv2d(F.pad(h, (1, 1, 1, 1), mode='reflect'), w1, b1))
            t = F.instance_norm(F.conv2d(F.pad(F.relu(t), (1, 1, 1, 1), mode='reflect'), w2, b2))
            h = h + t
        h.backward(gout.double())
        assert rel(out_f.double(), h.detach()) < 2e-5
        errs = {'gx': (rel(gx_f.double(), xr.grad), rel(gx_c.double(), xr.grad))}
        for k, (a, c, r) in enumerate(zip(gp_f, gp_c, pr)):
            if r.dim() == 4:        # (conv bias in front of InstanceNorm: an exactly-null gradient, rounding noise on every side)
                errs['w%d' % k] = (rel(a.double(), r.grad), rel(c.double(), r.grad))
        print('one-node / composed error against float64:', {k: ('%.2e' % a, '%.2e' % c) for k, (a, c) in errs.items()})
        # the one-node block must be as accurate as the composed one (same kernels; planes scaled by a bound instead of the maximum)
        for k, (a, c) in errs.items():
            assert a < 2e-2 and c < 2e-2 and a < 2.0 * c + 1e-5, (k, errs)
    assert rel(gx_f, gx_c) < 1e-4, rel(gx_f, gx_c)
    for k, (a, c) in enumerate(zip(gp_f, gp_c)):
        if a.dim() == 4:
            assert rel(a, c) < 1e-4, (k, rel(a, c))
        else:
            # bias gradients in front of InstanceNorm are sums that cancel to rounding noise: both must be negligible against the scale
            # of the weight gradient of the same layer
            assert float(a.abs().max()) < 1e-3 * float(gp_f[k - 1].abs().max()) * H * W


def test_own_autograd_nodes_match_autograd():
    """ops.cat_batch / split_batch / fork (nemar_concat_pieces, nemar_add2) against torch.cat, slices and autograd's own accumulation on the
    same graph: values and gradients bit for bit (copies, and sums of two terms), a slice nothing reads gets a zero gradient, and no
    gradient at all is materialised for an input that does not require one."""
    import torch
    from nemar_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(5)

    def graph(cat, split, fork):
        a = torch.randn(2, 3, 8, 12, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).requires_grad_()
        b = torch.randn(2, 3, 8, 12, device=dev, generator=torch.Generator(device=dev).manual_seed(2)).requires_grad_()
        c = torch.randn(2, 3, 8, 12, device=dev, generator=torch.Generator(device=dev).manual_seed(3))          # no gradient
        w = torch.randn(6, 3, 8, 12, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
        both = cat([a * 2.0, b * 3.0, c]) * w
        p0, p1, p2 = split(both, 3)
        f0, f1 = fork(p0)
        # (two consumers of p0, one gradient term each: a sum of TWO fp32 terms is the same in either order; p2 is read by nothing)
        loss = (f0 * w[:2]).sum() + f1.sum() * 0.5 + (p1 * 0.25).sum()
        loss.backward()
        return both.detach(), a.grad, b.grad

    own = graph(ops.cat_batch, ops.split_batch, lambda x: ops.fork(x, 2))
    ref = graph(lambda ts: torch.cat(ts, 0), lambda x, k: tuple(x[i * 2:(i + 1) * 2] for i in range(k)), lambda x: (x, x))
    for what, x, y in zip(('values', 'grad a', 'grad b'), own, ref):
        assert torch.equal(x, y), (what, float((x - y).abs().max()), int((x != y).sum()), x.numel())
    assert own[1].abs().max() > 0 and own[2].abs().max() > 0


@pytest.mark.parametrize("size", [4, 16, 64])
def test_skip_gradients_riding_in_another_pass(size):
    """ops.conv2d_with_skip (the ResnetBlock's skip gradient as the addend of the first convolution's data gradient: the sum-and-fold pass
    of the tiny maps at 4 x 4 / 16 x 16, the fold pass of the padded-domain data gradient at 64 x 64) and ops.max_pool2_with_skip (the U-Net
    skip's gradient as the addend of the pooling's backward kernel) against the same graph with autograd's own accumulation: bit for bit
    — the sum of the same two fp32 terms."""
    import torch
    from nemar_amd import ops
    dev = torch.device('cuda:0')
    C = 32 if size == 64 else 64
    gen = lambda s: torch.Generator(device=dev).manual_seed(s)
    w0 = (torch.randn(C, C, 3, 3, device=dev, generator=gen(1)) * 0.05)
    b0 = torch.randn(C, device=dev, generator=gen(2)) * 0.1
    x0 = torch.randn(8, C, size, size, device=dev, generator=gen(3))
    gy = torch.randn(8, C, size, size, device=dev, generator=gen(4))
    gs = torch.randn(8, C, size, size, device=dev, generator=gen(5))
    gp = torch.randn(8, C, size // 2, size // 2, device=dev, generator=gen(6))

    def run(own):
        x = x0.clone().requires_grad_()
        w = torch.nn.Parameter(w0.clone())
        b = torch.nn.Parameter(b0.clone())
        h = x * 1.5                                        # a producer in front: the summed gradient is what it receives
        if own:
            y, skip = ops.conv2d_with_skip(h, w, b, 1, 1, ops.PAD_REFLECT)
        else:
            y, skip = ops.conv2d(h, w, b, 1, 1, ops.PAD_REFLECT), h
        out = y + skip                                     # (the residual sum itself is not under test)
        if own:
            pooled, skip2 = ops.max_pool2_with_skip(out)
        else:
            pooled, skip2 = ops.max_pool2(out), out
        torch.autograd.backward([pooled, skip2], [gp, gs + gy])
        ops.join_side()
        return x.grad.clone(), w.grad.clone(), b.grad.clone()

    a, r = run(True), run(False)
    assert torch.equal(a[0], r[0]), float((a[0] - r[0]).abs().max())
    assert torch.equal(a[1], r[1]) and torch.equal(a[2], r[2])
