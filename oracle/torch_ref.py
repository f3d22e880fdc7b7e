"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Step-level CPU restatement of the NeMAR hot path, `NEMARModel.optimize_parameters()` (reference
models/nemar_model.py:266-288), written from scratch in a functional style on plain torch CPU fp32 ops — the same
third-party ATen operators the reference itself bottoms out in (torch 2.10.0, not vendored in
/root/reference and not pinned by it).  Parameters are plain tensors in dicts keyed by the reference's state_dict
names, so the same seeded weights load into the reference, into this oracle and into the
MI355X build.

Pinned by tests/golden/make_golden.py (imports the reference in the build container, stores its outputs as
fixtures) + tests/test_oracle_golden.py (this module must reproduce those fixtures).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; it is also the
"port" timed as the CPU baseline.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

UNET_NDF = {'A': [32, 64, 64, 64, 64, 64, 64], 'deep': [32, 64, 64, 64, 64, 64, 64, 64, 64]}
UNET_NUF = {'A': [64, 64, 64, 64, 64, 64, 32], 'deep': [64, 64, 64, 64, 64, 64, 64, 64, 32]}
UNET_NRES = {'A': 3, 'deep': 3}


# ---- building blocks -------------------------------------------------------------------------------------
def _conv(P, key, x, stride=1, pad=0, reflect=False):
    """nn.Conv2d (optionally behind nn.ReflectionPad2d(pad)) — reference models/networks.py:349-350,418-439."""
    if reflect and pad:
        x = F.pad(x, (pad, pad, pad, pad), mode='reflect')
        pad = 0
    return F.conv2d(x, P[key + '.weight'], P.get(key + '.bias'), stride=stride, padding=pad)


def _in(x):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False) — reference models/networks.py:24."""
    return F.instance_norm(x, eps=1e-5)


def _resblock(P, key, x, second, dropout_masks=None):
    """ResnetBlock (reflect, IN, ReLU[, Dropout]) — reference models/networks.py:389-446.  `second` is the
    nn.Sequential slot of the second conv (5 without dropout, 6 with)."""
    h = F.relu(_in(_conv(P, '%s.conv_block.1' % key, x, 1, 1, True)))
    if dropout_masks is not None:
        h = h * dropout_masks.pop(0)
    h = _in(_conv(P, '%s.conv_block.%d' % (key, second), h, 1, 1, True))
    return x + h


def resnet_generator(P, x, n_blocks, use_dropout=False, dropout_masks=None):
    """ResnetGenerator.forward — reference models/networks.py:323-386.  Dropout is only supported through
    explicit masks (parity runs use --no_dropout)."""
    second = 6 if use_dropout else 5
    h = F.relu(_in(_conv(P, 'model.1', x, 1, 3, True)))
    h = F.relu(_in(_conv(P, 'model.4', h, 2, 1)))
    h = F.relu(_in(_conv(P, 'model.7', h, 2, 1)))
    for i in range(n_blocks):
        h = _resblock(P, 'model.%d' % (10 + i), h, second, dropout_masks)
    for idx in (10 + n_blocks, 13 + n_blocks):
        h = F.conv_transpose2d(h, P['model.%d.weight' % idx], P.get('model.%d.bias' % idx), stride=2, padding=1,
                               output_padding=1)
        h = F.relu(_in(h))
    return torch.tanh(_conv(P, 'model.%d' % (17 + n_blocks), h, 1, 3, True))


def _lrelu(x, knife=None):
    """LeakyReLU(0.2).  `knife` (tests only, see RefModel.knife_band) = {'band': t, 'count': n}: the elements whose
    pre-activation lies within t x mean|x| of zero take the OTHER branch — the evaluation another correctly rounded fp32
    implementation could have produced; `count` accumulates how many there were."""
    if knife is None:
        return F.leaky_relu(x, 0.2)
    near = x.detach().abs() < knife['band'] * x.detach().abs().mean()
    knife['count'] += int(near.sum())
    return torch.where((x > 0) ^ near, x, 0.2 * x)


def nlayer_discriminator(P, x, knife=None):
    """NLayerDiscriminator('basic').forward — reference models/networks.py:556-602."""
    h = _lrelu(_conv(P, 'model.0', x, 2, 1), knife)
    h = _lrelu(_in(_conv(P, 'model.2', h, 2, 1)), knife)
    h = _lrelu(_in(_conv(P, 'model.5', h, 2, 1)), knife)
    h = _lrelu(_in(_conv(P, 'model.8', h, 1, 1)), knife)
    return _conv(P, 'model.11', h, 1, 1)


def _stn_resblock(P, key, x):
    return _resblock(P, key, x, 5)


def res_unet(P, a, b, cfg='A'):
    """ResUnet.forward — reference models/stn/unet_stn.py:79-102 (keys carry the `offset_map.` prefix)."""
    x = torch.cat([a, b], 1)
    skips = {}
    nd = len(UNET_NDF[cfg])
    for i in range(1, nd + 1):
        k = 'offset_map.down_%d.conv_0' % i
        x = F.leaky_relu(_conv(P, k + '.conv2d', x, 1, 1), 0.2)
        x = _stn_resblock(P, k + '.resnet_block.model.0', x)
        skips[i] = x
        x = F.max_pool2d(x, 2)
    x = F.leaky_relu(_conv(P, 'offset_map.c1.conv2d', x), 0.2)
    for j in range(UNET_NRES[cfg]):
        x = _stn_resblock(P, 'offset_map.t.model.%d' % j, x)
    x = F.leaky_relu(_conv(P, 'offset_map.c2.conv2d', x), 0.2)
    level = nd
    for _ in UNET_NUF[cfg]:
        s = skips[level]
        x = F.interpolate(x, (s.size(2), s.size(3)), mode='bilinear', align_corners=False)
        x = F.leaky_relu(_conv(P, 'offset_map.up_%d.conv2d' % level, torch.cat([x, s], 1), 1, 1), 0.2)
        level -= 1
    x = _stn_resblock(P, 'offset_map.refine.0.model.0', x)
    x = F.leaky_relu(_conv(P, 'offset_map.refine.1.conv2d', x), 0.2)
    return _conv(P, 'offset_map.output.conv2d', x, 1, 1)


def smoothness_loss(d, img=None, alpha=0.0):
    """reference models/stn/stn_losses.py:4-30."""
    def pairs(t):
        return ((t[:, :, 1:, :], t[:, :, :-1, :]), (t[:, :, :, 1:], t[:, :, :, :-1]),
                (t[:, :, :-1, :-1], t[:, :, 1:, 1:]), (t[:, :, :-1, 1:], t[:, :, 1:, :-1]))
    loss = 0.0
    ip = pairs(img) if (img is not None and alpha > 0.0) else [None] * 4
    for (p, q), w in zip(pairs(d), ip):
        diff = (p - q).abs()
        if w is not None:
            diff = torch.exp(-alpha * (w[0] - w[1]).abs()).mean(dim=1, keepdim=True) * diff
        loss = loss + diff.mean()
    return loss


def unet_stn(P, a, b, apply_on, alpha=0.0, multires=1, cfg='A'):
    """UnetSTN.forward — reference models/stn/unet_stn.py:148-201.  Returns (warped list, reg, offsets)."""
    d = res_unet(P, a, b, cfg)
    N, _, H, W = d.shape
    xs = torch.linspace(-1.0, 1.0, W, dtype=d.dtype)
    ys = torch.linspace(-1.0, 1.0, H, dtype=d.dtype)
    ident = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W)], 0)[None]
    grid = (ident + d).permute(0, 2, 3, 1)
    warped = [F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False) for img in apply_on]
    img0 = warped[0].detach()
    reg, factor = 0.0, 1.0
    for i in range(multires):
        if i == 0:
            dr, ir = d, img0
        else:
            size = (H // 2 ** i, W // 2 ** i)
            dr = F.interpolate(d, size, mode='bilinear', align_corners=False)
            ir = F.interpolate(img0, size, mode='bilinear', align_corners=False)
        reg = reg + factor * smoothness_loss(dr, ir, alpha)
        factor /= 2.0
    return warped, reg, d


def affine_stn(P, a, b, apply_on):
    """AffineSTN.forward — reference models/stn/affine_stn.py:108-138.  Returns (warped list, reg, dtheta)."""
    x = torch.cat([a, b], 1)
    for i in range(5):
        x = F.max_pool2d(F.relu(_in(_conv(P, 'net.convs.%d.conv_0.conv2d' % i, x, 1, 1))), 2)
    x = x.reshape(x.size(0), -1)
    x = F.relu(F.linear(x, P['net.local.0.weight'], P['net.local.0.bias']))
    dtheta = F.linear(x, P['net.local.2.weight'], P['net.local.2.bias'])
    theta = dtheta + torch.tensor([1.0, 0, 0, 0, 1, 0], dtype=dtheta.dtype)[None]
    warped = []
    for img in apply_on:
        grid = F.affine_grid(theta.view(-1, 2, 3), img.size(), align_corners=False)
        warped.append(F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False))
    return warped, dtheta.abs().mean(), dtheta


def gan_loss(pred, real, mode='vanilla'):
    """GANLoss.__call__ — reference models/networks.py:263-281."""
    if mode == 'vanilla':
        return F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, 1.0 if real else 0.0))
    if mode == 'lsgan':
        return F.mse_loss(pred, torch.full_like(pred, 1.0 if real else 0.0))
    if mode == 'wgangp':
        return -pred.mean() if real else pred.mean()
    raise NotImplementedError(mode)


# ---- the training step ----------------------------------------------------------------------------------------
class RefModel:
    """Holds parameter dicts T, R, D (+ reduced-resolution Ds) and three torch.optim.Adam optimizers, and performs
    the reference's optimize_parameters() with them."""

    def __init__(self, sd_T, sd_R, sd_D, sd_D_mr=(), *, n_blocks, stn_type='unet', gan_mode='vanilla', lr=2e-4,
                 beta1=0.5, lambda_GAN=1.0, lambda_recon=100.0, lambda_smooth=0.0, alpha=0.0, multires_reg=1,
                 stn_cfg='A', dtype=torch.float32):
        """dtype=torch.float64 gives the 'true value' run used to calibrate fp32 tolerances (conditioning)."""
        self.dtype = dtype
        leaf = lambda sd: OrderedDict((k, v.detach().clone().to(dtype).requires_grad_(True)) for k, v in sd.items())
        self.T, self.R, self.D = leaf(sd_T), leaf(sd_R), leaf(sd_D)
        self.D_mr = [leaf(s) for s in sd_D_mr]
        self.cfg = dict(n_blocks=n_blocks, stn_type=stn_type, gan_mode=gan_mode, lambda_GAN=lambda_GAN,
                        lambda_recon=lambda_recon, lambda_smooth=lambda_smooth, alpha=alpha, multires_reg=multires_reg,
                        stn_cfg=stn_cfg)
        # Knife-edge calibration of the D step (tests/step_parity.py): when set, optimize_parameters() also evaluates D's gradients
        # with every LeakyReLU pre-activation within knife_band x mean|x| of zero on its other branch -> grads_D_knife /
        # grads_D_mr_knife / knife_count.  One such element moves a whole layer's weight gradient by ~1 % whichever way an fp32
        # implementation happened to round it.
        self.knife_band = None
        mk = lambda ps: torch.optim.Adam(ps, lr=lr, betas=(beta1, 0.999))
        self.opt_T = mk(list(self.T.values()))
        self.opt_R = mk(list(self.R.values()))
        self.opt_D = mk(list(self.D.values()) + [p for s in self.D_mr for p in s.values()])

    def load_from(self, params_T, params_R, params_D, params_D_mr, adam_T, adam_R, adam_D):
        """Teacher forcing for multi-step parity: take parameters (dict name -> tensor) and Adam state
        (step, dict name -> exp_avg, dict name -> exp_avg_sq) from another implementation."""
        def put(dst, src, opt, adam):
            step, m, v = adam
            with torch.no_grad():
                for k, p in dst.items():
                    p.copy_(src[k].to(self.dtype))
                    if step > 0:
                        st = opt.state[p]
                        st['step'] = torch.tensor(float(step))
                        st['exp_avg'] = m[k].to(self.dtype).clone()
                        st['exp_avg_sq'] = v[k].to(self.dtype).clone()
        put(self.T, params_T, self.opt_T, adam_T)
        put(self.R, params_R, self.opt_R, adam_R)
        put(self.D, params_D, self.opt_D, adam_D[0])
        for d, s, a in zip(self.D_mr, params_D_mr, adam_D[1:]):
            put(d, s, self.opt_D, a)

    def _netT(self, x):
        return resnet_generator(self.T, x, self.cfg['n_blocks'])

    def _netR(self, a, b, apply_on):
        if self.cfg['stn_type'] == 'unet':
            return unet_stn(self.R, a, b, apply_on, self.cfg['alpha'], self.cfg['multires_reg'], self.cfg['stn_cfg'])
        return affine_stn(self.R, a, b, apply_on)

    def _d_all(self, a, img, real, knife=None):
        """sum over the full-res D and every reduced-resolution D of GANLoss(D(cat(a, img)), real)."""
        m = self.cfg['gan_mode']
        loss = gan_loss(nlayer_discriminator(self.D, torch.cat([a, img], 1), knife), real, m)
        for i, Dm in enumerate(self.D_mr):
            size = (a.size(2) // 2 ** (i + 1), a.size(3) // 2 ** (i + 1))
            ar = F.interpolate(a, size, mode='bilinear', align_corners=False)
            ir = F.interpolate(img, size, mode='bilinear', align_corners=False)
            loss = loss + gan_loss(nlayer_discriminator(Dm, torch.cat([ar, ir], 1), knife), real, m)
        return loss

    def forward(self, A, B):
        A, B = A.to(self.dtype), B.to(self.dtype)
        self.real_A, self.real_B = A, B
        self.fake_B = self._netT(A)
        warped, self.reg, self.offsets = self._netR(A, B, [A, self.fake_B])
        self.registered_real_A = warped[0]
        self.fake_TR_B = self._netT(self.registered_real_A)
        self.fake_RT_B = warped[1]

    @staticmethod
    def _freeze(dicts, flag):
        for d in dicts:
            for p in d.values():
                p.requires_grad_(flag)

    def optimize_parameters(self, A, B):
        c = self.cfg
        A, B = A.to(self.dtype), B.to(self.dtype)
        self.forward(A, B)
        losses = OrderedDict()
        # discriminator step (reference :217-264, 271-275)
        self._freeze([self.T, self.R], False)
        self.opt_D.zero_grad()
        d_real = self._d_all(A, B, True)
        d_tr = self._d_all(A, self.fake_TR_B.detach(), False)
        d_rt = self._d_all(A, self.fake_RT_B.detach(), False)
        loss_D = 0.5 * c['lambda_GAN'] * (d_real + d_tr + d_rt)
        loss_D.backward()
        self.grads_D = OrderedDict((k, p.grad.clone()) for k, p in self.D.items())
        self.grads_D_mr = [OrderedDict((k, p.grad.clone()) for k, p in d.items()) for d in self.D_mr]
        if self.knife_band:
            knife = {'band': self.knife_band, 'count': 0}
            alt = 0.5 * c['lambda_GAN'] * (self._d_all(A, B, True, knife) + self._d_all(A, self.fake_TR_B.detach(), False, knife)
                                           + self._d_all(A, self.fake_RT_B.detach(), False, knife))
            ps = list(self.D.values()) + [p for d in self.D_mr for p in d.values()]
            gs = iter(torch.autograd.grad(alt, ps))
            self.grads_D_knife = OrderedDict((k, next(gs)) for k in self.D)
            self.grads_D_mr_knife = [OrderedDict((k, next(gs)) for k in d) for d in self.D_mr]
            self.knife_count = knife['count']
        self.opt_D.step()
        self._freeze([self.T, self.R], True)
        # translation + registration step (reference :175-215, 278-284)
        self._freeze([self.D] + self.D_mr, False)
        self.opt_R.zero_grad()
        self.opt_T.zero_grad()
        l1_tr = c['lambda_recon'] * F.l1_loss(self.fake_TR_B, B)
        g_tr = c['lambda_GAN'] * self._d_all(A, self.fake_TR_B, True)
        l1_rt = c['lambda_recon'] * F.l1_loss(self.fake_RT_B, B)
        g_rt = c['lambda_GAN'] * self._d_all(A, self.fake_RT_B, True)
        smooth = c['lambda_smooth'] * self.reg
        (l1_tr + l1_rt + g_tr + g_rt + smooth).backward()
        self.grads_T = OrderedDict((k, p.grad.clone()) for k, p in self.T.items())
        self.grads_R = OrderedDict((k, p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                                   for k, p in self.R.items())
        self.opt_R.step()
        self.opt_T.step()
        self._freeze([self.D] + self.D_mr, True)
        for k, v in (('L1_TR', l1_tr), ('GAN_TR', g_tr), ('L1_RT', l1_rt), ('GAN_RT', g_rt), ('smoothness', smooth),
                     ('D_fake_TR', d_tr), ('D_fake_RT', d_rt), ('D', loss_D)):
            losses[k] = float(v)
        self.losses = losses
        return losses
