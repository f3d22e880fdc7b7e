"""Translation generator, PatchGAN discriminator, GAN objective and their factories — the MI355X-native mirror of
the reference's models/networks.py for the NeMAR hot path.

Same public names, arguments and error behaviour as the reference (define_G :116-165, define_D :168-209, GANLoss
:215-281, ResnetGenerator :323-386, ResnetBlock :389-446, NLayerDiscriminator :556-602,
init_weights :62-96, get_norm_layer :12-29, get_scheduler :32-59) and the same state_dict key layout (SURVEY.md
Appendix C), so checkpoints move in both directions.  The arithmetic is different by construction: every layer is
a call into the gfx950 kernel library (nemar_amd.ops) with padding, concatenation, bias, activation, InstanceNorm
epilogues and residual adds fused — there is no nn.Conv2d / nn.InstanceNorm2d / F.* on the path.

Only instance normalisation (the reference default, `--norm instance`) and `--norm none` have kernels; `batch`
raises NotImplementedError (cross-sample statistics are outside the per-sample hot path).
"""
import math
import os

import torch
import torch.nn as nn

from .. import ops


# ------------------------------------------------------------------------------------------------------
# parameter containers (no compute): keep the reference's module tree so state_dict keys line up
# ------------------------------------------------------------------------------------------------------
class ConvParams(nn.Module):
    """weight [K,C,R,S] (+ bias [K]) of one convolution; `transposed` stores [Cin,Cout,R,S] like nn.ConvTranspose2d."""

    def __init__(self, in_ch, out_ch, k, bias=True, transposed=False):
        super().__init__()
        shape = (in_ch, out_ch, k, k) if transposed else (out_ch, in_ch, k, k)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.zeros(out_ch)) if bias else None
        self.k, self.transposed = k, transposed
        # nn.Conv2d's default (kaiming-uniform a=sqrt(5)) so an un-initialised net is still sane
        fan_in = (out_ch if transposed else in_ch) * k * k
        bound = 1.0 / math.sqrt(fan_in)
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)


class LinearParams(nn.Module):
    """weight [out,in], bias [out] with nn.Linear's key names; applied as a 1x1 convolution on a 1x1 image."""

    def __init__(self, in_f, out_f):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_f, in_f))
        self.bias = nn.Parameter(torch.zeros(out_f))
        bound = 1.0 / math.sqrt(in_f)       # nn.Linear's default: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +1/sqrt(fan_in))
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)


def _ref(owner, name, module):
    """Keep a convenience handle to a sub-module WITHOUT registering it again (it already lives in a Slots tree, and
    a second registration would duplicate its state_dict keys)."""
    object.__setattr__(owner, name, module)
    return module


class Slots(nn.Module):
    """Children addressed by the integer positions the reference's nn.Sequential gives its parametrised layers."""

    def put(self, index, module):
        self.add_module(str(index), module)
        return module

    def at(self, index):
        return self._modules[str(index)]


# ------------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------------
def get_norm_layer(norm_type='instance'):
    """Name of the normalisation the kernels will fuse: 'instance' | 'none'.  (reference :12-29)"""
    if norm_type == 'instance':
        return 'instance'
    if norm_type == 'none':
        return None
    if norm_type == 'batch':
        raise NotImplementedError('normalization layer [batch] has no MI355X kernel on the NeMAR hot path; '
                                  'use --norm instance (the reference default)')
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def get_scheduler(optimizer, opt):
    """Learning-rate policy (reference :32-59).  The reference constructs these but train.py never steps them."""
    if opt.lr_policy == 'linear':
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + opt.epoch_count - opt.niter) / float(opt.niter_decay + 1)
        return LambdaLR(optimizer, lambda_rule)
    if opt.lr_policy == 'step':
        return LambdaLR(optimizer, lambda e: 0.1 ** (e // opt.lr_decay_iters))
    if opt.lr_policy == 'cosine':
        return LambdaLR(optimizer, lambda e: 0.5 * (1.0 + math.cos(math.pi * e / opt.niter)))
    if opt.lr_policy == 'plateau':
        return PlateauLR(optimizer, factor=0.2, threshold=0.01, patience=5)
    return NotImplementedError('learning rate policy [%s] is not implemented', opt.lr_policy)


class LambdaLR:
    """Multiplicative LR schedule over FlatAdam.param_groups (lr = base_lr * fn(epoch))."""

    def __init__(self, optimizer, fn):
        self.optimizer, self.fn, self.epoch = optimizer, fn, 0
        self.base = [g['lr'] for g in optimizer.param_groups]
        self._apply()

    def _apply(self):
        for g, b in zip(self.optimizer.param_groups, self.base):
            g['lr'] = b * self.fn(self.epoch)

    def step(self, metric=None):
        self.epoch += 1
        self._apply()


class PlateauLR:
    def __init__(self, optimizer, factor, threshold, patience):
        self.optimizer, self.factor, self.threshold, self.patience = optimizer, factor, threshold, patience
        self.best, self.bad = float('inf'), 0

    def step(self, metric):
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
            if self.bad > self.patience:
                for g in self.optimizer.param_groups:
                    g['lr'] *= self.factor
                self.bad = 0


def init_weights(net, init_type='normal', init_gain=0.02):
    """(Re-)initialise every conv / linear weight of `net` (reference :62-96): normal | xavier | kaiming | orthogonal,
    biases to zero."""
    if init_type not in ('normal', 'xavier', 'kaiming', 'orthogonal'):
        raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
    for m in net.modules():
        if isinstance(m, (ConvParams, LinearParams)):
            w = m.weight.data
            if init_type == 'normal':
                nn.init.normal_(w, 0.0, init_gain)
            elif init_type == 'xavier':
                nn.init.xavier_normal_(w, gain=init_gain)
            elif init_type == 'kaiming':
                nn.init.kaiming_normal_(w, a=0, mode='fan_in')
            else:
                nn.init.orthogonal_(w, gain=init_gain)
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)
    ops.invalidate_packed_weights()      # .data writes do not bump tensor._version: drop every cached packed image
    print('initialize network with %s' % init_type)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """Place the network on its device and initialise it (reference :98-113).  One process drives one GPU: data
    parallelism is per-process (nemar_amd.distributed), never nn.DataParallel."""
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(torch.device('cuda', gpu_ids[0]))
    init_weights(net, init_type, init_gain=init_gain)
    return net


_RESNET_BLOCKS = {'resnet_9blocks': 9, 'resnet_6blocks': 6, 'resnet_3blocks': 3, 'resnet_4blocks': 4,
                  'resnet_5blocks': 5}
_UNET_DOWNS = {'unet_128': 7, 'unet_256': 8}       # reference :157-160


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02,
             gpu_ids=[]):
    """Create the translation generator (reference :116-165)."""
    norm_layer = get_norm_layer(norm_type=norm)
    if netG in _RESNET_BLOCKS:
        net = ResnetGenerator(input_nc, output_nc, ngf, norm_layer=norm_layer, use_dropout=use_dropout,
                              n_blocks=_RESNET_BLOCKS[netG])
    elif netG in _UNET_DOWNS:
        net = UnetGenerator(input_nc, output_nc, _UNET_DOWNS[netG], ngf, norm_layer=norm_layer, use_dropout=use_dropout)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % netG)
    return init_net(net, init_type, init_gain, gpu_ids)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm='batch', init_type='normal', init_gain=0.02, gpu_ids=[]):
    """Create the discriminator (reference :168-209)."""
    norm_layer = get_norm_layer(norm_type=norm)
    if netD == 'basic':
        net = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm_layer=norm_layer)
    elif netD == 'n_layers':
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm_layer=norm_layer)
    elif netD == 'pixel':
        net = PixelDiscriminator(input_nc, ndf, norm_layer=norm_layer)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % netD)
    return init_net(net, init_type, init_gain, gpu_ids)


# ------------------------------------------------------------------------------------------------------
class GANLoss(nn.Module):
    """GAN objective against a constant label (reference :215-281): vanilla (BCE-with-logits) | lsgan | wgangp.
    `weight` folds the caller's lambda into the kernel so no scalar arithmetic follows."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        if gan_mode not in ('lsgan', 'vanilla', 'wgangp'):
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)
        if target_real_label != 1.0 or target_fake_label != 0.0:
            raise NotImplementedError('GANLoss kernels assume labels 1.0 / 0.0')
        self.gan_mode = gan_mode

    def __call__(self, prediction, target_is_real, weight=1.0):
        return ops.gan_loss(prediction, bool(target_is_real), self.gan_mode, weight)


# ------------------------------------------------------------------------------------------------------
class UnetSkipConnectionBlock(nn.Module):
    """|-- down: LeakyReLU, conv k4 s2 [, norm] -- submodule -- up: ReLU, convT k4 s2 [, norm][, dropout] --| with the
    identity skip concatenated in front (reference :486-553).  Parameter slots follow the reference's nn.Sequential
    positions (outermost: model.0 / model.1 (submodule) / model.3; innermost: model.1 / model.3; otherwise model.1 /
    model.3 (submodule) / model.5).

    The reference's first layer is an in-place LeakyReLU, so the tensor it concatenates as "x" is already
    LeakyReLU(x), and the parent's in-place ReLU then acts on the whole concatenation.  Here the activations ride in the
    producers' epilogues instead: a block RECEIVES LeakyReLU(x) (fused into the parent's conv / InstanceNorm) and
    RETURNS ReLU(cat[LeakyReLU(x), up]) = cat[ReLU(x), ReLU(up)] (ReLU fused into its own InstanceNorm / convT)."""

    def __init__(self, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False,
                 norm_layer='instance', use_dropout=False):
        super().__init__()
        self.outermost, self.innermost, self.norm, self.use_dropout = outermost, innermost, norm_layer, use_dropout
        use_bias = norm_layer == 'instance'
        if input_nc is None:
            input_nc = outer_nc
        m = self.model = Slots()
        if outermost:
            _ref(self, 'down', m.put(0, ConvParams(input_nc, inner_nc, 4, bias=use_bias)))
            _ref(self, 'sub', m.put(1, submodule))
            _ref(self, 'up', m.put(3, ConvParams(inner_nc * 2, outer_nc, 4, bias=True, transposed=True)))
        elif innermost:
            _ref(self, 'down', m.put(1, ConvParams(input_nc, inner_nc, 4, bias=use_bias)))
            object.__setattr__(self, 'sub', None)
            _ref(self, 'up', m.put(3, ConvParams(inner_nc, outer_nc, 4, bias=use_bias, transposed=True)))
        else:
            _ref(self, 'down', m.put(1, ConvParams(input_nc, inner_nc, 4, bias=use_bias)))
            _ref(self, 'sub', m.put(3, submodule))
            _ref(self, 'up', m.put(5, ConvParams(inner_nc * 2, outer_nc, 4, bias=use_bias, transposed=True)))

    def forward(self, a):
        """a: the block input (outermost) or LeakyReLU(block input) (otherwise)."""
        if self.outermost:
            child_in = ops.conv2d(a, self.down.weight, self.down.bias, 2, 1, ops.PAD_ZERO, act=ops.ACT_LRELU)
            s = self.sub(child_in)
            return ops.conv_transpose2d(s, self.up.weight, self.up.bias, 2, 1, 0, act=ops.ACT_TANH)
        if self.innermost:
            s = ops.conv2d(a, self.down.weight, self.down.bias, 2, 1, ops.PAD_ZERO, act=ops.ACT_RELU)
        else:
            d = ops.conv2d(a, self.down.weight, self.down.bias, 2, 1, ops.PAD_ZERO,
                           act=ops.ACT_NONE if self.norm else ops.ACT_LRELU)
            s = self.sub(_norm_act(d, self.norm, ops.ACT_LRELU))
        u = ops.conv_transpose2d(s, self.up.weight, self.up.bias, 2, 1, 0, act=ops.ACT_NONE if self.norm else ops.ACT_RELU)
        u = _norm_act(u, self.norm, ops.ACT_RELU)
        if self.use_dropout:
            u = ops.dropout(u, 0.5, self.training)      # commutes with the parent's ReLU (scale >= 0)
        return torch.cat([ops.activation(a, ops.ACT_RELU), u], 1)


class UnetGenerator(nn.Module):
    """U-Net generator built from the innermost block outwards (reference :449-483)."""

    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_layer='instance', use_dropout=False):
        super().__init__()
        block = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=None, norm_layer=norm_layer, innermost=True)
        for _ in range(num_downs - 5):
            block = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=block, norm_layer=norm_layer,
                                            use_dropout=use_dropout)
        block = UnetSkipConnectionBlock(ngf * 4, ngf * 8, submodule=block, norm_layer=norm_layer)
        block = UnetSkipConnectionBlock(ngf * 2, ngf * 4, submodule=block, norm_layer=norm_layer)
        block = UnetSkipConnectionBlock(ngf, ngf * 2, submodule=block, norm_layer=norm_layer)
        self.model = UnetSkipConnectionBlock(output_nc, ngf, input_nc=input_nc, submodule=block, outermost=True,
                                             norm_layer=norm_layer)

    def forward(self, x):
        return self.model(x)


# ------------------------------------------------------------------------------------------------------
def _norm_act(x, norm, act, residual=None, planes=False, dropout_p=0.0):
    """InstanceNorm + activation (+ dropout) (+ residual) as one kernel; with norm None the activation was already fused.
    planes: the consumer is a 3x3 reflect-padded convolution (ops.instance_norm)."""
    if norm == 'instance':
        return ops.instance_norm(x, act=act, residual=residual, planes=planes, dropout_p=dropout_p)
    if dropout_p > 0.0:
        x = ops.dropout(x, dropout_p, True)
    if residual is not None:
        raise NotImplementedError('residual add without normalisation')
    return x


class ResnetBlock(nn.Module):
    """x + IN(conv3x3(reflect(drop(relu(IN(conv3x3(reflect(x)))))))) — reference :389-446.
    Parameter slots follow the reference's nn.Sequential positions: conv_block.1 and conv_block.5
    (conv_block.6 when the Dropout layer is present)."""

    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias):
        super().__init__()
        if padding_type not in ('reflect', 'zero'):
            raise NotImplementedError('padding [%s] is not implemented' % padding_type)
        self.pad_mode = ops.PAD_REFLECT if padding_type == 'reflect' else ops.PAD_ZERO
        self.norm, self.use_dropout = norm_layer, use_dropout
        self.feeds_block = False        # set by the generator: the next layer is another ResnetBlock (a 3x3 reflect convolution)
        self.conv_block = Slots()
        first = 1 if padding_type == 'reflect' else 0
        _ref(self, 'c1', self.conv_block.put(first, ConvParams(dim, dim, 3, bias=use_bias)))
        second = first + 2 + (1 if use_dropout else 0) + (2 if padding_type == 'reflect' else 1)
        _ref(self, 'c2', self.conv_block.put(second, ConvParams(dim, dim, 3, bias=use_bias)))

    def forward(self, x):
        fused_act = ops.ACT_NONE if self.norm else ops.ACT_RELU
        reflect = self.pad_mode == ops.PAD_REFLECT
        if reflect and self.norm == 'instance':
            # the whole block as one autograd node where the wide-layer route takes it (ops._ResBlock): producer-written operand planes
            # for all six convolution calls, the skip gradient added in a data gradient's epilogue
            out = ops.resnet_block(x, self.c1.weight, self.c1.bias, self.c2.weight, self.c2.bias,
                                   dropout_p=0.5 if (self.use_dropout and self.training) else 0.0, feeds_block=self.feeds_block)
            if out is not None:
                return out
        # the block's input feeds conv1 and the skip: the skip's gradient is added in the last pass of conv1's data gradient
        h, x_skip = ops.conv2d_with_skip(x, self.c1.weight, self.c1.bias, 1, 1, self.pad_mode, act=fused_act)
        # (norm + ReLU + Dropout in one pass; where conv2 runs on the fp16 x 3 route its operand planes come out of the same pass)
        h = _norm_act(h, self.norm, ops.ACT_RELU, planes=reflect, dropout_p=0.5 if (self.use_dropout and self.training) else 0.0)
        h = ops.conv2d(h, self.c2.weight, self.c2.bias, 1, 1, self.pad_mode)
        if self.norm:
            return _norm_act(h, self.norm, ops.ACT_NONE, residual=x_skip, planes=reflect and self.feeds_block)
        return x_skip + h


class ResnetGenerator(nn.Module):
    """c7s1-ngf, d2ngf, d4ngf, n x R4ngf, u2ngf, u-ngf, c7s1-out + tanh (reference :323-386)."""

    def __init__(self, input_nc, output_nc, ngf=64, norm_layer='instance', use_dropout=False, n_blocks=6,
                 padding_type='reflect'):
        assert n_blocks >= 0
        super().__init__()
        use_bias = norm_layer == 'instance'
        self.norm, self.n_blocks = norm_layer, n_blocks
        per = 3 if norm_layer else 2          # layers per conv stage in the reference Sequential
        m = self.model = Slots()
        idx = 1
        _ref(self, 'stem', m.put(idx, ConvParams(input_nc, ngf, 7, bias=use_bias)))
        idx += per
        self.down = []
        for i in range(2):
            mult = 2 ** i
            self.down.append(m.put(idx, ConvParams(ngf * mult, ngf * mult * 2, 3, bias=use_bias)))
            idx += per
        self.blocks = []
        for _ in range(n_blocks):
            self.blocks.append(m.put(idx, ResnetBlock(ngf * 4, padding_type, norm_layer, use_dropout, use_bias)))
            idx += 1
        for b in self.blocks[:-1]:
            b.feeds_block = True
        self.reflect_blocks = padding_type == 'reflect' 
        self.up = []
        for i in range(2):
            mult = 2 ** (2 - i)
            self.up.append(m.put(idx, ConvParams(ngf * mult, ngf * mult // 2, 3, bias=use_bias, transposed=True)))
            idx += per
        idx += 1                               # the ReflectionPad2d(3) before the head
        _ref(self, 'head', m.put(idx, ConvParams(ngf, output_nc, 7, bias=True)))

    def forward(self, x):
        a = ops.ACT_NONE if self.norm else ops.ACT_RELU
        h = ops.conv2d(x, self.stem.weight, self.stem.bias, 1, 3, ops.PAD_REFLECT, act=a)
        h = _norm_act(h, self.norm, ops.ACT_RELU)
        for i, c in enumerate(self.down):
            h = ops.conv2d(h, c.weight, c.bias, 2, 1, ops.PAD_ZERO, act=a)
            h = _norm_act(h, self.norm, ops.ACT_RELU, planes=self.reflect_blocks and self.n_blocks > 0 and i == len(self.down) - 1)
        for b in self.blocks:
            h = b(h)
        for c in self.up:
            h = ops.conv_transpose2d(h, c.weight, c.bias, 2, 1, 1, act=a)
            h = _norm_act(h, self.norm, ops.ACT_RELU)
        return ops.conv2d(h, self.head.weight, self.head.bias, 1, 3, ops.PAD_REFLECT, act=ops.ACT_TANH)

    def init_to_identity(self):
        self.head.weight.data.normal_(mean=0.0, std=1e-5)
        ops.invalidate_packed_weights()


class NLayerDiscriminator(nn.Module):
    """PatchGAN: C64(k4s2)-LReLU, [C(k4s2)-IN-LReLU] x (n-1), C(k4s1)-IN-LReLU, C1(k4s1) — reference :556-602."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer='instance'):
        super().__init__()
        use_bias = norm_layer != 'batch'
        self.norm = norm_layer
        per = 3 if norm_layer else 2
        m = self.model = Slots()
        self.layers = []                      # (params, stride)
        idx = 0
        self.layers.append((m.put(idx, ConvParams(input_nc, ndf, 4, bias=True)), 2, False))
        idx += 2
        nf = 1
        for n in range(1, n_layers):
            prev, nf = nf, min(2 ** n, 8)
            self.layers.append((m.put(idx, ConvParams(ndf * prev, ndf * nf, 4, bias=use_bias)), 2, True))
            idx += per
        prev, nf = nf, min(2 ** n_layers, 8)
        self.layers.append((m.put(idx, ConvParams(ndf * prev, ndf * nf, 4, bias=use_bias)), 1, True))
        idx += per
        _ref(self, 'final', m.put(idx, ConvParams(ndf * nf, 1, 4, bias=True)))
        # Schedule hint (ops._Conv2d backward): D's backward chain is short (four data gradients) and its weight gradients long (the 4x4
        # stride-2 layers run on the exact-fp32 kernels), so with all of them on the side stream the compute stream waited 0.53 ms per step
        # at the join before D's Adam step (tools/side_tail.py).  The second layer's weight gradient stays on the compute stream: the two
        # lanes then end together.  NEMAR_D_WGRAD_MAIN=0 switches the hint off (A/B of the schedule; the results are the same bits).
        if len(self.layers) >= 3 and os.environ.get('NEMAR_D_WGRAD_MAIN', '1') != '0':
            self.layers[1][0].weight._nemar_wgrad_main = True

    def forward(self, x, x2=None):
        """`x2`: optional second tensor, logically concatenated after `x` along channels (the (real_A, image)
        pair of reference models/nemar_model.py:181,220) without materialising the concat."""
        h = x
        for i, (c, stride, normed) in enumerate(self.layers):
            fuse = ops.ACT_LRELU if not (normed and self.norm) else ops.ACT_NONE
            h = ops.conv2d(h, c.weight, c.bias, stride, 1, ops.PAD_ZERO, act=fuse, slope=0.2,
                           x2=x2 if i == 0 else None)
            if normed and self.norm:
                h = ops.instance_norm(h, act=ops.ACT_LRELU, slope=0.2)
        return ops.conv2d(h, self.final.weight, self.final.bias, 1, 1, ops.PAD_ZERO)


class PixelDiscriminator(nn.Module):
    """1x1 PatchGAN (reference :605-634); not reached by NeMAR defaults, provided for `--netD pixel`."""

    def __init__(self, input_nc, ndf=64, norm_layer='instance'):
        super().__init__()
        use_bias = norm_layer != 'instance'
        self.norm = norm_layer
        n = self.net = Slots()
        _ref(self, 'c0', n.put(0, ConvParams(input_nc, ndf, 1, bias=True)))
        _ref(self, 'c1', n.put(2, ConvParams(ndf, ndf * 2, 1, bias=use_bias)))
        _ref(self, 'c2', n.put(5 if norm_layer else 4, ConvParams(ndf * 2, 1, 1, bias=use_bias)))

    def forward(self, x, x2=None):
        h = ops.conv2d(x, self.c0.weight, self.c0.bias, act=ops.ACT_LRELU, x2=x2)
        if self.norm:
            h = ops.conv2d(h, self.c1.weight, self.c1.bias)
            h = ops.instance_norm(h, act=ops.ACT_LRELU)
        else:
            h = ops.conv2d(h, self.c1.weight, self.c1.bias, act=ops.ACT_LRELU)
        return ops.conv2d(h, self.c2.weight, self.c2.bias)
