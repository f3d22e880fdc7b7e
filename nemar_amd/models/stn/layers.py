"""Building blocks of the registration networks — mirror of reference models/stn/layers.py (Conv :73-106,
DownBlock :158-185, ResnetTransformer :218-242, get_init_function :25-55, get_activation :58-70) with the same
parameter names (`conv2d`, `resnet_block.model.N.conv_block.{1,5}`), executed by the fused gfx950 kernels.
UpBlock and AttentionGate (reference :109-155,188-215) have no caller in the reference and are not provided."""
from functools import partial

import torch
from torch import nn

from ... import ops
from ..networks import ConvParams, ResnetBlock, Slots, _ref

_ACT = {'relu': ops.ACT_RELU, 'leaky_relu': ops.ACT_LRELU, 'tanh': ops.ACT_TANH, None: ops.ACT_NONE}


def get_init_function(activation, init_function, **kwargs):
    """Initialiser for a conv weight, by name (reference :25-55).  'zeros' is N(0, 1e-5), as in the reference."""
    a = 0.0
    if activation == 'leaky_relu':
        a = kwargs.get('negative_slope', 0.2)
    gain = kwargs.get('gain', 0.02)
    if isinstance(init_function, str):
        if init_function == 'kaiming':
            activation = 'relu' if activation is None else activation
            return partial(torch.nn.init.kaiming_normal_, a=a, nonlinearity=activation, mode='fan_in')
        if init_function == 'dirac':
            return torch.nn.init.dirac_
        if init_function == 'xavier':
            activation = 'relu' if activation is None else activation
            g = torch.nn.init.calculate_gain(nonlinearity=activation, param=a)
            return partial(torch.nn.init.xavier_normal_, gain=g)
        if init_function == 'normal':
            return partial(torch.nn.init.normal_, mean=0.0, std=gain)
        if init_function == 'orthogonal':
            return partial(torch.nn.init.orthogonal_, gain=gain)
        if init_function == 'zeros':
            return partial(torch.nn.init.normal_, mean=0.0, std=1e-5)
        return None
    if init_function is None:
        if activation in ('relu', 'leaky_relu'):
            return partial(torch.nn.init.kaiming_normal_, a=a, nonlinearity=activation)
        if activation in ('tanh', 'sigmoid'):
            g = torch.nn.init.calculate_gain(nonlinearity=activation, param=a)
            return partial(torch.nn.init.xavier_normal_, gain=g)
        return None
    return init_function


def get_activation(activation, **kwargs):
    """Kernel activation code for a name (reference :58-70 returns nn modules; here it is an epilogue selector)."""
    if activation == 'sigmoid':
        raise NotImplementedError('sigmoid is only used by the reference\'s dead AttentionGate')
    return _ACT.get(activation, ops.ACT_NONE)


class ResnetTransformer(nn.Module):
    """n ResnetBlocks (reflect pad, InstanceNorm, ReLU, no dropout) — reference :218-242."""

    def __init__(self, dim, n_blocks, init_func):
        super().__init__()
        self.model = Slots()
        init_ = get_init_function('relu', init_func)
        for i in range(n_blocks):
            blk = self.model.put(i, ResnetBlock(dim, padding_type='reflect', norm_layer='instance',
                                                use_dropout=False, use_bias=True))
            for c in (blk.c1, blk.c2):
                init_(c.weight)
                c.bias.data.zero_()
        self.n_blocks = n_blocks

    def forward(self, x):
        for i in range(self.n_blocks):
            x = self.model.at(i)(x)
        return x


class Conv(nn.Module):
    """conv -> (InstanceNorm) -> activation -> (ResnetTransformer) — reference :73-106.  The activation is fused into
    the conv epilogue (or into the InstanceNorm kernel when a norm is present); `x2` is an optional second input
    that is logically concatenated after `x` (the decoder's skip connection) without materialising the concat."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, bias=True, activation='relu',
                 init_func='kaiming', use_norm=False, use_resnet=False, **kwargs):
        super().__init__()
        self.conv2d = ConvParams(in_channels, out_channels, kernel_size, bias=bias)
        self.resnet_block = ResnetTransformer(out_channels, 1, init_func) if use_resnet else None
        self.stride, self.padding, self.use_norm = stride, padding, use_norm
        self.act = get_activation(activation)
        self.slope = kwargs.get('negative_slope', 0.2)
        init_ = get_init_function(activation, init_func)
        init_(self.conv2d.weight)
        if self.conv2d.bias is not None:
            self.conv2d.bias.data.zero_()

    def forward(self, x, x2=None):
        c = self.conv2d
        if self.use_norm:
            h = ops.conv2d(x, c.weight, c.bias, self.stride, self.padding, ops.PAD_ZERO, x2=x2)
            h = ops.instance_norm(h, act=self.act, slope=self.slope)
        else:
            h = ops.conv2d(x, c.weight, c.bias, self.stride, self.padding, ops.PAD_ZERO, act=self.act,
                           slope=self.slope, x2=x2)
        if self.resnet_block is not None:
            h = self.resnet_block(h)
        return h


class DownBlock(nn.Module):
    """Conv (-> Conv) -> MaxPool2d(2), returning (pooled, skip) — reference :158-185."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, bias=False, activation='relu',
                 init_func='kaiming', use_norm=False, use_resnet=False, skip=True, refine=False, pool=True,
                 pool_size=2, **kwargs):
        super().__init__()
        if pool and pool_size != 2:
            raise NotImplementedError('only 2x2 max-pooling has a kernel')
        self.conv_0 = Conv(in_channels, out_channels, kernel_size, stride, padding, bias=bias, activation=activation,
                           init_func=init_func, use_norm=use_norm, use_resnet=use_resnet, **kwargs)
        self.conv_1 = None
        if refine:
            self.conv_1 = Conv(out_channels, out_channels, kernel_size, stride, padding, bias=bias,
                               activation=activation, init_func=init_func, use_norm=use_norm, use_resnet=use_resnet,
                               **kwargs)
        self.skip, self.pool = skip, pool

    def forward(self, x, x2=None):
        x = skip = self.conv_0(x, x2)
        if self.conv_1 is not None:
            x = skip = self.conv_1(x)
        if self.pool:
            if self.skip:
                # two consumers (the pooling and the decoder): the decoder's gradient is the addend of the pooling's backward kernel
                x, skip = ops.max_pool2_with_skip(x)
            else:
                x = ops.max_pool2(x)
        return (x, skip) if self.skip else x
