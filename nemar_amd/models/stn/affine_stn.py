"""Affine registration network — mirror of reference models/stn/affine_stn.py (AffineNetwork :22-83, AffineSTN
:86-138).  theta = dtheta + I and F.affine_grid are fused into the warp kernel (GRID_AFFINE); the two nn.Linear
layers run as 1x1 convolutions on a 1x1 image through the same MFMA implicit-GEMM kernels."""
import torch.nn as nn

from ... import ops
from ..networks import LinearParams, Slots
from .layers import DownBlock

cfg_conv1_nf = {'A': 32, }
cfg_mlp_nf = {'A': 256}
cfg_use_norm = {'A': True, }
cfg_nconvs = {'A': 5, }
cfg_use_resnet = {'A': False, }
cfg_activation = {'A': 'relu'}


class AffineNetwork(nn.Module):
    """5 x [conv3x3 - InstanceNorm - ReLU - maxpool] then Linear(256) - ReLU - Linear(6) (reference :22-83)."""

    def __init__(self, in_channels_a, in_channels_b, height, width, cfg='A', init_func='kaiming'):
        super().__init__()
        self.h, self.w = height, width
        self.nconvs = cfg_nconvs[cfg]
        self.convs = Slots()
        prev_nf = in_channels_a + in_channels_b
        nf = cfg_conv1_nf[cfg]
        for i in range(self.nconvs):
            self.convs.put(i, DownBlock(prev_nf, nf, 3, 1, 1, bias=True, activation=cfg_activation[cfg],
                                        init_func=init_func, use_norm=cfg_use_norm[cfg],
                                        use_resnet=cfg_use_resnet[cfg], skip=False, refine=False, pool=True))
            prev_nf = nf
            nf = min(2 * nf, cfg_mlp_nf[cfg])
        self.local = Slots()
        self.local.put(0, LinearParams(prev_nf * (self.h // 2 ** self.nconvs) * (self.w // 2 ** self.nconvs), nf))
        self.local.put(2, LinearParams(nf, 6))
        # start at the identity transformation (reference :75-76)
        self.local.at(2).weight.data.normal_(mean=0.0, std=5e-4)
        self.local.at(2).bias.data.zero_()

    def forward(self, img_a, img_b):
        x, x2 = img_a, img_b
        for i in range(self.nconvs):
            x = self.convs.at(i)(x, x2)
            x2 = None
        n = x.size(0)
        x = x.reshape(n, -1, 1, 1)
        l0, l2 = self.local.at(0), self.local.at(2)
        x = ops.conv2d(x, l0.weight, l0.bias, act=ops.ACT_RELU, wshape=(l0.weight.size(0), l0.weight.size(1), 1, 1))
        x = ops.conv2d(x, l2.weight, l2.bias, wshape=(6, l2.weight.size(1), 1, 1))
        return x.reshape(n, 6)


class AffineSTN(nn.Module):
    """Predicts and applies the affine transformation (reference :86-138)."""

    def __init__(self, nc_a, nc_b, height, width, cfg, init_func):
        super().__init__()
        self.net = AffineNetwork(nc_a, nc_b, height, width, cfg, init_func)

    def _get_theta(self, img_a, img_b):
        import torch
        dtheta = self.net(img_a, img_b)
        ident = torch.tensor([1, 0, 0, 0, 1, 0], dtype=torch.float32, device=dtheta.device)
        return dtheta + ident[None]

    def get_grid(self, img_a, img_b):
        """F.affine_grid of the predicted theta (reference :102-106); inspection API, built with torch ops."""
        import torch.nn.functional as F
        theta = self._get_theta(img_a, img_b)
        return F.affine_grid(theta.view(-1, 2, 3), img_a.size(), align_corners=False)

    # predict -> warp* -> regularization: same split as UnetSTN (see there)
    def predict(self, img_a, img_b):
        return self.net(img_a, img_b)

    def warp(self, field, imgs):
        return ops.warp_affine(field, list(imgs))

    def fork_field(self, field, n_warps):
        """-> ([one theta handle per warp() call], the handle for regularization()) — ops.fork, as UnetSTN.fork_field"""
        hs = ops.fork(field, n_warps + 1)
        return list(hs[:n_warps]), hs[n_warps]

    def regularization(self, field, warped_first=None):
        return self._calculate_regularization_term(field)

    def forward(self, img_a, img_b, apply_on=None):
        (f_warp,), f_reg = self.fork_field(self.predict(img_a, img_b), 1)
        warped = self.warp(f_warp, [img_a] if apply_on is None else apply_on)
        return warped, self.regularization(f_reg, warped[0])

    def _calculate_regularization_term(self, theta):
        """mean|dtheta| (reference :136-138)."""
        return ops.l1_loss(theta, None, 1.0)
