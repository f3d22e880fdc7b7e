"""`--dataset_mode gpupairs`: aligned A/B image pairs held in HBM, augmented on the GPU.

The pool is [M,3,H,W] per modality with values in [0,1] (what ToTensor produces): `--dataroot synthetic` builds it from
the seeded generator used by the tests, any other dataroot is a directory with `A.npy` / `B.npy` ([M,3,H,W] or [M,H,W,3],
uint8 or float).  A batch is ONE launch of nemar_crop_flip_normalize per modality: the crop position and flip of a pair are
drawn once on the host (reference get_params, data/base_dataset.py:63-78) and shipped as a [B,4] int32 tensor; crop,
flip, and Normalize((0.5,)*3, (0.5,)*3) (reference :81-112) happen in the kernel.  Returns the reference's dict:
{'A','B','A_paths','B_paths'} (README.md:18-25, nemar_model.py:151-159).

`--preprocess` as in the reference's get_transform (data/base_dataset.py:81-113):
  resize_and_crop       resize to load_size x load_size, random crop_size x crop_size crop
  crop                  random crop_size x crop_size crop of the image as it is
  scale_width           resize to width load_size (height follows), whole image
  scale_width_and_crop  ... then a random crop_size x crop_size crop
  none                  the whole image, sides rounded to multiples of 4 — the way to feed NON-SQUARE inputs such as the reference's
                        default --img_height 288 --img_width 384
The resize is a function of the image alone (the reference recomputes the same bicubic resize every epoch), so it is applied ONCE,
when the pool is brought into HBM (bicubic, antialiased, clamped to [0, 1] — the reference's PIL resize works on clamped 8-bit
values); per batch only the crop / flip / normalise launch runs."""
import ctypes
import os
import random

import numpy as np
import torch

from .. import ops
from .base_dataset import BaseDataset, get_params


class GpuPairsDataset(BaseDataset):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument('--pool_size_pairs', type=int, default=64, help='synthetic pool: number of A/B pairs')
        parser.add_argument('--data_seed', type=int, default=1234)
        return parser

    def __init__(self, opt):
        BaseDataset.__init__(self, opt)
        self.device = torch.device('cuda', opt.gpu_ids[0]) if opt.gpu_ids else torch.device('cuda')
        # crop positions / flips: one stream per rank (every rank augments its own samples)
        self.rng = random.Random(getattr(opt, 'data_seed', 1234) + 7919 * int(getattr(opt, 'shard_rank', 0)))
        self.pre = getattr(opt, 'preprocess', 'resize_and_crop')
        if self.pre not in ('resize_and_crop', 'crop', 'scale_width', 'scale_width_and_crop', 'none'):
            raise ValueError('--preprocess %s is not one of the reference\'s modes' % self.pre)
        if self.root == 'synthetic':
            m = int(getattr(opt, 'pool_size_pairs', 64))
            g = torch.Generator(device=self.device).manual_seed(getattr(opt, 'data_seed', 1234))
            if 'crop' in self.pre:            # images a little larger than the crop, at the load size already
                size = max(opt.crop_size, getattr(opt, 'load_size', opt.crop_size))
                shape = (m, 3, size, size)
            else:                             # whole images of the network's input size
                shape = (m, 3, opt.img_height, opt.img_width)
            self.pool_A = torch.rand(*shape, device=self.device, generator=g)
            self.pool_B = torch.rand(*shape, device=self.device, generator=g)
            self.paths_A = ['synthetic/A/%05d' % i for i in range(m)]
            self.paths_B = ['synthetic/B/%05d' % i for i in range(m)]
        else:
            self.pool_A, self.paths_A = self._load(os.path.join(self.root, 'A.npy'))
            self.pool_B, self.paths_B = self._load(os.path.join(self.root, 'B.npy'))
            assert self.pool_A.shape == self.pool_B.shape, "aligned pairs: A.npy and B.npy must have the same shape"
        self.pool_A, self.pool_B = self._resize(self.pool_A), self._resize(self.pool_B)
        self.M, _, self.H, self.W = self.pool_A.shape
        # what a batch looks like: a square crop, or the whole image
        self.out_hw = (opt.crop_size, opt.crop_size) if 'crop' in self.pre else (self.H, self.W)
        assert self.H >= self.out_hw[0] and self.W >= self.out_hw[1], "images smaller than the crop"

    def _resize(self, pool):
        """the image-only part of get_transform: resize / scale_width / make_power_2 (reference data/base_dataset.py:86-99), once"""
        opt, (h, w) = self.opt, pool.shape[2:]
        if 'resize' in self.pre:
            nh, nw = opt.load_size, opt.load_size
        elif 'scale_width' in self.pre:
            nw, nh = opt.load_size, int(opt.load_size * h / w)
        elif self.pre == 'none':
            nh, nw = int(round(h / 4) * 4), int(round(w / 4) * 4)
        else:
            nh, nw = h, w
        if (nh, nw) == (h, w):
            return pool
        out = torch.nn.functional.interpolate(pool, size=(nh, nw), mode='bicubic', align_corners=False, antialias=True)
        return out.clamp_(0.0, 1.0).contiguous()

    def _load(self, path):
        a = np.load(path)
        if a.ndim == 4 and a.shape[-1] == 3:
            a = a.transpose(0, 3, 1, 2)
        scale = 1.0 / 255.0 if a.dtype == np.uint8 else 1.0
        t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device, torch.float32) * scale
        return t.contiguous(), ['%s[%d]' % (path, i) for i in range(t.shape[0])]

    def __len__(self):
        return self.M

    def batch(self, indices):
        opt = self.opt
        hc, wc = self.out_hw
        params = np.zeros((len(indices), 4), dtype=np.int32)
        for b, i in enumerate(indices):
            p = get_params(opt, (self.W, self.H), self.rng)
            x0, y0 = p['crop_pos'] if 'crop' in self.pre else (0, 0)
            params[b] = (i % self.M, y0, x0, int(p['flip']))
        d_params = torch.from_numpy(params).to(self.device, non_blocking=True)
        out = {}
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for key, pool in (('A', self.pool_A), ('B', self.pool_B)):
            y = torch.empty((len(indices), 3, hc, wc), dtype=torch.float32, device=self.device)
            ops.L.crop_flip_normalize(ctypes.c_void_p(pool.data_ptr()), ctypes.c_void_p(d_params.data_ptr()),
                                      ctypes.c_void_p(y.data_ptr()), self.M, len(indices), 3, self.H, self.W, hc, wc, 1.0, st)
            out[key] = y
        out['A_paths'] = [self.paths_A[i % self.M] for i in indices]
        out['B_paths'] = [self.paths_B[i % self.M] for i in indices]
        self._last_params = params
        return out

    def __getitem__(self, index):
        b = self.batch([index])
        return {'A': b['A'][0], 'B': b['B'][0], 'A_paths': b['A_paths'][0], 'B_paths': b['B_paths'][0]}
