"""`--dataset_mode gpupairs`: aligned A/B image pairs held in HBM, augmented on the GPU.

The pool is [M,3,H,W] per modality with values in [0,1] (what ToTensor produces): `--dataroot synthetic` builds it from
the seeded generator used by the tests, any other dataroot is a directory with `A.npy` / `B.npy` ([M,3,H,W] or [M,H,W,3],
uint8 or float).  A batch is ONE launch of nemar_crop_flip_normalize per modality: the crop position and flip of a pair are
drawn once on the host (reference get_params, data/base_dataset.py:63-78) and shipped as a [B,4] int32 tensor; crop,
flip, and Normalize((0.5,)*3, (0.5,)*3) (reference :81-112) happen in the kernel.  Returns the reference's dict:
{'A','B','A_paths','B_paths'} (README.md:18-25, nemar_model.py:151-159)."""
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
        size = max(opt.crop_size, getattr(opt, 'load_size', opt.crop_size))
        if self.root == 'synthetic':
            m = int(getattr(opt, 'pool_size_pairs', 64))
            g = torch.Generator(device=self.device).manual_seed(getattr(opt, 'data_seed', 1234))
            self.pool_A = torch.rand(m, 3, size, size, device=self.device, generator=g)
            self.pool_B = torch.rand(m, 3, size, size, device=self.device, generator=g)
            self.paths_A = ['synthetic/A/%05d' % i for i in range(m)]
            self.paths_B = ['synthetic/B/%05d' % i for i in range(m)]
        else:
            self.pool_A, self.paths_A = self._load(os.path.join(self.root, 'A.npy'))
            self.pool_B, self.paths_B = self._load(os.path.join(self.root, 'B.npy'))
            assert self.pool_A.shape == self.pool_B.shape, "aligned pairs: A.npy and B.npy must have the same shape"
        self.M, _, self.H, self.W = self.pool_A.shape
        assert self.H >= opt.crop_size and self.W >= opt.crop_size

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
        cs = opt.crop_size
        params = np.zeros((len(indices), 4), dtype=np.int32)
        for b, i in enumerate(indices):
            p = get_params(opt, (self.W, self.H), self.rng)
            params[b] = (i % self.M, p['crop_pos'][1], p['crop_pos'][0], int(p['flip']))
        d_params = torch.from_numpy(params).to(self.device, non_blocking=True)
        out = {}
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for key, pool in (('A', self.pool_A), ('B', self.pool_B)):
            y = torch.empty((len(indices), 3, cs, cs), dtype=torch.float32, device=self.device)
            ops.L.crop_flip_normalize(ctypes.c_void_p(pool.data_ptr()), ctypes.c_void_p(d_params.data_ptr()),
                                      ctypes.c_void_p(y.data_ptr()), self.M, len(indices), 3, self.H, self.W, cs, cs, 1.0, st)
            out[key] = y
        out['A_paths'] = [self.paths_A[i % self.M] for i in indices]
        out['B_paths'] = [self.paths_B[i % self.M] for i in indices]
        self._last_params = params
        return out

    def __getitem__(self, index):
        b = self.batch([index])
        return {'A': b['A'][0], 'B': b['B'][0], 'A_paths': b['A_paths'][0], 'B_paths': b['B_paths'][0]}
