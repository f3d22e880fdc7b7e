"""Abstract dataset — same contract as reference data/base_dataset.py:13-60: __init__(opt) keeps `opt` and `root`,
modify_commandline_options(parser, is_train), __len__, __getitem__ -> dict with the data and its metadata."""
from abc import ABC, abstractmethod

import torch.utils.data as data


class BaseDataset(data.Dataset, ABC):
    def __init__(self, opt):
        self.opt = opt
        self.root = opt.dataroot

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    @abstractmethod
    def __len__(self):
        return 0

    @abstractmethod
    def __getitem__(self, index):
        pass


def get_params(opt, size, rng):
    """One crop position and flip decision per A/B pair (reference data/base_dataset.py:63-78, for pools that are already at
    load size): x in [0, W - crop], y in [0, H - crop], flip with probability 1/2."""
    w, h = size
    x = rng.randint(0, max(0, w - opt.crop_size))
    y = rng.randint(0, max(0, h - opt.crop_size))
    flip = (not opt.no_flip) and rng.random() > 0.5
    return {'crop_pos': (x, y), 'flip': flip}
