"""Model plug-in base class — same surface as reference models/base_model.py:8-234 (what train.py calls), for
models whose networks run on the gfx950 kernel library.

Kept verbatim in meaning: the abstract step API, `loss_<name>` / `net<name>` attribute conventions, checkpoint file
names `'%s_net_%s.pth' % (epoch, name)` holding plain state_dicts with the reference's key names, and
set_requires_grad gating (which is what lets the kernels skip weight-gradient work for frozen networks).
One process drives one device: there is no nn.DataParallel wrapper to strip.
"""
import os
from abc import ABC, abstractmethod
from collections import OrderedDict

import torch

from . import networks


class BaseModel(ABC):
    def __init__(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.device = torch.device('cuda:{}'.format(self.gpu_ids[0])) if self.gpu_ids else torch.device('cpu')
        if self.device.type != 'cuda':
            raise RuntimeError('nemar_amd models run on an MI355X only (pass --gpu_ids 0); '
                               'there is no CPU execution path in the product')
        torch.cuda.set_device(self.device)
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names = []
        self.model_names = []
        self.visual_names = []
        self.optimizers = []
        self.image_paths = []
        self.metric = 0  # for the 'plateau' learning-rate policy

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    @abstractmethod
    def set_input(self, input):
        pass

    @abstractmethod
    def forward(self):
        pass

    @abstractmethod
    def optimize_parameters(self):
        pass

    def setup(self, opt):
        """Create schedulers, optionally load networks, print them (reference :78-89)."""
        if self.isTrain:
            self.schedulers = [networks.get_scheduler(optimizer, opt) for optimizer in self.optimizers]
        if not self.isTrain or opt.continue_train:
            load_suffix = 'iter_%d' % opt.load_iter if opt.load_iter > 0 else opt.epoch
            self.load_networks(load_suffix)
        self.print_networks(opt.verbose)

    def _nets(self):
        for name in self.model_names:
            if isinstance(name, str):
                yield name, getattr(self, 'net' + name)

    def eval(self):
        for _, net in self._nets():
            net.eval()

    def test(self):
        with torch.no_grad():
            self.forward()
            self.compute_visuals()

    def compute_visuals(self):
        pass

    def get_image_paths(self):
        return self.image_paths

    def update_learning_rate(self):
        for scheduler in self.schedulers:
            if self.opt.lr_policy == 'plateau':
                scheduler.step(self.metric)
            else:
                scheduler.step()
        lr = self.optimizers[0].param_groups[0]['lr']
        print('learning rate = %.7f' % lr)

    def get_current_visuals(self):
        visual_ret = OrderedDict()
        for name in self.visual_names:
            if isinstance(name, str):
                value = getattr(self, name)
                if isinstance(value, list):
                    for i, x in enumerate(value):
                        visual_ret['{}_{}'.format(name, i)] = x
                else:
                    visual_ret[name] = value
        return visual_ret

    def get_current_losses(self):
        """OrderedDict name -> float.  One device->host copy for all losses instead of one sync per loss."""
        names = [n for n in self.loss_names if isinstance(n, str)]
        vals = [getattr(self, 'loss_' + n) for n in names]
        tens = [v.detach().reshape(1).float() for v in vals if torch.is_tensor(v)]
        host = torch.cat(tens).cpu().tolist() if tens else []
        out, k = OrderedDict(), 0
        for n, v in zip(names, vals):
            if torch.is_tensor(v):
                out[n] = float(host[k])
                k += 1
            else:
                out[n] = float(v)
        return out

    def save_networks(self, epoch):
        """'%s_net_%s.pth' % (epoch, name): plain CPU state_dict with the reference's key names (reference :148-164)."""
        os.makedirs(self.save_dir, exist_ok=True)
        for name, net in self._nets():
            path = os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch, name))
            torch.save(OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items()), path)

    def load_networks(self, epoch):
        """Load '%s_net_%s.pth' (reference :180-203); tolerates `module.` prefixes and legacy InstanceNorm buffers."""
        for name, net in self._nets():
            path = os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch, name))
            print('loading the model from %s' % path)
            state_dict = torch.load(path, map_location='cpu')
            if hasattr(state_dict, '_metadata'):
                del state_dict._metadata
            clean = OrderedDict()
            for k, v in state_dict.items():
                if k.startswith('module.'):
                    k = k[len('module.'):]
                if k.endswith(('running_mean', 'running_var', 'num_batches_tracked')):
                    continue            # InstanceNorm here has no buffers (affine=False, no running stats)
                clean[k] = v
            net.load_state_dict(clean)

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name, net in self._nets():
            num_params = sum(p.numel() for p in net.parameters())
            if verbose:
                print(net)
            print('[Network %s] Total number of parameters : %.3f M' % (name, num_params / 1e6))
        print('-----------------------------------------------')

    def set_requires_grad(self, nets, requires_grad=False):
        if not isinstance(nets, list):
            nets = [nets]
        for net in nets:
            if net is not None:
                for param in net.parameters():
                    param.requires_grad = requires_grad
