"""Command-line surface — the flags of reference options/base_options.py:20-60 and options/train_options.py:10-54
(plus the model's and STN's own flags, collected through modify_commandline_options exactly as
base_options.py:62-88 does), with the same names and defaults so existing command lines keep working.
Flags that only configure out-of-scope subsystems (visdom/HTML display, dataset loading) are accepted and ignored.
"""
import argparse
import os

from . import models


class BaseOptions:
    def __init__(self):
        self.initialized = False
        self.isTrain = False

    def initialize(self, parser):
        a = parser.add_argument
        a('--dataroot', default='synthetic', help='path to images; "synthetic" generates U[-1,1) A/B pairs on device')
        a('--name', type=str, default='experiment_name')
        a('--gpu_ids', type=str, default='0', help='gpu ids: e.g. 0  0,1,2 (one process drives the first id)')
        a('--checkpoints_dir', type=str, default='./checkpoints')
        a('--model', type=str, default='nemar')
        a('--input_nc', type=int, default=3)
        a('--output_nc', type=int, default=3)
        a('--ngf', type=int, default=64)
        a('--ndf', type=int, default=64)
        a('--netD', type=str, default='basic')
        a('--netG', type=str, default='resnet_9blocks')
        a('--n_layers_D', type=int, default=3)
        a('--norm', type=str, default='instance')
        a('--init_type', type=str, default='normal')
        a('--init_gain', type=float, default=0.02)
        a('--no_dropout', action='store_true')
        a('--dataset_mode', type=str, default='unaligned')
        a('--direction', type=str, default='AtoB')
        a('--serial_batches', action='store_true')
        a('--num_threads', default=4, type=int)
        a('--batch_size', type=int, default=1)
        a('--load_size', type=int, default=286)
        a('--img_height', type=int, default=288)
        a('--img_width', type=int, default=384)
        a('--crop_size', type=int, default=256)
        a('--max_dataset_size', type=int, default=float("inf"))
        a('--preprocess', type=str, default='resize_and_crop')
        a('--no_flip', action='store_true')
        a('--display_winsize', type=int, default=256)
        a('--epoch', type=str, default='latest')
        a('--load_iter', type=int, default=0)
        a('--verbose', action='store_true')
        a('--suffix', default='', type=str)
        self.initialized = True
        return parser

    def gather_options(self, argv=None):
        parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        parser = self.initialize(parser)
        opt, _ = parser.parse_known_args(argv)
        parser = models.get_option_setter(opt.model)(parser, self.isTrain)
        # dataset-specific flags (reference options/base_options.py:80-82); the modes of the reference itself ('unaligned',
        # ...) are not part of this build and add none
        opt, _ = parser.parse_known_args(argv)
        try:
            from . import data
            parser = data.get_option_setter(opt.dataset_mode)(parser, self.isTrain)
        except ModuleNotFoundError:
            pass
        self.parser = parser
        return parser.parse_args(argv)

    def print_options(self, opt):
        lines = ['----------------- Options ---------------']
        for k, v in sorted(vars(opt).items()):
            default = self.parser.get_default(k)
            comment = '\t[default: %s]' % str(default) if v != default else ''
            lines.append('{:>25}: {:<30}{}'.format(str(k), str(v), comment))
        lines.append('----------------- End -------------------')
        message = '\n'.join(lines)
        print(message)
        expr_dir = os.path.join(opt.checkpoints_dir, opt.name)
        os.makedirs(expr_dir, exist_ok=True)
        with open(os.path.join(expr_dir, '{}_opt.txt'.format(opt.phase)), 'wt') as f:
            f.write(message + '\n')

    def parse(self, argv=None, quiet=False):
        opt = self.gather_options(argv)
        opt.isTrain = self.isTrain
        if opt.suffix:
            opt.name = opt.name + '_' + opt.suffix.format(**vars(opt))
        if not quiet:
            self.print_options(opt)
        opt.gpu_ids = [int(s) for s in opt.gpu_ids.split(',') if int(s) >= 0]
        self.opt = opt
        return opt


class TrainOptions(BaseOptions):
    def initialize(self, parser):
        parser = BaseOptions.initialize(self, parser)
        a = parser.add_argument
        for flag, typ, default in (('--display_freq', int, 400), ('--display_ncols', int, 4), ('--display_id', int, -1),
                                   ('--display_server', str, "http://localhost"), ('--display_env', str, 'main'),
                                   ('--display_port', int, 8097), ('--update_html_freq', int, 1000),
                                   ('--print_freq', int, 100), ('--save_latest_freq', int, 5000),
                                   ('--save_epoch_freq', int, 5), ('--epoch_count', int, 1), ('--niter', int, 100),
                                   ('--niter_decay', int, 100), ('--pool_size', int, 50), ('--lr_decay_iters', int, 50)):
            a(flag, type=typ, default=default)
        a('--no_html', action='store_true')
        a('--step_graph', action='store_true',
          help='(MI355X build) capture optimize_parameters() into a hipGraph after the first batch and replay it: for launch-bound '
               'shapes (small images, batch 1); single process, fixed batch shape, no --enable_tbvis')
        a('--seed', type=int, default=None,
          help='(MI355X build) seed torch + the dropout generator for a repeatable run; default: unseeded, as the reference')
        a('--save_by_iter', action='store_true')
        a('--continue_train', action='store_true')
        a('--phase', type=str, default='train')
        a('--beta1', type=float, default=0.5)
        a('--lr', type=float, default=0.0002)
        a('--gan_mode', type=str, default='vanilla')
        a('--lr_policy', type=str, default='linear')
        self.isTrain = True
        return parser


class TestOptions(BaseOptions):
    def initialize(self, parser):
        parser = BaseOptions.initialize(self, parser)
        a = parser.add_argument
        a('--ntest', type=int, default=float("inf"))
        a('--results_dir', type=str, default='./results/')
        a('--aspect_ratio', type=float, default=1.0)
        a('--phase', type=str, default='test')
        a('--eval', action='store_true')
        a('--num_test', type=int, default=50)
        self.isTrain = False
        return parser
