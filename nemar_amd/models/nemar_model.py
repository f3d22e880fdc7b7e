"""NeMAR training step on MI355X — drop-in for reference models/nemar_model.py (NEMARModel :12-288).

Same flags (:30-45), same attributes (`netT`, `netR`, `netD`, `netD_multiresolution`, `loss_*`, `visual_names`,
`model_names`, `optimizers`), same call order inside optimize_parameters (:266-288): forward, discriminator step,
then translation+registration step against the freshly updated discriminator.  What differs is how the step
executes:
  * every operator is a gfx950 kernel (nemar_amd.ops); concatenations (real_A, X) are two-pointer conv inputs;
  * each loss term leaves its kernel already multiplied by its lambda, and the step back-propagates from the list
    of terms at once (`torch.autograd.backward(terms)` == backward of their sum), so no scalar arithmetic kernels
    run; the reference's `loss_*` scalars are formed lazily, only when somebody reads them;
  * the three Adam optimizers are single fused launches over flat buffers (ops.FlatAdam), whose gradient buffers are
    also the data-parallel all-reduce buckets (nemar_amd.distributed): one bucket after backward_D, two after
    backward_T_and_R.
"""
import itertools
import os

import torch

from .. import distributed as dist
from .. import ops
from . import networks
from . import stn
from .base_model import BaseModel


class _LazyLoss:
    """sum_i scale_i * term_i, evaluated on first use (float(), .item(), arithmetic via .value())."""

    def __init__(self, parts):
        self.parts = parts            # list of (0-dim tensor, python scale)
        self._v = None

    def value(self):
        if self._v is None:
            with torch.no_grad():
                v = None
                for t, s in self.parts:
                    x = t.detach() * s if s != 1.0 else t.detach()
                    v = x if v is None else v + x
                self._v = v if v is not None else torch.zeros((), device='cuda')
        return self._v

    def __float__(self):
        return float(self.value())

    def item(self):
        return float(self)

    def detach(self):
        return self.value()

    def __repr__(self):
        return 'LazyLoss(%g)' % float(self)


def _loss_property(name):
    """`model.loss_<name>` is a 0-dim TENSOR, as in the reference (models/nemar_model.py:179-261) — arithmetic, .item(),
    SummaryWriter.add_scalar all work — but it is only FORMED when somebody reads it: the step itself back-propagates from the
    list of weighted terms and runs no scalar arithmetic kernels."""
    key = '_lazy_loss_' + name

    def get(self):
        try:
            return self.__dict__[key].value()
        except KeyError:        # as a plain attribute of the reference would: hasattr / getattr(default) keep working
            raise AttributeError('loss_%s has not been computed yet (no optimize_parameters() call so far)' % name) from None

    def put(self, parts):
        self.__dict__[key] = parts if isinstance(parts, _LazyLoss) else _LazyLoss([(parts, 1.0)])
    return property(get, put)


class NEMARModel(BaseModel):
    """netT: translation A->B; netR: registration (STN) A~>B; netD: PatchGAN on (A, B) pairs."""

    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        parser = stn.modify_commandline_options(parser, is_train)
        if is_train:
            parser.add_argument('--lambda_GAN', type=float, default=1.0, help='Weight for the GAN loss.')
            parser.add_argument('--lambda_recon', type=float, default=100.0,
                                help='Weight for the L1 reconstruction loss.')
            parser.add_argument('--lambda_smooth', type=float, default=0.0, help='Regularization term used by the STN')
            parser.add_argument('--enable_tbvis', action='store_true',
                                help='Enable tensorboard visualizer (default : False)')
            parser.add_argument('--multi_resolution', type=int, default=1,
                                help='Use of multi-resolution discriminator.'
                                     '(if equals to 1 then no multi-resolution training is applied)')
            # flags of the reference's TensorboardVisualizer (util/tb_visualizer.py:6-15), accepted for CLI parity
            parser.add_argument('--tbvis_iteration_update_rate', type=int, default=1000)
            parser.add_argument('--tbvis_disable_report_weights', action='store_true')
            parser.add_argument('--tbvis_disable_report_offsets', action='store_true')
        return parser

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        self.train_stn = True
        self.setup_visualizers()
        self.tb_visualizer = None
        # T's two applications and D's 3 + 2 applications per step run as single batches (valid without cross-sample ops, i.e. not
        # with BatchNorm): the same graph with half the launches.  Round 1 measured no gain (78.15 vs 78.00 ms/step: every layer
        # already filled the chip at batch 8); with the split-16 kernels, whose max / split / slab-sum passes are paid per launch,
        # it is 47.9 vs 50.2 ms/step on the same box, so it is the default.  NEMAR_BATCHED_PASSES=0 restores the reference's call
        # order (tests/test_step_gpu.py compares the two).
        self._batched = opt.norm != 'batch' and os.environ.get('NEMAR_BATCHED_PASSES', '1') == '1'
        # dropout masks: one Philox stream per process, governed by torch.manual_seed() and different on every rank
        ops.manual_seed(torch.initial_seed() + dist.rank())
        self.define_networks()
        if self.isTrain:
            self.criterionGAN = networks.GANLoss(opt.gan_mode)
            self.criterionL1 = ops.l1_loss
            self.setup_optimizers()
            if getattr(opt, 'enable_tbvis', False):
                # scalars the reference's TensorboardVisualizer reports (util/tb_visualizer.py:53-92): losses and the mean
                # deformation offsets, the latter reduced on the device (nemar_amd/util/visualizer.py)
                from ..util.visualizer import TrainingMonitor
                self.tb_visualizer = TrainingMonitor(self, opt)
            self._one = torch.ones((), dtype=torch.float32, device=self.device)
            self._lam_smooth = torch.full((), float(opt.lambda_smooth), dtype=torch.float32, device=self.device)

    loss_L1_TR = _loss_property('L1_TR')
    loss_GAN_TR = _loss_property('GAN_TR')
    loss_L1_RT = _loss_property('L1_RT')
    loss_GAN_RT = _loss_property('GAN_RT')
    loss_smoothness = _loss_property('smoothness')
    loss_D_fake_TR = _loss_property('D_fake_TR')
    loss_D_fake_RT = _loss_property('D_fake_RT')
    loss_D = _loss_property('D')

    def setup_visualizers(self):
        # <loss>_TR: registration-first branch T(R(a)); <loss>_RT: translation-first branch R(T(a))
        self.loss_names = ['L1_TR', 'GAN_TR', 'L1_RT', 'GAN_RT', 'smoothness', 'D_fake_TR', 'D_fake_RT', 'D']
        self.visual_names = ['real_A', 'real_B', 'fake_TR_B', 'fake_RT_B', 'registered_real_A', 'fake_B']
        self.model_names = ['T', 'R'] + (['D'] if self.isTrain else [])

    def define_networks(self):
        opt = self.opt
        AtoB = opt.direction == 'AtoB'
        in_c = opt.input_nc if AtoB else opt.output_nc
        out_c = opt.output_nc if AtoB else opt.input_nc
        self.netT = networks.define_G(in_c, out_c, opt.ngf, opt.netG, opt.norm, not opt.no_dropout, opt.init_type,
                                      opt.init_gain, self.gpu_ids)
        self.netR = stn.define_stn(self.opt, self.opt.stn_type)
        if self.isTrain:
            self.netD = networks.define_D(opt.output_nc + opt.input_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm,
                                          opt.init_type, opt.init_gain, self.gpu_ids)
            # extra discriminators on 1/2, 1/4, ... resolution inputs (reference :106-113)
            self.netD_multiresolution = []
            for _ in range(max(0, opt.multi_resolution - 1)):
                self.netD_multiresolution.append(
                    networks.define_D(opt.output_nc + opt.input_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm,
                                      opt.init_type, opt.init_gain, self.gpu_ids))

    def reset_weights(self):
        opt = self.opt
        networks.init_weights(self.netT, opt.init_type, opt.init_gain)
        networks.init_weights(self.netD, opt.init_type, opt.init_gain)
        for netD_S in self.netD_multiresolution:
            networks.init_weights(netD_S, opt.init_type, opt.init_gain)

    def setup_optimizers(self):
        opt = self.opt
        betas = (opt.beta1, 0.999)
        self.optimizer_R = ops.FlatAdam(self.netR.parameters(), lr=opt.lr, betas=betas)
        self.optimizer_T = ops.FlatAdam(self.netT.parameters(), lr=opt.lr, betas=betas)
        d_params = itertools.chain(self.netD.parameters(), *[x.parameters() for x in self.netD_multiresolution])
        self.optimizer_D = ops.FlatAdam(d_params, lr=opt.lr, betas=betas)
        self.optimizers += [self.optimizer_T, self.optimizer_D, self.optimizer_R]
        # identical replicas on every rank before the first step; bucketed gradient averaging overlapped with backward
        dist.broadcast_parameters(self.optimizers)
        self.sync_T, self.sync_D, self.sync_R = dist.grad_sync_for([self.optimizer_T, self.optimizer_D, self.optimizer_R])

    def set_input(self, input):
        AtoB = self.opt.direction == 'AtoB'
        a, b = ('A', 'B') if AtoB else ('B', 'A')
        self.real_A = input[a].to(self.device, dtype=torch.float32, non_blocking=True).contiguous()
        self.real_B = input[b].to(self.device, dtype=torch.float32, non_blocking=True).contiguous()
        self.image_paths = input[a + '_paths']
        self._resized = {}

    # ---- forward -------------------------------------------------------------------------------------------
    def forward(self):
        if not self._batched:
            # the reference's own order (models/nemar_model.py:161-176)
            self.fake_B = self.netT(self.real_A)
            warped, reg_term = self.netR(self.real_A, self.real_B, apply_on=[self.real_A, self.fake_B])
            self.stn_reg_term = reg_term
            self.registered_real_A = warped[0]
            self.fake_TR_B = self.netT(self.registered_real_A)     # registration first, then translation
            self.fake_RT_B = warped[1]                             # translation first, then registration
        else:
            # Same graph, evaluated with T's two applications as ONE batch: the deformation depends on (a, b) only, so
            # R(a) is available before T runs, and T (per-sample InstanceNorm, no cross-sample op) maps [a ; R(a)] to
            # [T(a) ; T(R(a))].  Every T layer then launches once on 2x the pixels (full second round of workgroups,
            # one weight-gradient reduction instead of two).
            n = self.real_A.size(0)
            # (the field has three consumers — two warps and the regulariser —, the batch of T two sources and two slices: the library's own
            # nodes for fan-out, concatenation and slicing, ops.fork / cat_batch / split_batch, instead of autograd's ATen kernels)
            (f_a, f_b), f_reg = self.netR.fork_field(self.netR.predict(self.real_A, self.real_B), 2)
            self.registered_real_A = self.netR.warp(f_a, [self.real_A])[0]
            # (only the second half is differentiated: the stem's data gradient runs on it alone)
            both = self.netT(ops.grad_from(ops.cat_batch([self.real_A, self.registered_real_A]), n))
            self.fake_B, self.fake_TR_B = ops.split_batch(both, 2)
            self.fake_RT_B = self.netR.warp(f_b, [self.fake_B])[0]
            self.stn_reg_term = self.netR.regularization(f_reg, self.registered_real_A)
        self._resized = {}
        if self.tb_visualizer is not None:
            # the reference runs netR a second time here (get_grid, :172-173); the field of the pass above is the same tensor
            self.deformation_field_A_to_B = getattr(self.netR, 'last_offsets', None)

    def _half(self, name, tensor, level):
        """tensor bilinearly resized to 1/2^level resolution (reference :185-188 etc.).  Only the step's INPUTS (real_A,
        real_B: `name` given) are cached — they are resized once instead of once per discriminator pass and the cache dies
        with set_input(); generated images are resized where they are used."""
        sh, sw = self.real_A.size(2) // (2 ** level), self.real_A.size(3) // (2 ** level)
        if name is None:
            return ops.resize_bilinear(tensor, sh, sw)
        key = (name, level)
        if key not in self._resized:
            self._resized[key] = ops.resize_bilinear(tensor, sh, sw)
        return self._resized[key]

    def _d_terms_batched(self, specs):
        """_d_terms for several (image, image_name, target_is_real, weight, detach) at once: the discriminators see the
        images as one batch (no cross-sample op in D), the loss terms are taken on the per-image slices."""
        k, n = len(specs), self.real_A.size(0)
        imgs = [(im.detach() if det else im) for (im, _, _, _, det) in specs]
        # an image every discriminator reads: one handle per reader (ops.fork: the readers' gradients are added by the library's kernel)
        readers = 1 + len(self.netD_multiresolution)
        handles = [ops.fork(im, readers) for im in imgs]
        a_rep = ops.cat_batch([self.real_A] * k)
        out = ops.split_batch(self.netD(a_rep, ops.cat_batch([h[0] for h in handles])), k)
        terms = [[self.criterionGAN(out[i], tr, w)] for i, (_, _, tr, w, _) in enumerate(specs)]
        for lvl, netD_S in enumerate(self.netD_multiresolution):
            a_r = self._half('real_A', self.real_A, lvl + 1)
            img_r = [self._half(name, h[lvl + 1], lvl + 1) for h, (_, name, _, _, det) in zip(handles, specs)]
            out = ops.split_batch(netD_S(ops.cat_batch([a_r] * k), ops.cat_batch(img_r)), k)
            for i, (_, _, tr, w, _) in enumerate(specs):
                terms[i].append(self.criterionGAN(out[i], tr, w))
        return terms

    def _d_terms(self, image, image_name, target_is_real, weight, detach):
        """[weight * GANLoss(D_i(real_A_i, image_i), target)] over the full-resolution discriminator and every
        reduced-resolution one."""
        img = image.detach() if detach else image
        terms = [self.criterionGAN(self.netD(self.real_A, img), target_is_real, weight)]
        for i, netD_S in enumerate(self.netD_multiresolution):
            a_r = self._half('real_A', self.real_A, i + 1)
            img_r = self._half(image_name, img, i + 1)
            terms.append(self.criterionGAN(netD_S(a_r, img_r), target_is_real, weight))
        return terms

    # ---- discriminator step ---------------------------------------------------------------------------------
    def backward_D(self):
        w = 0.5 * self.opt.lambda_GAN
        if self._batched:
            real, fake_tr, fake_rt = self._d_terms_batched([(self.real_B, 'real_B', True, w, True),
                                                            (self.fake_TR_B, None, False, w, True),
                                                            (self.fake_RT_B, None, False, w, True)])
        else:
            real = self._d_terms(self.real_B, 'real_B', True, w, detach=True)
            fake_tr = self._d_terms(self.fake_TR_B, None, False, w, detach=True)
            fake_rt = self._d_terms(self.fake_RT_B, None, False, w, detach=True)
        inv = 1.0 / w if w != 0 else 0.0
        self.loss_D_fake_TR = _LazyLoss([(t, inv) for t in fake_tr])
        self.loss_D_fake_RT = _LazyLoss([(t, inv) for t in fake_rt])
        terms = real + fake_tr + fake_rt
        self.loss_D = _LazyLoss([(t, 1.0) for t in terms])
        torch.autograd.backward(terms, [self._one] * len(terms))
        return self.__dict__['_lazy_loss_D']

    # ---- translation + registration step ----------------------------------------------------------------------
    def backward_T_and_R(self):
        opt = self.opt
        # each generated image is read by its L1 term and by the discriminators: two handles (ops.fork)
        tr_l1, tr_d = ops.fork(self.fake_TR_B, 2)
        rt_l1, rt_d = ops.fork(self.fake_RT_B, 2)
        l1_tr = self.criterionL1(tr_l1, self.real_B, opt.lambda_recon)
        l1_rt = self.criterionL1(rt_l1, self.real_B, opt.lambda_recon)
        if self._batched:
            gan_tr, gan_rt = self._d_terms_batched([(tr_d, None, True, opt.lambda_GAN, False),
                                                    (rt_d, None, True, opt.lambda_GAN, False)])
        else:
            gan_tr = self._d_terms(tr_d, None, True, opt.lambda_GAN, detach=False)
            gan_rt = self._d_terms(rt_d, None, True, opt.lambda_GAN, detach=False)
        self.loss_L1_TR = _LazyLoss([(l1_tr, 1.0)])
        self.loss_GAN_TR = _LazyLoss([(t, 1.0) for t in gan_tr])
        self.loss_L1_RT = _LazyLoss([(l1_rt, 1.0)])
        self.loss_GAN_RT = _LazyLoss([(t, 1.0) for t in gan_rt])
        self.loss_smoothness = _LazyLoss([(self.stn_reg_term, float(opt.lambda_smooth))])
        roots = [l1_tr, l1_rt] + gan_tr + gan_rt
        grads = [self._one] * len(roots)
        if opt.lambda_smooth != 0.0:
            roots.append(self.stn_reg_term)         # d(lambda * reg) = lambda * d(reg): the weight is the seed
            grads.append(self._lam_smooth)
        torch.autograd.backward(roots, grads)
        return _LazyLoss([(r, 1.0) for r in roots[:len(roots) - (1 if opt.lambda_smooth != 0.0 else 0)]] +
                         [(self.stn_reg_term, float(opt.lambda_smooth))])

    # ---- the step as a captured hipGraph (launch-bound small configurations: BASELINE config 1) ---------------------------------
    def enable_step_graph(self, warmup=3):
        """Capture optimize_parameters() for the CURRENT input shapes into a hipGraph and replay it from then on: one graph launch
        instead of ~1100 kernel launches per step.  Single process only (the gradient all-reduce is not captured).  Inputs are copied
        into static buffers; dropout offsets and Adam's step-dependent scalars live in device memory (ops.step_params), so every
        replay is a fresh step; an eager run in that mode produces the same bits (tests/test_step_gpu.py)."""
        if dist.is_distributed():
            raise RuntimeError('enable_step_graph: single-process only')
        if self.tb_visualizer is not None:
            raise RuntimeError('enable_step_graph: disable the tensorboard visualiser (it reads tensors between launches)')
        ops.step_params(True, self.device)
        self._static_A, self._static_B = self.real_A.clone(), self.real_B.clone()
        opts = [self.optimizer_D, self.optimizer_R, self.optimizer_T]
        # The warm-up steps must not train: they run with lr = 0 (Adam leaves every parameter bit-unchanged: p - 0 * x) and the
        # moments, step counts and the dropout step counter are put back afterwards — the first replay is then step 1 of the same
        # schedule an eager loop would follow, on untouched weights.  (The packed-weight images are rebuilt from the same values, so
        # the state the capture sees — T and R to be re-packed, D's forward images valid — is the steady state of every later step.)
        saved = [(o.m.clone(), o.v.clone(), o.step_count, o.param_groups[0]["lr"]) for o in opts]
        saved_step = ops._step_params["step"]
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                         # warm-up on a side stream (allocator pools, lazy attributes, packs)
            try:
                for o in opts:
                    o.param_groups[0]["lr"] = 0.0
                for _ in range(warmup):
                    self.real_A, self.real_B = self._static_A, self._static_B
                    ops.begin_step()
                    self._optimize_parameters_eager()
            finally:
                for o, (m, v, n, lr) in zip(opts, saved):
                    o.m.copy_(m)
                    o.v.copy_(v)
                    o.step_count = n
                    o.param_groups[0]["lr"] = lr
                ops._step_params["step"] = saved_step
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        self.real_A, self.real_B = self._static_A, self._static_B
        ops.begin_step()
        for o in opts:
            o.prepare_step(o.step_count + 1)
        ops._step_params["capturing"] = True
        try:
            with torch.cuda.graph(self._graph):
                self._optimize_parameters_eager()
        finally:
            ops._step_params["capturing"] = False
        # (the capture itself executed nothing: the step it describes is run by the first replay; undo its host book-keeping)
        for o in opts:
            o.step_count -= 1
        ops._step_params["step"] -= 1
        self._graph_losses = {k: v for k, v in self.__dict__.items() if k.startswith('_lazy_loss_')}
        ops.pin_workspaces(True)          # the graph holds their addresses: from now on they may grow but are never freed

    def _replay_step(self):
        if self.real_A.shape != self._static_A.shape or self.real_B.shape != self._static_B.shape:
            # a batch of another shape (the ragged last batch of an epoch): the captured launches do not describe it — run this step
            # eagerly, in the same device-parameter mode (bit-identical arithmetic), and keep the graph for the next full batch
            ops.begin_step()
            return self._optimize_parameters_eager()
        if self.real_A is not self._static_A:
            self._static_A.copy_(self.real_A)
            self._static_B.copy_(self.real_B)
            self.real_A, self.real_B = self._static_A, self._static_B
        ops.begin_step()
        for o in (self.optimizer_D, self.optimizer_R, self.optimizer_T):
            o.prepare_step(o.step_count + 1)
        self._graph.replay()
        for o in (self.optimizer_D, self.optimizer_R, self.optimizer_T):
            o.replayed_step()
        # the loss_* attributes were created during the capture: their term buffers hold THIS step's values now, any sum formed by an
        # earlier read is stale
        for k, v in self._graph_losses.items():
            v._v = None
            self.__dict__[k] = v          # (an eager step in between — a ragged batch — had replaced them)

    def optimize_parameters(self):
        if getattr(self, '_graph', None) is not None:
            return self._replay_step()
        ops.begin_step()
        return self._optimize_parameters_eager()

    def _optimize_parameters_eager(self):
        # data parallel: how many gradient contributions each parameter will receive is COUNTED while the forward passes run
        # (T: once batched / twice; every discriminator: once batched / three times) — GradSync launches a bucket's all-reduce when
        # its last contribution has been issued
        self.sync_T.count_uses()
        self.sync_R.count_uses()
        self.forward()
        # D step
        self.set_requires_grad([self.netT, self.netR], False)
        self.optimizer_D.zero_grad()
        self.sync_D.count_uses()
        self.sync_D.begin()
        self.backward_D()
        self.sync_D.finish()
        self.optimizer_D.step()
        self.set_requires_grad([self.netT, self.netR], True)
        # T + R step (sees the updated D)
        self.set_requires_grad([self.netD, *self.netD_multiresolution], False)
        self.optimizer_R.zero_grad()
        self.optimizer_T.zero_grad()
        self.sync_T.begin()
        self.sync_R.begin()
        self.backward_T_and_R()
        self.sync_R.finish()
        self.sync_T.finish()
        self.optimizer_R.step()
        self.optimizer_T.step()
        self.set_requires_grad([self.netD, *self.netD_multiresolution], True)
        if self.tb_visualizer is not None:
            self.tb_visualizer.iteration_step()
