"""Generate the golden fixtures in tests/golden/*.npz by IMPORTING THE REFERENCE (/root/reference) in the build
container and running its own code on seeded inputs/weights (tests/seeded.py).  The reference cannot travel to the
GPU box; only these small data files do.  Re-run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures are DATA (inputs are regenerated from seeds; expected outputs are stored): losses, regularisation terms,
image crops/means, offsets, per-parameter gradient norms and post-Adam parameter checksums of
NEMARModel.optimize_parameters() (reference models/nemar_model.py:266-288), plus op-level values of the reference's
own functions (smoothness_loss, UnetSTN identity warp, GANLoss).
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, '/root/reference')
tb = types.ModuleType('torch.utils.tensorboard')
tb.SummaryWriter = object
sys.modules['torch.utils.tensorboard'] = tb          # the only missing import on the model's import path

import torch  # noqa: E402

import seeded  # noqa: E402
from full_record import full_step_record  # noqa: E402
from step_configs import STEP_CONFIGS, FULL_CONFIGS, DEEP_STN_CFG, make_opt, hw  # noqa: E402


KEYS = {}


def load_seeded(net, seed, overrides):
    sd = net.state_dict()
    new = seeded.seeded_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed, overrides)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})


def crop(t):
    return t[:, :, :16, :16].detach().numpy().copy()


def run_step_config(name, cfg):
    from models.nemar_model import NEMARModel
    opt = make_opt(cfg)
    torch.manual_seed(0)
    m = NEMARModel(opt)
    m.setup(opt)
    load_seeded(m.netT, cfg['seed'] + 1, cfg.get('overrides_T'))
    load_seeded(m.netR, cfg['seed'] + 2, cfg.get('overrides_R'))
    load_seeded(m.netD, cfg['seed'] + 3, cfg.get('overrides_D'))
    for i, d in enumerate(m.netD_multiresolution):
        load_seeded(d, cfg['seed'] + 10 + i, cfg.get('overrides_D'))
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    out = {}
    KEYS[name] = {nm: [[k, list(v.shape)] for k, v in net.state_dict().items()]
                  for nm, net in (('T', m.netT), ('R', m.netR), ('D', m.netD))}
    for step in range(cfg.get('steps', 1)):
        m.set_input({'A': torch.from_numpy(A), 'B': torch.from_numpy(B), 'A_paths': ['a'], 'B_paths': ['b']})
        # gradients are captured by hooks-free inspection: run the pieces exactly as optimize_parameters does
        m.optimize_parameters()
        losses = m.get_current_losses()
        for k, v in losses.items():
            out['s%d/loss/%s' % (step, k)] = np.float64(v)
        out['s%d/reg' % step] = np.float64(float(m.stn_reg_term))
        for nm in ('fake_B', 'registered_real_A', 'fake_TR_B', 'fake_RT_B'):
            t = getattr(m, nm)
            out['s%d/crop/%s' % (step, nm)] = crop(t)
            out['s%d/mean/%s' % (step, nm)] = np.float64(t.double().mean().item())
            out['s%d/absmean/%s' % (step, nm)] = np.float64(t.double().abs().mean().item())
        for nm, net in (('T', m.netT), ('R', m.netR), ('D', m.netD)):
            for k, p in net.named_parameters():
                if p.grad is not None:
                    out['s%d/gradnorm/%s/%s' % (step, nm, k)] = np.float64(p.grad.double().norm().item())
                out['s%d/psum/%s/%s' % (step, nm, k)] = np.float64(p.detach().double().sum().item())
                out['s%d/pabs/%s/%s' % (step, nm, k)] = np.float64(p.detach().double().abs().sum().item())
    np.savez_compressed(os.path.join(HERE, 'step_%s.npz' % name), **out)
    print(name, {k: float(v) for k, v in out.items() if '/loss/' in k and k.startswith('s0')})


def run_full_config(name, cfg):
    """Full-width step of the reference in fp32 AND fp64 (tests/step_configs.FULL_CONFIGS)."""
    import time
    from models.nemar_model import NEMARModel
    import models.stn.unet_stn as ref_unet
    for key, val in DEEP_STN_CFG.items():            # the 'deep' cfg: new entries in the reference's own dicts
        getattr(ref_unet, key)['deep'] = val
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    out = {}
    # cfg['perturbed'] = n: n further fp32 runs of the reference on inputs moved by 4 ulps up / down — what a different summation order
    # does to the first layer's output.  The gradient through the discriminator's LeakyReLU masks (and with it everything of the
    # affine registration net, whose whole gradient is one 6-vector) moves by PERCENTS under such a change: one fp32-vs-fp64 gap
    # is a single draw of a heavy-tailed quantity, the test takes the largest of the n + 1 (tests/test_step_full_gpu.py compare()).
    runs = [('f32', torch.float32, 0), ('f64', torch.float64, 0)] + [('f32p%d' % (i + 1), torch.float32, (-1) ** i * 4) for i in range(cfg.get('perturbed', 0))]
    A0, B0 = A, B
    for tag, dt, ulps in runs:
        if tag == 'f64' and not cfg.get('f64', True):
            continue          # 1024x1024 in fp64 does not fit this container's memory/time: see tests/test_step_gpu.py
        A, B = A0, B0
        for _ in range(abs(ulps)):
            A = np.nextafter(A, np.float32(np.inf if ulps > 0 else -np.inf))
            B = np.nextafter(B, np.float32(np.inf if ulps > 0 else -np.inf))
        t0 = time.time()
        torch.set_default_dtype(dt)
        try:
            opt = make_opt(cfg)
            torch.manual_seed(0)
            m = NEMARModel(opt)
            m.setup(opt)
            load_seeded(m.netT, cfg['seed'] + 1, cfg.get('overrides_T'))
            load_seeded(m.netR, cfg['seed'] + 2, cfg.get('overrides_R'))
            load_seeded(m.netD, cfg['seed'] + 3, cfg.get('overrides_D'))
            for i, d in enumerate(m.netD_multiresolution):
                load_seeded(d, cfg['seed'] + 10 + i, cfg.get('overrides_D'))
            assert next(m.netT.parameters()).dtype == dt
            rec = full_step_record(m, A, B, cfg['seed'])
        finally:
            torch.set_default_dtype(torch.float32)
        for k, v in rec.items():
            if ulps and not k.startswith(('grad', 'psum', 'pabs')):
                continue      # (the perturbed runs calibrate the gradient / post-Adam rows only)
            out['%s/%s' % (tag, k)] = np.asarray(v, dtype=np.float64)
        print(name, tag, '%.1fs' % (time.time() - t0), {k: round(v, 6) for k, v in rec.items() if k.startswith('loss/')},
              flush=True)
        del m
    np.savez_compressed(os.path.join(HERE, 'step_%s.npz' % name), **out)


TRAJ_STEPS = 10


def run_trajectory(name='c2_full'):
    """TRAJ_STEPS consecutive steps of the reference (fp32 and fp64) at a full-width configuration: step_c2_traj10.npz.  The one-step
    fixtures start from seeded uniform weights; this one pins what the build does with Adam-moved weights, step after step
    (tests/full_record.traj_record; tests/test_step_full_gpu.py::test_ten_step_trajectory_vs_reference)."""
    import time
    from models.nemar_model import NEMARModel
    from full_record import traj_record
    cfg = FULL_CONFIGS[name]
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    out = {}
    # 'f32p1' / 'f32p2': fp32 runs on inputs moved by +-4 ulps.  A free-running GAN trajectory is chaotic (Adam's first steps move every
    # weight by +-lr: the sign of a gradient element at rounding distance of zero decides); how fast two CORRECT fp32 runs drift apart
    # is what these measure, and the test bounds the build's drift by the largest of the three.
    A0, B0 = A, B
    for tag, dt, ulps in (('f32', torch.float32, 0), ('f64', torch.float64, 0), ('f32p1', torch.float32, 4), ('f32p2', torch.float32, -4)):
        A, B = A0, B0
        for _ in range(abs(ulps)):
            A = np.nextafter(A, np.float32(np.inf if ulps > 0 else -np.inf))
            B = np.nextafter(B, np.float32(np.inf if ulps > 0 else -np.inf))
        t0 = time.time()
        torch.set_default_dtype(dt)
        try:
            opt = make_opt(cfg)
            torch.manual_seed(0)
            m = NEMARModel(opt)
            m.setup(opt)
            load_seeded(m.netT, cfg['seed'] + 1, cfg.get('overrides_T'))
            load_seeded(m.netR, cfg['seed'] + 2, cfg.get('overrides_R'))
            load_seeded(m.netD, cfg['seed'] + 3, cfg.get('overrides_D'))
            rec = traj_record(m, A, B, cfg['seed'], TRAJ_STEPS)
        finally:
            torch.set_default_dtype(torch.float32)
        for k, v in rec.items():
            out['%s/%s' % (tag, k)] = np.asarray(v, dtype=np.float64)
        print(name, 'trajectory', tag, '%.1fs' % (time.time() - t0), [round(rec['s%02d/loss/L1_TR' % s], 5) for s in range(TRAJ_STEPS)], flush=True)
        del m
    np.savez_compressed(os.path.join(HERE, 'step_c2_traj%d.npz' % TRAJ_STEPS), **out)


def run_registration_submodel(name='c5_full'):
    """fp32 AND fp64 runs of the reference's UnetSTN sub-model (tests/full_record.registration_record) at a full configuration's
    geometry — for BASELINE config 5 (1024x1024, 'deep' cfg) the fp64 truth the full step cannot have in this container."""
    import time
    from models import stn as ref_stn
    import models.stn.unet_stn as ref_unet
    from full_record import registration_record
    for key, val in DEEP_STN_CFG.items():
        getattr(ref_unet, key)['deep'] = val
    cfg = FULL_CONFIGS[name]
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    out = {}
    for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
        t0 = time.time()
        torch.set_default_dtype(dt)
        try:
            torch.manual_seed(0)
            net = ref_stn.define_stn(make_opt(cfg), 'unet')
            load_seeded(net, cfg['seed'] + 2, cfg.get('overrides_R'))
            l1 = lambda x, y, wgt: torch.nn.functional.l1_loss(x, y) * wgt
            rec = registration_record(net, l1, A, B, cfg['seed'], 100.0, cfg['lambda_smooth'])
        finally:
            torch.set_default_dtype(torch.float32)
        for k, v in rec.items():
            out['%s/%s' % (tag, k)] = np.asarray(v, dtype=np.float64)
        print(name, 'registration sub-model', tag, '%.1fs' % (time.time() - t0),
              {k: round(v, 6) for k, v in rec.items() if k in ('loss/recon', 'reg')}, flush=True)
        del net
    np.savez_compressed(os.path.join(HERE, 'regsub_%s.npz' % name), **out)


def run_op_fixtures():
    from models.stn.stn_losses import smoothness_loss
    from models.stn.unet_stn import UnetSTN
    from models.networks import GANLoss
    out = {}
    d = torch.from_numpy(seeded.uniform((2, 2, 9, 13), 5, 0) * 0.1)
    img = torch.from_numpy(seeded.uniform((2, 3, 9, 13), 5, 1))
    for al in (0.0, 1.7):
        dd = d.clone().requires_grad_(True)
        l = smoothness_loss(dd, img, alpha=al)
        l.backward()
        out['smooth/a%g/loss' % al] = np.float64(l.item())
        out['smooth/a%g/grad' % al] = dd.grad.numpy().copy()
    # UnetSTN identity-grid warp: the reference's "identity" is a slight zoom
    stn = UnetSTN(1, 1, 8, 12, 'A', 'normal', 0.0, True, 1)
    ident = stn.get_identity_grid()
    out['unet/identity_grid'] = ident.numpy().copy()
    row = torch.arange(12, dtype=torch.float32).view(1, 1, 1, 12).repeat(1, 1, 8, 1)
    warped = torch.nn.functional.grid_sample(row, ident.permute(0, 2, 3, 1), mode='bilinear', padding_mode='zeros',
                                             align_corners=False)
    out['unet/identity_warp_of_arange'] = warped.numpy().copy()
    lg = torch.from_numpy(seeded.uniform((2, 1, 6, 6), 9, 0) * 4)
    for mode in ('vanilla', 'lsgan', 'wgangp'):
        for real in (True, False):
            x = lg.clone().requires_grad_(True)
            l = GANLoss(mode)(x, real)
            l.backward()
            out['gan/%s/%d/loss' % (mode, real)] = np.float64(l.item())
            out['gan/%s/%d/grad' % (mode, real)] = x.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'ops_reference.npz'), **out)
    print('ops fixtures:', len(out))


def run_init_stats():
    """Per-tensor (numel, mean, std) of the reference's own initialisation: define_G / define_D through
    init_net (models/networks.py:62-113), UnetSTN / AffineSTN through their per-Conv initialisers incl. the quirks —
    decoder convs always kaiming (`init_fun` typo, models/stn/unet_stn.py:62-63), 'zeros' = N(0, 1e-5)
    (models/stn/layers.py:46-47), affine head N(0, 5e-4) (models/stn/affine_stn.py:75-76)."""
    import json
    from models import networks as ref_networks
    from models import stn as ref_stn
    from step_configs import FULL_CONFIGS
    out = {}
    torch.manual_seed(1234)
    nets = {
        'T': ref_networks.define_G(3, 3, 64, 'resnet_9blocks', 'instance', True, 'normal', 0.02, []),
        'D': ref_networks.define_D(6, 64, 'basic', 3, 'instance', 'normal', 0.02, []),
        'R_unet': ref_stn.define_stn(make_opt(FULL_CONFIGS['c2_full']), 'unet'),
        'R_unet_noident': ref_stn.define_stn(argparse.Namespace(**{**vars(make_opt(FULL_CONFIGS['c2_full'])),
                                                                    'stn_no_identity_init': True}), 'unet'),
        'R_affine': ref_stn.define_stn(make_opt(STEP_CONFIGS['affine128']), 'affine'),
    }
    for nm, net in nets.items():
        out[nm] = [[k, int(v.numel()), float(v.double().mean()), float(v.double().std()) if v.numel() > 1 else 0.0]
                   for k, v in net.state_dict().items()]
    with open(os.path.join(HERE, 'init_stats.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('init stats:', {k: len(v) for k, v in out.items()})


def run_unet_generator():
    """The reference's UnetGenerator (`--netG unet_128`, models/networks.py:449-553) forward + backward on seeded
    weights / input: full output, input gradient and every parameter gradient's L2 norm + a 64-element head."""
    import functools
    from models import networks as ref_networks
    torch.manual_seed(0)
    norm = functools.partial(torch.nn.InstanceNorm2d, affine=False, track_running_stats=False)
    out = {}
    for tag, use_dropout in (('plain', False),):
        net = ref_networks.UnetGenerator(3, 3, 7, ngf=4, norm_layer=norm, use_dropout=use_dropout)
        load_seeded(net, 77, None)
        x = torch.from_numpy(seeded.seeded_images(2, 3, 128, 128, 78)[0]).clone().requires_grad_(True)
        r = torch.from_numpy(seeded.seeded_images(2, 3, 128, 128, 79)[1])
        y = net(x)
        (y * r).sum().backward()
        out[tag + '/y'] = y.detach().numpy()
        out[tag + '/gx'] = x.grad.numpy()
        out[tag + '/keys'] = np.array([k for k, _ in net.state_dict().items()])
        out[tag + '/shapes'] = np.array([str(tuple(v.shape)) for _, v in net.state_dict().items()])
        for k, p in net.named_parameters():
            out[tag + '/gnorm/' + k] = np.array(float(p.grad.norm()))
            out[tag + '/ghead/' + k] = p.grad.reshape(-1)[:64].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'unet_generator.npz'), **out)
    print('unet generator fixture:', len(out), 'arrays')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    torch.set_num_threads(8)
    if a.only in (None, 'ops'):
        run_op_fixtures()
    if a.only in (None, 'init'):
        run_init_stats()
    if a.only in (None, 'unet_generator'):
        run_unet_generator()
    if a.only in (None, 'regsub', 'full'):
        run_registration_submodel('c5_full')
    if a.only in (None, 'traj', 'full'):
        run_trajectory('c2_full')
    for name, cfg in STEP_CONFIGS.items():
        if a.only in (None, name):
            run_step_config(name, cfg)
    for name, cfg in FULL_CONFIGS.items():
        if a.only in (None, name, 'full'):
            run_full_config(name, cfg)
    if a.only is None:
        import json
        with open(os.path.join(HERE, 'state_dict_keys.json'), 'w') as f:
            json.dump(KEYS, f, indent=0)
