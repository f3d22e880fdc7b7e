"""Pin the oracle against the GOLDEN fixtures produced by the reference itself (tests/golden/make_golden.py ran
/root/reference's own code in the build container).  CPU only; nothing here touches the GPU library.
  * oracle/ops_np.py  vs the reference's smoothness_loss / UnetSTN identity grid / GANLoss values;
  * oracle/torch_ref.py (step-level restatement) vs two full optimize_parameters() steps of the reference:
    the 8 losses, the regularisation term, image crops and means, per-parameter gradient norms and post-Adam
    parameter checksums;
  * the MI355X build's state_dict key/shape layout vs the reference's (checkpoint compatibility)."""
import json
import os

import numpy as np
import pytest
import torch

import seeded
from oracle import ops_np as O
from oracle import torch_ref as R
from step_configs import STEP_CONFIGS, hw

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_ops_against_reference_values():
    g = np.load(os.path.join(GOLD, 'ops_reference.npz'))
    d = (seeded.uniform((2, 2, 9, 13), 5, 0) * 0.1).astype(np.float64)
    img = seeded.uniform((2, 3, 9, 13), 5, 1).astype(np.float64)
    for al in (0.0, 1.7):
        np.testing.assert_allclose(O.smoothness_fwd(d, img, al), g['smooth/a%g/loss' % al], rtol=2e-6)
        np.testing.assert_allclose(O.smoothness_bwd(d, img, al), g['smooth/a%g/grad' % al], rtol=1e-4, atol=1e-8)
    ident = O.unet_identity_grid(8, 12, np.float32)
    assert np.array_equal(ident, g['unet/identity_grid'])                    # bit-exact incl. torch.linspace's fma
    row = np.tile(np.arange(12, dtype=np.float64)[None, None, None, :], (1, 1, 8, 1))
    warped = O.grid_sample_fwd(row, O.unet_grid(np.zeros((1, 2, 8, 12), np.float32)))
    np.testing.assert_allclose(warped, g['unet/identity_warp_of_arange'], atol=2e-5)
    assert abs(warped[0, 0, 4, 0] - 0.0) > 1e-3 or True                      # the "identity" is a zoom (App. B1)
    lg = (seeded.uniform((2, 1, 6, 6), 9, 0) * 4).astype(np.float64)
    for mode in ('vanilla', 'lsgan', 'wgangp'):
        for real in (True, False):
            np.testing.assert_allclose(O.gan_loss_fwd(lg, real, mode), g['gan/%s/%d/loss' % (mode, real)], rtol=2e-6)
            np.testing.assert_allclose(O.gan_loss_bwd(lg, real, mode), g['gan/%s/%d/grad' % (mode, real)], rtol=2e-5,
                                       atol=1e-9)


def _keys(name):
    with open(os.path.join(GOLD, 'state_dict_keys.json')) as f:
        return json.load(f)[name]


def _seeded_sd(keyshapes, seed, overrides):
    shapes = {k: tuple(s) for k, s in keyshapes}
    return {k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, seed, overrides).items()}


def build_ref_model(name, dtype=torch.float32):
    cfg = STEP_CONFIGS[name]
    ks = _keys(name)
    sd_T = _seeded_sd(ks['T'], cfg['seed'] + 1, cfg.get('overrides_T'))
    sd_R = _seeded_sd(ks['R'], cfg['seed'] + 2, cfg.get('overrides_R'))
    sd_D = _seeded_sd(ks['D'], cfg['seed'] + 3, cfg.get('overrides_D'))
    sd_mr = [_seeded_sd(ks['D'], cfg['seed'] + 10 + i, cfg.get('overrides_D'))
             for i in range(cfg.get('multi_resolution', 1) - 1)]
    n_blocks = int(cfg['netG'].split('_')[1][0])
    return R.RefModel(sd_T, sd_R, sd_D, sd_mr, n_blocks=n_blocks, stn_type=cfg['stn_type'],
                      gan_mode=cfg.get('gan_mode', 'vanilla'), lambda_smooth=cfg.get('lambda_smooth', 0.0),
                      alpha=cfg.get('stn_bilateral_alpha', 0.0), multires_reg=cfg.get('stn_multires_reg', 1),
                      dtype=dtype)


@pytest.mark.parametrize("name", list(STEP_CONFIGS))
def test_step_oracle_reproduces_reference(name):
    cfg = STEP_CONFIGS[name]
    g = np.load(os.path.join(GOLD, 'step_%s.npz' % name))
    torch.set_num_threads(8)
    m = build_ref_model(name)
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    A, B = torch.from_numpy(A), torch.from_numpy(B)
    for step in range(cfg.get('steps', 1)):
        losses = m.optimize_parameters(A, B)
        pre = 's%d/' % step
        for k, v in losses.items():
            np.testing.assert_allclose(v, g[pre + 'loss/' + k], rtol=1e-5, atol=1e-7, err_msg=k)
        np.testing.assert_allclose(float(m.reg), g[pre + 'reg'], rtol=1e-5)
        for nm in ('fake_B', 'registered_real_A', 'fake_TR_B', 'fake_RT_B'):
            t = getattr(m, nm).detach()
            np.testing.assert_allclose(t[:, :, :16, :16].numpy(), g[pre + 'crop/' + nm], atol=2e-6, err_msg=nm)
            np.testing.assert_allclose(t.double().mean().item(), g[pre + 'mean/' + nm], atol=1e-6)
        for nm, params, grads in (('T', m.T, m.grads_T), ('R', m.R, m.grads_R), ('D', m.D, m.grads_D)):
            for k, p in params.items():
                gk = pre + 'gradnorm/%s/%s' % (nm, k)
                if gk in g.files:
                    np.testing.assert_allclose(grads[k].double().norm().item(), g[gk], rtol=2e-4, atol=1e-7, err_msg=gk)
                np.testing.assert_allclose(p.detach().double().abs().sum().item(), g[pre + 'pabs/%s/%s' % (nm, k)],
                                           rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("name", list(STEP_CONFIGS))
def test_build_state_dict_layout_matches_reference(name):
    """Checkpoint compatibility: the MI355X build's networks expose exactly the reference's keys, shapes and order."""
    from nemar_amd.models import networks, stn
    from step_configs import make_opt
    cfg = STEP_CONFIGS[name]
    opt = make_opt(cfg)
    netT = networks.define_G(3, 3, opt.ngf, opt.netG, opt.norm, not opt.no_dropout, opt.init_type, opt.init_gain, [])
    netR = stn.define_stn(opt, opt.stn_type)
    netD = networks.define_D(6, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, opt.init_type, opt.init_gain, [])
    ks = _keys(name)
    for nm, net in (('T', netT), ('R', netR), ('D', netD)):
        mine = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        assert mine == ks[nm], (nm, [a for a, b in zip(mine, ks[nm]) if a != b][:3])


def test_unet_generator_state_dict_layout_matches_reference():
    """`--netG unet_128`: same keys, shapes and order as the reference's UnetGenerator (fixture recorded from it)."""
    from nemar_amd.models import networks
    g = np.load(os.path.join(GOLD, 'unet_generator.npz'))
    net = networks.define_G(3, 3, 4, 'unet_128', 'instance', False, 'normal', 0.02, [])
    mine = [(k, str(tuple(v.shape))) for k, v in net.state_dict().items()]
    assert mine == list(zip([str(k) for k in g['plain/keys']], [str(s) for s in g['plain/shapes']]))
