"""Scaled-down step configurations shared by the golden generator (reference), the oracle tests and the GPU
parity tests.  UnetSTN cfg 'A' cannot go below 256x256 (7 poolings + ReflectionPad at the 2x2 bottleneck,
SURVEY.md Appendix B5), so the small cases shrink widths and batch instead."""
import argparse

STEP_CONFIGS = {
    # BASELINE config 1 shape (affine, 128x128) at reduced width
    'affine128': dict(stn_type='affine', netG='resnet_3blocks', ngf=8, ndf=8, size=128, batch=2, seed=11,
                      lambda_smooth=0.5, steps=2,
                      overrides_R={'net.local.2.weight': 0.02, 'net.local.2.bias': 0.05}),
    # BASELINE config 2/3/4 features: dense deformation + bilateral smoothness + multi-resolution D and regulariser
    'unet256': dict(stn_type='unet', netG='resnet_3blocks', ngf=8, ndf=8, size=256, batch=1, seed=23,
                    lambda_smooth=10.0, stn_bilateral_alpha=1.5, multi_resolution=2, stn_multires_reg=2, steps=2,
                    overrides_R={'offset_map.output.conv2d.weight': 0.02}),
    # plain unet path, lsgan objective
    'unet256_lsgan': dict(stn_type='unet', netG='resnet_3blocks', ngf=8, ndf=8, size=256, batch=1, seed=31,
                          lambda_smooth=1.0, gan_mode='lsgan', steps=1,
                          overrides_R={'offset_map.output.conv2d.weight': 0.05}),
}

# Full-width steps at the BASELINE.json shapes (ngf = ndf = 64, resnet_9blocks): the production kernel dispatch
# (128x128 wave-specialised tiles, one-round weight-gradient splits, ...) compared with fixtures recorded from TWO runs of
# the reference itself (fp32 and fp64 — the gap between them calibrates each quantity's tolerance).  Scalars only
# (losses, image statistics, per-parameter gradient norms / seeded projections, post-Adam checksums): SURVEY.md §8c.
FULL_CONFIGS = {
    # C1: BASELINE config 1 at full width — affine STN, resnet_6blocks, 128x128, batch 1 (the reference's own CPU-runnable case)
    'c1_full': dict(stn_type='affine', netG='resnet_6blocks', ngf=64, ndf=64, size=128, batch=1, seed=37,
                    lambda_smooth=0.5, steps=1, perturbed=2, overrides_R={'net.local.2.weight': 0.02, 'net.local.2.bias': 0.05}),
    # C2: unet cfg 'A', 256x256 (the bench workload, at batch 1)
    'c2_full': dict(stn_type='unet', netG='resnet_9blocks', ngf=64, ndf=64, size=256, batch=1, seed=41,
                    lambda_smooth=10.0, steps=1, overrides_R={'offset_map.output.conv2d.weight': 0.02}),
    # C2 at the BENCH batch (8): the batched T / D passes, the batch-16 / batch-24 kernel launches of the timed step
    'c2_b8': dict(stn_type='unet', netG='resnet_9blocks', ngf=64, ndf=64, size=256, batch=8, seed=59,
                  lambda_smooth=10.0, steps=1, overrides_R={'offset_map.output.conv2d.weight': 0.02}),
    # C3: + multi-resolution discriminators, batch 2
    'c3_full': dict(stn_type='unet', netG='resnet_9blocks', ngf=64, ndf=64, size=256, batch=2, seed=43,
                    lambda_smooth=10.0, multi_resolution=2, steps=1,
                    overrides_R={'offset_map.output.conv2d.weight': 0.02}),
    # C4: 512x512, bilateral smoothness, multi-resolution regulariser
    'c4_full': dict(stn_type='unet', netG='resnet_9blocks', ngf=64, ndf=64, size=512, batch=2, seed=47,
                    lambda_smooth=10.0, stn_bilateral_alpha=1.5, stn_multires_reg=2, steps=1,
                    overrides_R={'offset_map.output.conv2d.weight': 0.02}),
    # C5: 1024x1024, the deeper registration net (stn_cfg 'deep': 9 levels -> 2x2 bottleneck at 1024x1024).  fp32 reference run
    # only: the fp64 run of the reference needs more host memory than the build container has (a 26 GB single allocation failed
    # under a 56 GB limit, /tmp/make_golden_c5.log of round 3), so C5's tolerances are the fixed bases of compare().
    # The reference's DEFAULT geometry and nets (options/base_options.py:34,47-48: --img_height 288 --img_width 384, resnet_9blocks;
    # models/stn/__init__.py:12: --stn_type affine) — non-square, not a power of two: 72x96 residual-block maps, 36x48 / 35x47 D maps
    'default_full': dict(stn_type='affine', netG='resnet_9blocks', ngf=64, ndf=64, height=288, width=384, size=None, batch=1,
                         seed=61, lambda_smooth=0.5, steps=1, perturbed=2,
                         overrides_R={'net.local.2.weight': 0.02, 'net.local.2.bias': 0.05}),
    # a non-square dense-field case: unet cfg 'A' at 256x384 (2x3 bottleneck), batch 1
    'c2_256x384': dict(stn_type='unet', netG='resnet_9blocks', ngf=64, ndf=64, height=256, width=384, size=None, batch=1,
                       seed=67, lambda_smooth=10.0, steps=1, overrides_R={'offset_map.output.conv2d.weight': 0.02}),
    'c5_full': dict(stn_type='unet', stn_cfg='deep', netG='resnet_9blocks', ngf=64, ndf=64, size=1024, batch=1,
                    seed=53, lambda_smooth=10.0, steps=1, f64=False, overrides_R={'offset_map.output.conv2d.weight': 0.02}),
}

# BASELINE.json config 5's "deep" registration cfg does not exist in the reference (models/stn/unet_stn.py:11-25 defines
# 'A' only); the build adds it through the same dict mechanism, and the fixture generator injects the same entries into
# the imported reference's dicts (data, not code) so that the reference's own ResUnet builds it.
DEEP_STN_CFG = dict(ndf=[32, 64, 64, 64, 64, 64, 64, 64, 64], nuf=[64, 64, 64, 64, 64, 64, 64, 64, 32],
                    use_down_resblocks=True, resnet_nblocks=3, refine_output=True, down_activation='leaky_relu',
                    up_activation='leaky_relu')


def hw(cfg):
    """(height, width) of a configuration: 'size' for the square ones, 'height' / 'width' otherwise."""
    return (cfg['height'], cfg['width']) if cfg.get('size') is None else (cfg['size'], cfg['size'])


def make_opt(cfg, gpu_ids=()):
    return argparse.Namespace(
        gpu_ids=list(gpu_ids), isTrain=True, checkpoints_dir='/tmp/nemar_ck', name='golden', preprocess='none',
        input_nc=3, output_nc=3, ngf=cfg['ngf'], ndf=cfg['ndf'], netG=cfg['netG'], netD='basic', n_layers_D=3,
        norm='instance', init_type='normal', init_gain=0.02, no_dropout=True, direction='AtoB',
        img_height=hw(cfg)[0], img_width=hw(cfg)[1], lr=2e-4, beta1=0.5, gan_mode=cfg.get('gan_mode', 'vanilla'),
        lambda_GAN=1.0, lambda_recon=100.0, lambda_smooth=cfg.get('lambda_smooth', 0.0), enable_tbvis=False,
        multi_resolution=cfg.get('multi_resolution', 1), stn_cfg=cfg.get('stn_cfg', 'A'), stn_type=cfg['stn_type'],
        stn_bilateral_alpha=cfg.get('stn_bilateral_alpha', 0.0), stn_no_identity_init=False,
        stn_multires_reg=cfg.get('stn_multires_reg', 1), lr_policy='linear', epoch_count=1, niter=100,
        niter_decay=100, continue_train=False, verbose=False, batch_size=cfg['batch'], load_iter=0, epoch='latest',
        lr_decay_iters=50, model='nemar')
