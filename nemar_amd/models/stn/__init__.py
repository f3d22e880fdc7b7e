"""STN factory and command-line flags of the registration network: the drop-in contract of reference models/stn/__init__.py
(flag names and defaults :10-25, `define_stn(opt, stn_type)` :28-48).  Names / types / defaults are the interface and are identical;
everything else is this build's own."""
from .affine_stn import AffineSTN
from .unet_stn import UnetSTN

sampling_align_corners = False
sampling_mode = 'bilinear'

# (flag, argparse keywords, train-only) — the reference's five --stn_* options
_FLAGS = (
    ('--stn_cfg', dict(type=str, default='A',
                       help="layer table of the registration net: 'A' (the reference's), 'deep' (9 levels, for 1024x1024 inputs)"), False),
    ('--stn_type', dict(type=str, default='affine',
                        help='registration model: affine (6 parameters per pair) | unet (dense deformation field)'), False),
    ('--stn_bilateral_alpha', dict(type=float, default=0.0,
                                   help='unet only: edge-aware weight exp(-alpha |dI|) on the smoothness penalty; 0 switches it off'), True),
    ('--stn_no_identity_init', dict(action='store_true',
                                    help='unet only: start from a random field instead of the near-identity initialisation'), True),
    ('--stn_multires_reg', dict(type=int, default=1,
                                help='unet only: apply the smoothness penalty at this many resolutions (1 = full resolution only)'), True),
)
_BUILDERS = {
    'affine': lambda a, b, h, w, opt: AffineSTN(a, b, h, w, opt.stn_cfg, opt.init_type),
    'unet': lambda a, b, h, w, opt: UnetSTN(a, b, h, w, opt.stn_cfg, opt.init_type, opt.stn_bilateral_alpha,
                                            not opt.stn_no_identity_init, opt.stn_multires_reg),
}


def modify_commandline_options(parser, is_train=True):
    for flag, kw, train_only in _FLAGS:
        if is_train or not train_only:
            parser.add_argument(flag, **kw)
    return parser


def define_stn(opt, stn_type='affine'):
    """The STN for `opt`, on this process's GPU.  One process drives one GPU (no nn.DataParallel wrap, hence no `.module`
    indirection); an unknown type gives None, as the reference's factory does."""
    import torch
    make = _BUILDERS.get(stn_type)
    if make is None:
        return None
    a_to_b = opt.direction == 'AtoB'
    net = make(opt.input_nc if a_to_b else opt.output_nc, opt.output_nc if a_to_b else opt.input_nc, opt.img_height, opt.img_width, opt)
    if len(opt.gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(torch.device('cuda', opt.gpu_ids[0]))
    return net
