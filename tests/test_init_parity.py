"""the build's initialisation reproduces the reference's, tensor by tensor — incl. its quirks (decoder
convs of the registration U-Net are kaiming whatever --init_type says: the `init_fun` typo at reference
models/stn/unet_stn.py:62-63; 'zeros' is N(0, 1e-5): models/stn/layers.py:46-47; the affine head is N(0, 5e-4) with
zero bias: models/stn/affine_stn.py:75-76; nn.Linear keeps torch's default init).  CPU only: the nets are constructed,
never run.  Expected values: tests/golden/init_stats.json = (numel, mean, std) of every state_dict tensor after the
REFERENCE's own constructors (tests/golden/make_golden.py --only init).  Both sides are random draws, so the bound is
statistical: 5 sigma of the estimators of both samples."""
import argparse
import json
import math
import os

import pytest
import torch

from step_configs import FULL_CONFIGS, STEP_CONFIGS, make_opt

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _nets():
    from nemar_amd.models import networks, stn
    torch.manual_seed(4321)
    c2 = make_opt(FULL_CONFIGS['c2_full'])
    return {
        'T': networks.define_G(3, 3, 64, 'resnet_9blocks', 'instance', True, 'normal', 0.02, []),
        'D': networks.define_D(6, 64, 'basic', 3, 'instance', 'normal', 0.02, []),
        'R_unet': stn.define_stn(c2, 'unet'),
        'R_unet_noident': stn.define_stn(argparse.Namespace(**{**vars(c2), 'stn_no_identity_init': True}), 'unet'),
        'R_affine': stn.define_stn(make_opt(STEP_CONFIGS['affine128']), 'affine'),
    }


@pytest.mark.parametrize("net", ['T', 'D', 'R_unet', 'R_unet_noident', 'R_affine'])
def test_init_statistics_match_reference(net):
    with open(os.path.join(GOLD, 'init_stats.json')) as f:
        want = json.load(f)[net]
    sd = _nets()[net].state_dict()
    assert [k for k, *_ in want] == list(sd.keys())
    for (k, n, mean, std), v in zip(want, sd.values()):
        assert v.numel() == n, k
        v = v.double()
        if std == 0.0:                         # biases the reference zero-fills
            assert float(v.abs().max()) == 0.0, k
            continue
        s = float(v.std())
        # two independent samples of size n: std estimators differ by ~ std * sqrt(2 / (2n)) ; means by std * sqrt(2/n)
        assert abs(s - std) <= 5.0 * std * math.sqrt(1.0 / n) + 1e-12, (k, s, std)
        assert abs(float(v.mean()) - mean) <= 5.0 * std * math.sqrt(2.0 / n), (k, float(v.mean()), mean)
