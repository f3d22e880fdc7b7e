"""`-m gpu`: the parts of the reference's plug-in API around the hot path (SURVEY.md §8f-1,2): checkpoint files that
interchange with the reference (`{epoch}_net_{T,R,D}.pth`, plain state_dicts with the reference's keys), the inference
path (`eval()` / `test()`), the inspection API `netR.get_grid()`, and `update_learning_rate()`."""
import json
import os

import numpy as np
import pytest
import torch

import seeded
import step_parity
from step_configs import STEP_CONFIGS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _inputs(cfg):
    a, b = seeded.seeded_images(cfg['batch'], 3, cfg['size'], cfg['size'], cfg['seed'])
    return {'A': torch.from_numpy(a), 'B': torch.from_numpy(b), 'A_paths': ['a'], 'B_paths': ['b']}


@pytest.mark.parametrize("name", ["affine128", "unet256"])
def test_checkpoint_roundtrip_and_inference(name, tmp_path):
    cfg = STEP_CONFIGS[name]
    m = step_parity.build_hip_model(name)
    m.save_dir = str(tmp_path)
    m.set_input(_inputs(cfg))
    m.optimize_parameters()                      # weights move away from the seeded ones
    m.save_networks('7')
    keys = json.load(open(os.path.join(GOLD, 'state_dict_keys.json')))
    for net in ('T', 'R', 'D'):
        sd = torch.load(os.path.join(str(tmp_path), '7_net_%s.pth' % net), map_location='cpu')
        assert all(isinstance(v, torch.Tensor) and v.device.type == 'cpu' for v in sd.values())
        ref = keys[name][net]                    # [[key, shape], ...] recorded from the reference for this configuration
        assert [(k, list(v.shape)) for k, v in sd.items()] == [(k, list(s)) for k, s in ref]
    # inference on the trained model, then on a fresh model that loaded the checkpoint: identical outputs
    m.eval()
    m.test()
    want = {k: getattr(m, k).detach().cpu().clone() for k in ('fake_B', 'registered_real_A', 'fake_TR_B', 'fake_RT_B')}
    assert not any(t.requires_grad for t in (m.fake_B, m.fake_TR_B, m.fake_RT_B))
    m2 = step_parity.build_hip_model(name)
    m2.save_dir = str(tmp_path)
    m2.load_networks('7')
    m2.set_input(_inputs(cfg))
    m2.eval()
    m2.test()
    for k, w in want.items():      # same weights, same kernels; split reductions sum through fp32 atomics => not bitwise
        assert (getattr(m2, k).detach().cpu() - w).abs().max().item() < 2e-4, k   # the warp amplifies 1e-7 changes of theta by the image slope
    vis = m2.get_current_visuals()
    assert list(vis.keys()) == ['real_A', 'real_B', 'fake_TR_B', 'fake_RT_B', 'registered_real_A', 'fake_B']
    assert m2.get_image_paths() == ['a']


@pytest.mark.parametrize("name", ["affine128", "unet256"])
def test_get_grid_matches_the_warp(name):
    """netR.get_grid() (inspection API, torch ops) and the fused warp kernels describe the same sampling."""
    import torch.nn.functional as F
    cfg = STEP_CONFIGS[name]
    m = step_parity.build_hip_model(name)
    m.set_input(_inputs(cfg))
    m.eval()
    with torch.no_grad():
        grid = m.netR.get_grid(m.real_A, m.real_B)
        warped, _ = m.netR(m.real_A, m.real_B, apply_on=[m.real_A])
        assert grid.shape == (cfg['batch'], cfg['size'], cfg['size'], 2)
        ref = F.grid_sample(m.real_A.cpu(), grid.cpu(), mode='bilinear', padding_mode='zeros', align_corners=False)
    assert (warped[0].cpu() - ref).abs().max().item() < 2e-4


def test_update_learning_rate_steps_the_schedulers():
    m = step_parity.build_hip_model("affine128")
    lr0 = m.optimizers[0].param_groups[0]['lr']
    m.update_learning_rate()
    assert m.optimizers[0].param_groups[0]['lr'] <= lr0
