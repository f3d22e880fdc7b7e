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
from step_configs import STEP_CONFIGS, hw

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _inputs(cfg):
    a, b = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
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
    for k, w in want.items():      # same weights, same kernels, no atomics on the path: the reloaded model reproduces the images
        assert (getattr(m2, k).detach().cpu() - w).abs().max().item() < 2e-4, k   # (bound kept from round 1; the warp amplifies 1e-7 changes of theta by the image slope)
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


def test_loss_attributes_are_tensors_and_grads_stay_in_the_flat_buffer():
    """`model.loss_<name>` are 0-dim tensors like the reference's (arithmetic / .item() / add_scalar work); a parameter
    whose .grad was cleared or re-seated by user code goes back to its view of the optimizer's flat gradient buffer."""
    cfg = STEP_CONFIGS["affine128"]
    m = step_parity.build_hip_model("affine128")
    m.set_input(_inputs(cfg))
    m.optimize_parameters()
    for name in m.loss_names:
        v = getattr(m, 'loss_' + name)
        assert isinstance(v, torch.Tensor) and v.dim() == 0 and v.is_cuda, name
    total = m.loss_L1_TR + m.loss_L1_RT + 0.5 * m.loss_GAN_TR
    assert abs(total.item() - (float(m.loss_L1_TR) + float(m.loss_L1_RT) + 0.5 * float(m.loss_GAN_TR))) < 1e-4
    assert abs(float(m.loss_D) - m.get_current_losses()['D']) < 1e-6
    # re-seated gradients
    p = next(m.netD.parameters())
    view_ptr = p.grad.data_ptr()
    p.grad = None
    q = list(m.netD.parameters())[2]
    q.grad = torch.zeros_like(q)
    m.set_input(_inputs(cfg))
    m.optimize_parameters()
    assert p.grad is not None and p.grad.data_ptr() == view_ptr and float(p.grad.abs().sum()) > 0
    assert q.grad.data_ptr() == q._flat_grad.data_ptr()


def test_dropout_masks_follow_the_torch_seed_and_are_regenerated_in_backward():
    """the Philox stream is seeded from torch.initial_seed() + rank in NEMARModel.__init__ (different per rank /
    per torch.manual_seed), and the backward pass regenerates exactly the forward mask."""
    from nemar_amd import ops
    x = torch.ones(1, 4, 64, 64, device='cuda', requires_grad=True)
    masks = []
    for seed in (11, 11, 12):
        ops.manual_seed(seed)
        y = ops.dropout(x, 0.5, True)
        g, = torch.autograd.grad(y.sum(), x)
        assert torch.equal(g, (y != 0).float() * 2.0)          # same mask, same 1/(1-p) scale
        masks.append((y != 0).cpu())
    assert torch.equal(masks[0], masks[1]) and not torch.equal(masks[0], masks[2])
    torch.manual_seed(123)
    m1 = step_parity.build_hip_model("affine128")
    s1 = ops._dropout_state["seed"]
    torch.manual_seed(124)
    m2 = step_parity.build_hip_model("affine128")
    assert ops._dropout_state["seed"] != s1 and s1 == (123 & 0xFFFFFFFFFFFFFFFF)
