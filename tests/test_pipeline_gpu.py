"""`-m gpu`: input pipeline on the device and the observability scalars (f4) with the real model."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _opt(tmp, extra=()):
    from nemar_amd.train import _Options
    argv = ['--model', 'nemar', '--stn_type', 'affine', '--netG', 'resnet_3blocks', '--ngf', '8', '--ndf', '8', '--dataset_mode',
            'gpupairs', '--dataroot', 'synthetic', '--img_height', '128', '--img_width', '128', '--crop_size', '128',
            '--load_size', '140', '--batch_size', '2', '--pool_size_pairs', '6', '--checkpoints_dir', str(tmp), '--name',
            'pipe', '--no_dropout', '--print_freq', '2', '--niter', '1', '--niter_decay', '0', '--save_epoch_freq', '100',
            '--gpu_ids', '0', *extra]
    return _Options().parse(argv, quiet=True)


def test_gpupairs_batches_follow_the_reference_contract(tmp_path):
    from nemar_amd.data import create_dataset
    opt = _opt(tmp_path)
    ds = create_dataset(opt)
    assert len(ds) == 6
    seen = 0
    for data in ds:
        assert set(data) == {'A', 'B', 'A_paths', 'B_paths'}
        A, B = data['A'], data['B']
        assert A.is_cuda and A.shape == (2, 3, 128, 128) and B.shape == A.shape and A.dtype == torch.float32
        assert float(A.min()) >= -1.0 and float(A.max()) <= 1.0 and len(data['A_paths']) == 2
        # the SAME crop / flip for both images of a pair, and exactly crop -> flip -> (v - 0.5) / 0.5
        par = ds.dataset._last_params
        for b, (i, y0, x0, flip) in enumerate(par):
            for pool, got in ((ds.dataset.pool_A, A), (ds.dataset.pool_B, B)):
                crop = pool[i, :, y0:y0 + 128, x0:x0 + 128]
                if flip:
                    crop = crop.flip(-1)
                assert torch.allclose(got[b], (crop - 0.5) / 0.5, atol=1e-6)
        seen += 1
    assert seen == 3


def test_gpupairs_whole_non_square_images_and_resize_modes(tmp_path):
    """--preprocess none feeds whole NON-SQUARE images (the reference's default geometry 288 x 384); scale_width / resize_and_crop resize
    the pool once (reference data/base_dataset.py:81-113) and then behave like the crop path."""
    from nemar_amd.data import create_dataset
    opt = _opt(tmp_path, ['--preprocess', 'none', '--img_height', '288', '--img_width', '384', '--no_flip'])
    ds = create_dataset(opt)
    data = next(iter(ds))
    assert data['A'].shape == (2, 3, 288, 384)
    for b, (i, y0, x0, flip) in enumerate(ds.dataset._last_params):
        assert (y0, x0, flip) == (0, 0, 0)
        assert torch.allclose(data['A'][b], (ds.dataset.pool_A[i] - 0.5) / 0.5, atol=1e-6)
        assert torch.allclose(data['B'][b], (ds.dataset.pool_B[i] - 0.5) / 0.5, atol=1e-6)
    # a pool on disk at another size: scale_width resizes it to width load_size, the height follows
    root = tmp_path / 'pairs'
    root.mkdir()
    rng = np.random.default_rng(0)
    np.save(root / 'A.npy', (rng.random((4, 60, 90, 3)) * 255).astype(np.uint8))
    np.save(root / 'B.npy', (rng.random((4, 60, 90, 3)) * 255).astype(np.uint8))
    opt = _opt(tmp_path, ['--dataroot', str(root), '--preprocess', 'scale_width', '--load_size', '120'])
    ds = create_dataset(opt)
    data = next(iter(ds))
    assert data['A'].shape == (2, 3, 80, 120) and float(data['A'].abs().max()) <= 1.0
    opt = _opt(tmp_path, ['--dataroot', str(root), '--preprocess', 'resize_and_crop', '--load_size', '72', '--crop_size', '64'])
    ds = create_dataset(opt)
    assert ds.dataset.pool_A.shape == (4, 3, 72, 72)
    data = next(iter(ds))
    assert data['A'].shape == (2, 3, 64, 64)


def test_train_loop_writes_the_loss_line_and_offset_scalars(tmp_path, capsys):
    import json
    import os
    from nemar_amd import train
    train.main(['--model', 'nemar', '--stn_type', 'unet', '--netG', 'resnet_3blocks', '--ngf', '8', '--ndf', '8',
                '--dataset_mode', 'gpupairs', '--dataroot', 'synthetic', '--img_height', '256', '--img_width', '256',
                '--crop_size', '256', '--load_size', '256', '--batch_size', '2', '--pool_size_pairs', '4', '--checkpoints_dir',
                str(tmp_path), '--name', 'obs', '--print_freq', '2', '--niter', '1', '--niter_decay', '0', '--save_epoch_freq',
                '100', '--gpu_ids', '0', '--lambda_smooth', '1.0', '--enable_tbvis', '--tbvis_iteration_update_rate', '1'])
    out = capsys.readouterr().out
    line = [l for l in out.splitlines() if l.startswith('(epoch: 1, iters: 2, time: ')]
    assert line and ' L1_TR: ' in line[0] and ' D: ' in line[0]
    log = open(os.path.join(str(tmp_path), 'obs', 'loss_log.txt')).read()
    assert line[0] in log
    rows = [json.loads(l) for l in open(os.path.join(str(tmp_path), 'obs', 'obs_tensorboard_logs', 'scalars.jsonl'))]
    tags = {r['tag'] for r in rows}
    assert {'offset/mean_x', 'offset/mean_y', 'loss/L1_TR', 'loss/D'} <= tags


def test_train_loop_with_the_step_as_a_graph(tmp_path, capsys):
    """--step_graph: the loop's steps are replays of one captured hipGraph; the reference's loss line comes out as before."""
    from nemar_amd import ops, train
    try:
        train.main(['--model', 'nemar', '--stn_type', 'affine', '--netG', 'resnet_3blocks', '--ngf', '8', '--ndf', '8',
                    '--dataset_mode', 'gpupairs', '--dataroot', 'synthetic', '--img_height', '128', '--img_width', '128',
                    '--crop_size', '128', '--load_size', '128', '--batch_size', '2', '--pool_size_pairs', '8', '--checkpoints_dir',
                    str(tmp_path), '--name', 'graph', '--print_freq', '4', '--niter', '1', '--niter_decay', '0', '--save_epoch_freq',
                    '100', '--gpu_ids', '0', '--step_graph'])
    finally:
        ops.step_params(False)
    out = capsys.readouterr().out
    line = [l for l in out.splitlines() if l.startswith('(epoch: 1, iters: 4, time: ')]
    assert line and ' L1_TR: ' in line[0] and ' D: ' in line[0]
    vals = [float(t) for t in line[0].replace(',', ' ').split() if t.replace('.', '', 1).replace('-', '', 1).isdigit()]
    assert all(v == v for v in vals)


def test_offset_meter_matches_the_reference_arithmetic():
    """reference tb_visualizer.py:71-74: mean over iterations of np.mean(offset[:, c])."""
    from nemar_amd.util.visualizer import OffsetMeter
    dev = torch.device('cuda:0')
    m = OffsetMeter(dev)
    g = torch.Generator(device=dev).manual_seed(3)
    fields = [torch.randn(2, 2, 64, 48, device=dev, generator=g) * 0.1 + 0.01 * k for k in range(3)]
    for f in fields:
        m.update(f)
    mx, my = m.means()
    want_x = np.mean([f[:, 0].double().mean().item() for f in fields])
    want_y = np.mean([f[:, 1].double().mean().item() for f in fields])
    assert abs(mx - want_x) < 1e-6 and abs(my - want_y) < 1e-6
    assert m.means() == (0.0, 0.0)
