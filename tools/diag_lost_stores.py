"""The lost-store anomaly of the side-stream weight-gradient branch: statistics under one experimental condition.

Runs DIAG_RUNS fresh one-step runs of a full-width config with the 7x7 layers' weight gradients on the side queue
(the default since round 5; NEMAR_SIDE_K7=0 takes them off) and compares the gradient w.r.t. the deformation field that the FIRST warp's backward returns (the victim:
grid_sample_bwd_kernel<UNET, false>, warp.hip) with the first run's, element by element.  Prints one summary line.

Conditions (environment):
  DIAG_LIB=path        a variant build of the library (tools/diag_variants.py)
  DIAG_SENTINEL=1      the victim's output buffers come from a pool that was NaN-filled long before (no fill kernel next to the
                       victim): a LOST store shows as NaN, a wrongly computed / overwritten value as a finite number
  DIAG_CFG=c2_full     which FULL_CONFIGS entry
  K7_SKIP_SMAX/_MAIN/_SUMS (with the k7env variant)   drop one of the stem weight-gradient call's launches
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
os.environ.setdefault('NEMAR_SIDE_STREAM', '1')
import torch  # noqa: E402
from nemar_amd import _lib  # noqa: E402
if os.environ.get('DIAG_LIB'):
    _lib.DEFAULT_PATH = os.path.abspath(os.environ['DIAG_LIB'])
import seeded  # noqa: E402
from nemar_amd import ops  # noqa: E402
from step_configs import FULL_CONFIGS, hw  # noqa: E402
import test_step_full_gpu  # noqa: E402

name = os.environ.get('DIAG_CFG', 'c2_full')
runs = int(os.environ.get('DIAG_RUNS', '80'))
cfg = FULL_CONFIGS[name]
a, b = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
dev = torch.device('cuda', 0)

pool = []
if os.environ.get('DIAG_SENTINEL'):
    H, W = hw(cfg)
    pool = [torch.full((cfg['batch'], 2, H, W), float('nan'), device=dev) for _ in range(2 * runs + 4)]
    torch.cuda.synchronize()

    def alloc(gs):
        if pool and tuple(pool[-1].shape) == tuple(gs.shape):
            return pool.pop()
        return torch.empty_like(gs)
    ops._ggs_alloc = alloc

ref = None
bad = nan_runs = 0
lanes = {}
planes = {}
for run in range(runs):
    m = test_step_full_gpu.build(name)
    got = {}
    R = m.netR
    o_warp = R.warp
    count = [0]

    def warp(field, imgs):
        count[0] += 1
        fc = field[1].clone()
        k = count[0]
        fc.register_hook(lambda g, k=k: got.setdefault(k, g))      # the tensor itself: read after the step
        return o_warp((field[0], fc), imgs)

    R.warp = warp
    m.set_input({'A': torch.from_numpy(a), 'B': torch.from_numpy(b), 'A_paths': [''], 'B_paths': ['']})
    m.optimize_parameters()
    torch.cuda.synchronize()
    cur = got[1].detach().cpu()
    if ref is None:
        ref = cur
        assert not torch.isnan(ref).any(), "the first run already shows the sentinel"
        continue
    diff = (cur != ref) & ~(torch.isnan(cur) & torch.isnan(ref))
    if diff.any():
        bad += 1
        idx = diff.nonzero()
        nn = int(torch.isnan(cur[diff]).sum())
        nan_runs += nn > 0
        for i in idx:
            lanes[int(i[3]) % 64 // 16] = lanes.get(int(i[3]) % 64 // 16, 0) + 1
            planes[int(i[1])] = planes.get(int(i[1]), 0) + 1
        if bad <= 3:
            i = idx[0]
            print('   run %d: %d elements differ (%d NaN); first at (n %d, plane %d, y %d, x %d): %r vs %r' % (
                run, int(diff.sum()), nn, int(i[0]), int(i[1]), int(i[2]), int(i[3]), float(cur[tuple(i)]), float(ref[tuple(i)])))
print('%-10s lib=%s sentinel=%s skip=%s: %d of %d runs differ; runs with NaN among the differing elements: %d; '
      'differing elements by 16-lane quarter of a wave %s, by plane %s' % (
          name, os.path.basename(os.environ.get('DIAG_LIB', 'product')), bool(pool or os.environ.get('DIAG_SENTINEL')),
          [k for k in ('K7_SKIP_SMAX', 'K7_SKIP_MAIN', 'K7_SKIP_SUMS') if os.environ.get(k)], bad, runs - 1, nan_runs,
          dict(sorted(lanes.items())), dict(sorted(planes.items()))))
