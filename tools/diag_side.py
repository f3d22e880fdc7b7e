"""Run-to-run determinism of the full-width step with the side-stream weight-gradient branch (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seeded
from nemar_amd import ops
from step_configs import FULL_CONFIGS, hw
import test_step_full_gpu
name = sys.argv[1] if len(sys.argv) > 1 else 'c2_full'
cfg = FULL_CONFIGS[name]
a, b = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
snaps = []
for run in range(int(os.environ.get('DIAG_RUNS', '3'))):
    m = test_step_full_gpu.build(name)
    for _ in range(2):
        m.set_input({'A': torch.from_numpy(a), 'B': torch.from_numpy(b), 'A_paths': [''], 'B_paths': ['']})
        m.optimize_parameters()
    torch.cuda.synchronize()
    snaps.append([o.flat_p.detach().cpu().clone() for o in m.optimizers] + [o.flat_g.detach().cpu().clone() for o in m.optimizers])
names = ['p_D', 'p_R', 'p_T', 'g_D', 'g_R', 'g_T']
for r in range(1, len(snaps)):
    for nme, x, y in zip(names, snaps[0], snaps[r]):
        d = (x - y).abs()
        print('run 0 vs %d  %s: max diff %.3e  differing %d of %d' % (r, nme, float(d.max()), int((x != y).sum()), x.numel()))
        if nme.startswith('g') and int((x != y).sum()):
            idx = (x != y).nonzero().flatten()
            print('     first / last differing index', int(idx[0]), int(idx[-1]))
