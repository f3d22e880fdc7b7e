"""Which parameters' gradients differ run to run after ONE step (diagnostic for the side-stream branch)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seeded
from nemar_amd import ops
from step_configs import FULL_CONFIGS, hw
import test_step_full_gpu
name = 'c2_full'
cfg = FULL_CONFIGS[name]
a, b = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
ref = None
for run in range(int(os.environ.get('DIAG_RUNS', '6'))):
    m = test_step_full_gpu.build(name)
    m.set_input({'A': torch.from_numpy(a), 'B': torch.from_numpy(b), 'A_paths': [''], 'B_paths': ['']})
    m.optimize_parameters()
    torch.cuda.synchronize()
    cur = {}
    for net in ('netT', 'netR', 'netD'):
        for n, p in getattr(m, net).named_parameters():
            if p.grad is not None:
                cur[net + '.' + n] = p.grad.detach().cpu().clone()
    if ref is None:
        ref = cur
        continue
    bad = [(k, int((cur[k] != ref[k]).sum()), cur[k].numel(), float((cur[k] - ref[k]).abs().max())) for k in cur if not torch.equal(cur[k], ref[k])]
    print('run %d: %d parameters differ' % (run, len(bad)))
    for k, c, n, d in bad[:12]:
        print('    %-40s %8d of %8d  max %.3e' % (k, c, n, d))
