"""How far apart are the gradients of the batched passes (T([a ; R(a)]), D([real ; fakes])) and of the reference's separate calls on the
reduced-width configuration of tests/test_step_gpu.py::test_batched_passes_match_reference_call_order, over several image seeds?  The two
forms differ in fp32 summation order only (reduction splits depend on the batch); an element of a LeakyReLU / max-pool input at rounding
distance of its kink turns that into a discrete gradient difference.  Prints, per seed offset, max |x - y| / max |y| of D's, T's and R's flat
gradients and the relative L2 distance.   python tools/diag_batched_order.py [name] [offsets...]      (NEMAR_TUNE=36=1 for the other setting)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import seeded  # noqa: E402
import step_parity  # noqa: E402
from step_configs import STEP_CONFIGS, hw  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'unet256'
offsets = [int(a) for a in sys.argv[2:]] or [0, 1000, 2000, 3000, 4000, 5000]
cfg = STEP_CONFIGS[name]
print('NEMAR_TUNE=%r  config %s' % (os.environ.get('NEMAR_TUNE', ''), name))
for off in offsets:
    results = []
    for flag in ('1', '0'):
        os.environ['NEMAR_BATCHED_PASSES'] = flag
        m = step_parity.build_hip_model(name)
        a, b = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'] + off)
        m.set_input({'A': torch.from_numpy(a), 'B': torch.from_numpy(b), 'A_paths': [''], 'B_paths': ['']})
        m.forward()
        m.set_requires_grad([m.netT, m.netR], False)
        m.optimizer_D.zero_grad()
        m.backward_D()
        gd = m.optimizer_D.flat_g.detach().cpu().clone()
        m.set_requires_grad([m.netT, m.netR], True)
        m.set_requires_grad([m.netD, *m.netD_multiresolution], False)
        m.optimizer_R.zero_grad(); m.optimizer_T.zero_grad()
        m.backward_T_and_R()
        results.append(dict(gd=gd, gt=m.optimizer_T.flat_g.detach().cpu().clone(), gr=m.optimizer_R.flat_g.detach().cpu().clone()))
    x, y = results
    row = []
    for g in ('gd', 'gt', 'gr'):
        d = (x[g] - y[g]).double()
        row.append('%s max %.2e l2 %.2e' % (g, d.abs().max().item() / y[g].abs().max().item(), d.norm().item() / y[g].double().norm().item()))
    print('image seed +%-5d %s' % (off, '   '.join(row)), flush=True)
