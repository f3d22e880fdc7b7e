"""Diagnostic: the D step of a step-parity configuration on the fp32 oracle's fakes, once per nemar_tune setting, with the
route every convolution launch took — where does a D gradient differ from the oracle, and is the run reproducible?"""
import os, sys
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
import step_parity as sp, seeded
from step_configs import STEP_CONFIGS
from nemar_amd import ops, _lib

name = sys.argv[1] if len(sys.argv) > 1 else 'unet256'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = STEP_CONFIGS[name]
L = _lib.load()
log = []
for op in ('conv2d_fwd', 'conv2d_bwd_data', 'conv2d_bwd_weight'):
    inner = getattr(L, op)
    def wrap(*a, _inner=inner, _op=op):
        rc = _inner(*a)
        log.append((_op, tuple(x for x in a if isinstance(x, int) and abs(x) < 1 << 20), L.last_route()))
        return rc
    L.__dict__[op] = wrap

ref = sp.build_ref_model(name)
hip = sp.build_hip_model(name)
A, B = seeded.seeded_images(cfg['batch'], 3, cfg['size'], cfg['size'], cfg['seed'])
tA, tB = torch.from_numpy(A), torch.from_numpy(B)

def dgrads():
    hip.optimizer_D.zero_grad()
    hip.backward_D()
    out = {k: p.grad.detach().cpu().numpy().copy() for k, p in hip.netD.named_parameters()}
    for i, d in enumerate(hip.netD_multiresolution):
        out.update({'mr%d.%s' % (i, k): p.grad.detach().cpu().numpy().copy() for k, p in d.named_parameters()})
    return out

for step in range(steps):
    if step > 0:
        sp._force(ref, hip)
    ref.optimize_parameters(tA, tB)
    hip.set_input({'A': tA, 'B': tB, 'A_paths': ['a'], 'B_paths': ['b']})
    hip.forward()
    hip.set_requires_grad([hip.netT, hip.netR], False)
    own = (hip.fake_TR_B, hip.fake_RT_B)
    hip.fake_TR_B = ref.fake_TR_B.detach().to(hip.device)
    hip.fake_RT_B = ref.fake_RT_B.detach().to(hip.device)
    want = dict(ref.grads_D)
    for i, g in enumerate(ref.grads_D_mr):
        want.update({'mr%d.%s' % (i, k): v for k, v in g.items()})
    res = {}
    for label, keys in (('default', ()), ('default again', ()), ('exact', ((20, 0), (24, 0)))):
        for k, v in keys:
            ops.tune(k, v)
        del log[:]
        res[label] = dgrads()
        if step == 0 and label != 'default again':
            print('--- routes, %s' % label)
            for e in log:
                print('   ', e)
        for k, v in keys:
            ops.tune(k, 1)
    print('=== step %d' % step)
    for k, v in want.items():
        vmax = float(v.abs().max())
        e = [sp._maxabs(res[l][k], v.numpy()) / max(vmax, 1e-30) for l in ('default', 'exact')]
        same = np.array_equal(res['default'][k], res['default again'][k])
        print('  %-28s max %.3e  err default %.2e exact %.2e  %s' % (k, vmax, e[0], e[1], '' if same else 'NOT REPRODUCIBLE'))
    hip.fake_TR_B, hip.fake_RT_B = own
    hip.optimizer_D.zero_grad()
    hip.backward_D()
    hip.optimizer_D.step()
    hip.set_requires_grad([hip.netT, hip.netR], True)
    hip.set_requires_grad([hip.netD, *hip.netD_multiresolution], False)
    hip.optimizer_R.zero_grad(); hip.optimizer_T.zero_grad()
    hip.backward_T_and_R()
    hip.optimizer_R.step(); hip.optimizer_T.step()
    hip.set_requires_grad([hip.netD, *hip.netD_multiresolution], True)
    torch.cuda.synchronize()
