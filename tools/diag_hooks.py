"""Which intermediate gradient of the T + R backward pass differs first from run to run?  (diagnostic for the side-stream branch)"""
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
for run in range(int(os.environ.get('DIAG_RUNS', '10'))):
    m = test_step_full_gpu.build(name)
    grads = {}

    live = {}

    def keep(tag):
        def hook(g):
            grads.setdefault(tag, []).append(g.detach().clone())
            live.setdefault(tag, []).append(g)          # the tensor itself (a second reference: the engine will not add into it in place)
        return hook

    R = m.netR
    o_predict, o_warp = R.predict, R.warp
    count = [0]

    def predict(*a_, **k_):
        f = o_predict(*a_, **k_)
        for i, t in enumerate(f if isinstance(f, (tuple, list)) else [f]):
            if torch.is_tensor(t) and t.requires_grad:
                t.register_hook(keep('field%d' % i))
        return f

    def warp(field, imgs):
        count[0] += 1
        fc = field[1].clone()
        fc.register_hook(keep('field_from_warp%d' % count[0]))
        out = o_warp((field[0], fc), imgs)
        for i, t in enumerate(out):
            if t.requires_grad:
                t.register_hook(keep('warp%d_out' % count[0]))
        for i, t in enumerate(imgs):
            if t.requires_grad:
                t.register_hook(keep('warp%d_in' % count[0]))
        return out

    o_reg = R.regularization

    def reg(field, wf):
        fc = field[0].clone()
        fc.register_hook(keep('field_from_reg'))
        return o_reg((fc, field[1]), wf)

    R.predict, R.warp, R.regularization = predict, warp, reg
    m.set_input({'A': torch.from_numpy(a), 'B': torch.from_numpy(b), 'A_paths': [''], 'B_paths': ['']})
    m.optimize_parameters()
    torch.cuda.synchronize()
    for k in list(grads):
        for i, (c, l) in enumerate(zip(grads[k], live[k])):
            if not torch.equal(c, l):
                print('   run %d: %s[%d]: the clone taken in the hook differs from the tensor re-read after the step: %d elements'
                      % (run, k, i, int((c != l).sum())))
    cur = {k: [t.cpu() for t in v] for k, v in grads.items()}
    cur.update({k + '_reread': [t.cpu() for t in v] for k, v in live.items()})
    cur['R.grad'] = [m.optimizer_R.flat_g.detach().cpu().clone()]
    cur['T.grad'] = [m.optimizer_T.flat_g.detach().cpu().clone()]
    if ref is None:
        ref = cur
        print('tags:', {k: [tuple(t.shape) for t in v] for k, v in cur.items()})
        continue
    bad = []
    for k in cur:
        for i, (x, y) in enumerate(zip(cur[k], ref[k])):
            if not torch.equal(x, y):
                bad.append('%s[%d]: %d of %d differ, max %.2e' % (k, i, int((x != y).sum()), x.numel(), float((x - y).abs().max())))
    print('run %d: %s' % (run, '; '.join(bad) if bad else 'identical'))
    if bad and 'field_from_warp1' in cur:
        x, y = cur['field_from_warp1'][0], ref['field_from_warp1'][0]
        idx = (x != y).nonzero()
        ys = sorted(set(int(i[2]) for i in idx)); xs = sorted(set(int(i[3]) for i in idx)); cs = sorted(set(int(i[1]) for i in idx))
        print('   planes', cs, 'rows', ys[:40], 'cols', xs[:60])
        for i in idx[:6]:
            c_, y_, x_ = int(i[1]), int(i[2]), int(i[3])
            print('   (%d,%d,%d): %.6f vs %.6f' % (c_, y_, x_, float(x[0, c_, y_, x_]), float(y[0, c_, y_, x_])))
        print('   NaN among the differing elements: %d of %d' % (int(torch.isnan(x[x != y]).sum()), int((x != y).sum())))
