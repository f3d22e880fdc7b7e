"""the body of tests/test_step_gpu.py::test_step_graph_replay_equals_eager with per-step comparisons"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import seeded, step_parity
from nemar_amd import ops
from nemar_amd.models import create_model
from step_configs import STEP_CONFIGS, make_opt, hw
name = 'affine128'
cfg = STEP_CONFIGS[name]
a, b = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
data = {'A': torch.from_numpy(a), 'B': torch.from_numpy(b), 'A_paths': [''], 'B_paths': ['']}
def build():
    opt = make_opt(cfg, gpu_ids=[0]); opt.no_dropout = bool(os.environ.get('DIAG_NO_DROPOUT'))
    m = create_model(opt); m.setup(opt)
    step_parity.load_seeded_into(m.netT, cfg['seed'] + 1, cfg.get('overrides_T'))
    step_parity.load_seeded_into(m.netR, cfg['seed'] + 2, cfg.get('overrides_R'))
    step_parity.load_seeded_into(m.netD, cfg['seed'] + 3, cfg.get('overrides_D'))
    ops.manual_seed(1234); ops._step_params["step"] = 0
    return m
def snap(m):
    torch.cuda.synchronize()
    return ([o.flat_p.detach().cpu().clone() for o in m.optimizers], [o.m.detach().cpu().clone() for o in m.optimizers],
            [o.v.detach().cpu().clone() for o in m.optimizers], dict(m.get_current_losses()))
small = {'A': data['A'][:1], 'B': data['B'][:1], 'A_paths': [''], 'B_paths': ['']}
seq = [data, data, small, data]
variant = sys.argv[1] if len(sys.argv) > 1 else 'test'
ops.step_params(True, torch.device('cuda:0'))
m = build(); E = []; EG = None
for i, d in enumerate(seq):
    m.set_input(d); m.optimize_parameters(); E.append(snap(m))
    if i == 0: EG = m.optimizer_T.flat_g.detach().cpu().clone(); EB = m.fake_B.detach().cpu().clone(); ETR = m.fake_TR_B.detach().cpu().clone()
m = build()
if variant in ('test', 'before', 'b_eq', 'b_abs'):
    before = [o.flat_p.detach().cpu().clone() for o in m.optimizers]
m.set_input(data)
if not os.environ.get('DIAG_NO_GRAPH'):
    m.enable_step_graph(warmup=2)
else:
    ops.pin_workspaces(False)
if variant == 'test':
    torch.cuda.synchronize()
    assert all(torch.equal(x, o.flat_p.detach().cpu()) for x, o in zip(before, m.optimizers))
    assert all(float(o.m.abs().max()) == 0.0 and float(o.v.abs().max()) == 0.0 and o.step_count == 0 for o in m.optimizers)
if variant == 'poison':
    keep = []
    for sz in (128, 512, 2048, 8192, 32768, 131072, 70000, 69444, 45000, 300000, 1 << 20, 1 << 22, 1 << 24):
        for rep in range(6):
            keep.append(torch.full((sz,), float('nan'), device='cuda'))
    torch.cuda.synchronize()
if variant in ('b_eq', 'eq_abs'):
    torch.cuda.synchronize()
    _ = all(torch.equal(x, o.flat_p.detach().cpu()) for x, o in zip([o.flat_p.detach().cpu() for o in m.optimizers] if variant == 'eq_abs' else before, m.optimizers))
if variant in ('b_abs', 'eq_abs'):
    _ = all(float(o.m.abs().max()) == 0.0 and float(o.v.abs().max()) == 0.0 and o.step_count == 0 for o in m.optimizers)
if variant == 'after':
    torch.cuda.synchronize()
    assert all(float(o.m.abs().max()) == 0.0 and float(o.v.abs().max()) == 0.0 and o.step_count == 0 for o in m.optimizers)
if variant == 'after2':
    torch.cuda.synchronize()
    _ = [(float(o.m.abs().max()), float(o.v.abs().max())) for o in m.optimizers]
if variant == 'gc':
    import gc; gc.collect()
if variant == 'cpu':
    torch.cuda.synchronize(); _ = [o.flat_p.detach().cpu() for o in m.optimizers]
if variant == 'abs':
    _ = [float(o.m.abs().max()) for o in m.optimizers]
if variant == 'absT':
    _ = float(m.optimizers[0].m.abs().max())
if variant == 'alloc':
    _ = torch.zeros(69444, device='cuda'); torch.cuda.synchronize()
G = []
for i, d in enumerate(seq):
    m.set_input(d); m.optimize_parameters(); G.append(snap(m))
    if i == 0:
        GG = m.optimizer_T.flat_g.detach().cpu().clone()
        print('fake_B equal', bool(torch.equal(EB, m.fake_B.detach().cpu())), 'fake_TR_B equal', bool(torch.equal(ETR, m.fake_TR_B.detach().cpu())))
        for (k, p_), o_ in zip(m.netT.named_parameters(), m.optimizer_T.offsets):
            a_, b_ = EG[o_:o_ + p_.numel()], GG[o_:o_ + p_.numel()]
            print('   T grad %-28s equal %s  maxdiff %.3e  |g| %.3e' % (k, bool(torch.equal(a_, b_)), float((a_ - b_).abs().max()), float(a_.abs().max())))
for i, (e, g) in enumerate(zip(E, G)):
    if variant == 'poison' and i == 0:
        print('nan in params / m / v:', [[bool(torch.isnan(x).any()) for x in g[k]] for k in range(3)], 'nan in losses', {k: v for k, v in g[3].items() if v != v})
    print(variant, 'step', i + 1, [[bool(torch.equal(x, y)) for x, y in zip(e[k], g[k])] for k in range(3)], e[3] == g[3])
