"""diagnostic: eager (device step parameters) vs hipGraph replay, step by step, with / without a ragged batch in the sequence"""
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
small = {'A': data['A'][:1], 'B': data['B'][:1], 'A_paths': [''], 'B_paths': ['']}


def build():
    opt = make_opt(cfg, gpu_ids=[0]); opt.no_dropout = False
    m = create_model(opt); m.setup(opt)
    step_parity.load_seeded_into(m.netT, cfg['seed'] + 1, cfg.get('overrides_T'))
    step_parity.load_seeded_into(m.netR, cfg['seed'] + 2, cfg.get('overrides_R'))
    step_parity.load_seeded_into(m.netD, cfg['seed'] + 3, cfg.get('overrides_D'))
    ops.manual_seed(1234); ops._step_params["step"] = 0
    return m


def snap(m):
    torch.cuda.synchronize()
    return [o.flat_p.detach().cpu().clone() for o in m.optimizers]


import itertools
for (label, seq), read in itertools.product((('no ragged', [data, data, data, data]), ('ragged 3rd', [data, data, small, data])), (False, True)):
    ops.step_params(True, torch.device('cuda:0'))
    m = build(); E = []
    for d in seq:
        m.set_input(d); m.optimize_parameters(); E.append(snap(m))
        if read: m.get_current_losses()
    m = build(); m.set_input(data); m.enable_step_graph(warmup=2); G = []
    for d in seq:
        m.set_input(d); m.optimize_parameters(); G.append(snap(m))
        if read: m.get_current_losses()
    ops.step_params(False); ops.pin_workspaces(False)
    for i, (e, g) in enumerate(zip(E, G)):
        print(label, 'read' if read else 'noread', 'step', i + 1, ['T', 'D', 'R'], [bool(torch.equal(x, y)) for x, y in zip(e, g)],
              [float((x - y).abs().max()) for x, y in zip(e, g)])
