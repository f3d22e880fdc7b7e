"""TEST INFRASTRUCTURE (child process of tests/test_product_lib_gpu.py::test_side_stream_soak_*): is the training step AS THE BENCH TIMES IT —
product library, dropout ON, weight-gradient branch on the side stream — bit-identical to the single-stream order, step after step?
One model, learning rate 0, the dropout stream re-seeded before every step: every step must produce the same three flat gradient buffers.
The single-stream order gives the reference; then STEPS steps with the side stream on, each compared with it ON THE DEVICE (no host round
trip: the host keeps running ahead as in training).  An event = a step whose gradients differ.  (The -m gpu form of tools/diag_step_events.py.)

    NEMAR_AB_LIBRARY=0 python tests/side_stream_soak.py c2_b8 300
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402

import seeded  # noqa: E402
from nemar_amd import _lib, ops  # noqa: E402
from step_configs import FULL_CONFIGS, hw, make_opt  # noqa: E402
import test_step_full_gpu as T  # noqa: E402


def main(name, steps):
    from nemar_amd.models import create_model
    lib = _lib.load()
    cfg = FULL_CONFIGS[name]
    opt = make_opt(cfg, gpu_ids=[0])
    opt.no_dropout = False                                   # the reference's default (models/nemar_model.py:102) and what bench.py times
    m = create_model(opt)
    m.setup(opt)
    T.load_seeded_into(m.netT, cfg['seed'] + 1, cfg.get('overrides_T'))
    T.load_seeded_into(m.netR, cfg['seed'] + 2, cfg.get('overrides_R'))
    T.load_seeded_into(m.netD, cfg['seed'] + 3, cfg.get('overrides_D'))
    for i, d in enumerate(m.netD_multiresolution):
        T.load_seeded_into(d, cfg['seed'] + 10 + i, cfg.get('overrides_D'))
    for o in m.optimizers:
        o.param_groups[0]['lr'] = 0.0
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    data = {'A': torch.from_numpy(A), 'B': torch.from_numpy(B), 'A_paths': [''], 'B_paths': ['']}

    def step():
        ops.manual_seed(1234)
        m.set_input(data)
        m.optimize_parameters()

    ops.side_stream(False)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ref = [o.flat_g.detach().clone() for o in m.optimizers]
    same = 0
    for _ in range(5):
        step()
        same += all(torch.equal(o.flat_g, r) for o, r in zip(m.optimizers, ref))
    dropped = bool(m.netT.training) and any(getattr(b, 'use_dropout', False) for b in m.netT.modules())
    ops.side_stream(True)
    events = torch.zeros((), dtype=torch.int64, device=ref[0].device)
    for k in range(steps):
        step()
        bad = torch.zeros((), dtype=torch.bool, device=ref[0].device)
        for o, r in zip(m.optimizers, ref):
            bad = bad | (o.flat_g != r).any()
        events += bad
        if k % 32 == 31:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(json.dumps({"library": os.path.basename(lib.path), "config": name, "dropout": dropped, "single_stream_repeats": same,
                      "steps": steps, "events": int(events)}))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
