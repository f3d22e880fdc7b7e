"""TEST INFRASTRUCTURE (child process of tests/test_product_lib_gpu.py): STEPS training steps of one FULL_CONFIGS entry on whichever
build of the library the environment selects (NEMAR_AB_LIBRARY: 0 = the product libnemar_hip.so, 1 = the measurement build), on one
stream (bitwise reproducible run to run).  Prints one JSON line: the library's file name, whether it has switches, the per-step losses
as hex floats and a SHA-1 of each optimizer's parameter / moment buffers after the last step.

    NEMAR_AB_LIBRARY=0 python tests/product_step.py c2_full 3
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ["NEMAR_SIDE_STREAM"] = "0"

import torch  # noqa: E402

import seeded  # noqa: E402
from nemar_amd import _lib, ops  # noqa: E402
from step_configs import FULL_CONFIGS, hw  # noqa: E402
import test_step_full_gpu as T  # noqa: E402


def main(name, steps):
    lib = _lib.load()
    cfg = FULL_CONFIGS[name]
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    data = {'A': torch.from_numpy(A), 'B': torch.from_numpy(B), 'A_paths': [''], 'B_paths': ['']}
    ops.side_stream(False)
    m = T.build(name)
    ops.manual_seed(1234)
    losses = []
    for _ in range(steps):
        m.set_input(data)
        m.optimize_parameters()
        losses.append({k: float(v).hex() for k, v in sorted(m.get_current_losses().items())})
    torch.cuda.synchronize()
    digests = {}
    for nm, o in zip(('T', 'R', 'D'), m.optimizers):
        for tag, t in (('p', o.flat_p), ('m', o.m), ('v', o.v)):
            digests[nm + '.' + tag] = hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()
    print(json.dumps({"library": os.path.basename(lib.path), "has_switches": lib.has_switches, "losses": losses, "buffers": digests}))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
