"""Out-of-bounds writes into the shared scratch buffers?  Every workspace / arena handed to the library is a window of a larger buffer
whose margins hold a sentinel; after a step the margins are checked (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seeded
from nemar_amd import ops
from step_configs import FULL_CONFIGS, hw
import test_step_full_gpu
G = 4 << 20          # guard floats on either side (16 MB)
SENT = 12345.0
pools = {}


def guarded(kind, key, nfloats, device):
    ent = pools.get((kind, key))
    if ent is None or ent[1] < nfloats:
        big = torch.full((nfloats + 2 * G,), SENT, dtype=torch.float32, device=device)
        ent = pools[(kind, key)] = (big, nfloats)
    return ent[0][G:G + ent[1]]


def ws(nbytes, device):
    return guarded('ws', (device, ops._lane[0]), max(int(nbytes) // 4 + 64, 1 << 20), device)


orig_scratch = ops._conv_scratch


def scratch(N, H, W, K, C, R, S, stride, pad, device):
    need = ops.L.conv2d_scratch(N, H, W, K, C, R, S, stride, pad)
    if not need:
        return None
    return guarded('arena', (device, ops._lane[0]), int(need) // 4 + 64, device)


ops._workspace = ws
ops._conv_scratch = scratch
name = 'c2_full'
cfg = FULL_CONFIGS[name]
a, b = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
m = test_step_full_gpu.build(name)
for step in range(2):
    m.set_input({'A': torch.from_numpy(a), 'B': torch.from_numpy(b), 'A_paths': [''], 'B_paths': ['']})
    m.optimize_parameters()
    torch.cuda.synchronize()
    for (kind, key), (big, n) in pools.items():
        lo, hi = big[:G], big[G + n:]
        bl, bh = int((lo != SENT).sum()), int((hi != SENT).sum())
        msg = ''
        if bh:
            idx = (hi != SENT).nonzero().flatten()
            msg = '  first/last touched float beyond the end: %d .. %d' % (int(idx[0]), int(idx[-1]))
        if bl:
            idx = (lo != SENT).nonzero().flatten()
            msg += '  touched before the start: %d .. %d (of %d)' % (int(idx[0]), int(idx[-1]), G)
        print('step %d  %-6s lane %d  %9d floats: %d written below, %d written above%s' % (step, kind, key[1], n, bl, bh, msg))
