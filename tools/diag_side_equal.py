"""Side stream vs single stream, run after run (diagnostic of tests/test_step_full_gpu.py::test_side_stream_equals_single_stream_on_every_config):
DIAG_CFG, DIAG_RUNS runs of DIAG_STEPS steps each in the order off, on, on, off, ...; every run is compared with the first (single-stream) one:
first step whose losses differ, and which optimizer buffers differ at the end."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nemar_amd import _lib  # noqa: E402
if os.environ.get('DIAG_LIB'):
    _lib.DEFAULT_PATH = os.path.abspath(os.environ['DIAG_LIB'])
import seeded  # noqa: E402
from nemar_amd import ops  # noqa: E402
from step_configs import FULL_CONFIGS, hw  # noqa: E402
import test_step_full_gpu as T  # noqa: E402

name = os.environ.get('DIAG_CFG', 'c3_full')
runs, steps = int(os.environ.get('DIAG_RUNS', '8')), int(os.environ.get('DIAG_STEPS', '50'))
cfg = FULL_CONFIGS[name]
A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
data = {'A': torch.from_numpy(A), 'B': torch.from_numpy(B), 'A_paths': [''], 'B_paths': ['']}


def run(side):
    ops.side_stream(side)
    m = T.build(name)
    ops.manual_seed(1234)
    losses, grads = [], []
    for s in range(steps):
        m.set_input(data)
        m.optimize_parameters()
        losses.append(tuple(sorted(m.get_current_losses().items())))
        if os.environ.get('DIAG_GRADS'):
            grads.append([o.flat_g.detach().clone() for o in m.optimizers])          # (device copies on the compute stream: no host sync)
    torch.cuda.synchronize()
    names = [n for o in ('T', 'R', 'D') for n in (o + '.p', o + '.m', o + '.v')]
    return losses, dict(zip(names, [t.detach().cpu().clone() for o in m.optimizers for t in (o.flat_p, o.m, o.v)])), grads, m


ref_l, ref_b, ref_g, _ = run(False)


def dbg_words():
    import ctypes
    dll = _lib.load()._dll
    if not hasattr(dll, 'nemar_wg_dbg'):
        return None
    out = (ctypes.c_uint * 8)()
    dll.nemar_wg_dbg(out)
    return list(out)


print('self-check words after the single-stream run:', dbg_words())
for r in range(1, runs):
    side = r % 4 in (1, 2) or bool(os.environ.get('DIAG_ALWAYS_SIDE'))
    l, b, g, m = run(side)
    first = next((i for i, (x, y) in enumerate(zip(l, ref_l)) if x != y), None)
    bad = [k for k in b if not torch.equal(b[k], ref_b[k])]
    msg = 'run %d (side %s) check %s: ' % (r, 'on ' if side else 'off', dbg_words())
    if first is None and not bad:
        print(msg + 'identical')
        continue
    print(msg + 'losses differ from step %s; buffers %s' % (first, bad))
    if first is not None:
        d = [(k, x, y) for (k, x), (_, y) in zip(l[first], ref_l[first]) if x != y]
        print('     step %d: %s' % (first, d))
    if g:
        done = False
        for s_ in range(steps):
            for j, nm in enumerate(('T', 'R', 'D')):
                x, y = g[s_][j].cpu(), ref_g[s_][j].cpu()
                if torch.equal(x, y):
                    continue
                net = getattr(m, 'net' + nm)
                base = m.optimizers[j].flat_g.data_ptr()
                for pn, p_ in net.named_parameters():
                    if p_.grad is None:
                        continue
                    o = (p_.grad.data_ptr() - base) // 4
                    dx, dy = x[o:o + p_.numel()].view(p_.shape), y[o:o + p_.numel()].view(p_.shape)
                    if not torch.equal(dx, dy):
                        idx = (dx != dy).nonzero()
                        dims = [sorted(set(int(i[d]) for i in idx)) for d in range(idx.shape[1])]
                        print('     step %d %s.%s %s: %d elements differ, max |diff| %.3e of max %.3e; index sets per dim: %s' % (
                            s_, nm, pn, tuple(p_.shape), idx.shape[0], float((dx - dy).abs().max()), float(dy.abs().max()),
                            [d if len(d) <= 12 else '%d values %d..%d' % (len(d), d[0], d[-1]) for d in dims]))
                done = True
            if done:
                break
