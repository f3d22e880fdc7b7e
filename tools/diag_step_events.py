"""The second side-stream difference as a fast in-situ repro: ONE model, learning rate 0 and the dropout stream re-seeded before
every step, so every step must produce bit-identical gradients; each step's three flat gradient buffers are compared with a single-stream
reference on the device.  An EVENT = a step whose gradients differ.  Runs a list of experiments for DIAG_SECONDS each and prints events / steps.

    NEMAR_SIDE_STREAM=1 NEMAR_GY_HANDOVER=1 python tools/diag_step_events.py [experiments, comma separated]

Experiments: base (hand-over beside the side stream) | noho (hand-over off: the shipped default) | serial (the compute stream waits for every
wide weight gradient: same streams, same buffers, no concurrency) | sleep40 (the side stream idles ~40 us before a wide weight gradient) |
widest (only the wide layers' weight gradients on the side stream) | noplan (lazy weight packing)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
os.environ.setdefault('NEMAR_SIDE_STREAM', '1')
os.environ.setdefault('NEMAR_GY_HANDOVER', '1')
import torch  # noqa: E402
import seeded  # noqa: E402
from nemar_amd import _lib  # noqa: E402
if os.environ.get('DIAG_LIB'):
    _lib.DEFAULT_PATH = os.path.abspath(os.environ['DIAG_LIB'])
from nemar_amd import ops  # noqa: E402
from step_configs import FULL_CONFIGS, hw  # noqa: E402
import test_step_full_gpu as T  # noqa: E402

name = os.environ.get('DIAG_CFG', 'c3_full')
seconds = float(os.environ.get('DIAG_SECONDS', '40'))
exps = (sys.argv[1] if len(sys.argv) > 1 else 'base,serial,sleep40,noho').split(',')
cfg = FULL_CONFIGS[name]
A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
data = {'A': torch.from_numpy(A), 'B': torch.from_numpy(B), 'A_paths': [''], 'B_paths': ['']}
dev = torch.device('cuda', 0)
m = T.build(name)
for o in m.optimizers:
    o.param_groups[0]['lr'] = 0.0


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
print('%s: single stream repeats bit for bit in %d of 5 steps' % (name, same), flush=True)
assert same == 5

orig_wgrad = ops.L.conv2d_bwd_weight_ex
mode = {'serial': False, 'sleep': 0, 'widest': False}


def is_wide(a):
    return a[10] == 256 and a[1] == 256 and a[13] == 3            # K, C0, R


lags = []


def wgrad(*a):
    on_side = ops._lane[0] == 1
    if mode['sleep'] and on_side and is_wide(a):
        torch.cuda._sleep(mode['sleep'])
    if mode.get('lag') and on_side and is_wide(a) and len(lags) < 4000:
        em, es = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        em.record(torch.cuda.default_stream(dev))           # the compute stream, just behind the fork
        es.record(torch.cuda.current_stream(dev))           # the side stream, just before this weight gradient
        lags.append((em, es))
    r = orig_wgrad(*a)
    if mode.get('postgap') and on_side and is_wide(a):
        torch.cuda._sleep(mode['postgap'])
    if mode['serial'] and on_side and is_wide(a):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        mode.setdefault('pending', []).append(ev)
    if (mode.get('inwait') or mode.get('dgwait')) and on_side and is_wide(a):
        last_wide[0] = torch.cuda.Event()
        last_wide[0].record(torch.cuda.current_stream(dev))
    return r


ops.L.__dict__['conv2d_bwd_weight_ex'] = wgrad
orig_dgrad = ops.L.conv2d_bwd_data_ex


def dgrad(*a):
    if mode.get('dgwait') and last_wide[0] is not None:
        torch.cuda.current_stream(dev).wait_event(last_wide[0])          # the next data gradient only after the last wide weight gradient
        last_wide[0] = None
    return orig_dgrad(*a)


ops.L.__dict__['conv2d_bwd_data_ex'] = dgrad
last_wide = [None]
for nm in ('instnorm_bwd', 'instnorm_bwd_max'):
    def make(orig):
        def f(*a):
            if mode.get('inwait') and last_wide[0] is not None:
                torch.cuda.current_stream(dev).wait_event(last_wide[0])      # InstanceNorm backward only after the last wide weight gradient
                last_wide[0] = None
            return orig(*a)
        return f
    ops.L.__dict__[nm] = make(getattr(ops.L, nm))
orig_exit = ops._on_side.__exit__


def exit_(self, *a):
    r = orig_exit(self, *a)
    for ev in mode.pop('pending', []):
        torch.cuda.current_stream(dev).wait_event(ev)            # (back on the compute stream: wait for that weight gradient)
    return r


ops._on_side.__exit__ = exit_


def where(o, r, net):
    bad = (o.flat_g != r).nonzero().flatten()
    base = o.flat_g.data_ptr()
    out = []
    for pn, p_ in net.named_parameters():
        if p_.grad is None:
            continue
        lo = (p_.grad.data_ptr() - base) // 4
        sel = bad[(bad >= lo) & (bad < lo + p_.grad.numel())] - lo
        if sel.numel() == 0:
            continue
        if p_.grad.dim() == 4:
            Kk, Cc, Rr, Ss = p_.grad.shape
            k, c, t = sel // (Cc * Rr * Ss), sel // (Rr * Ss) % Cc, sel % (Rr * Ss)
            out.append('%s %s: %d elements, k %d..%d, c %d..%d, taps %s' % (pn, tuple(p_.grad.shape), sel.numel(), int(k.min()), int(k.max()), int(c.min()),
                                                                           int(c.max()), sorted(set(t.tolist()))))
        else:
            out.append('%s: %d elements' % (pn, sel.numel()))
    return out


for e in exps:
    mode.update(serial=False, sleep=0, widest=False, inwait=e == 'inwait', dgwait=e == 'dgwait', lag=e == 'lag',
                postgap=int(e[7:]) * 1750 if e.startswith('postgap') else 0)
    last_wide[0] = None
    ops.side_stream(True)
    ops._GY_HANDOVER = e != 'noho'
    if e == 'serial':
        mode['serial'] = True
    if e.startswith('sleep'):
        mode['sleep'] = int(e[5:]) * 1750           # ~1.75 GHz under load: cycles per microsecond
    steps = events = 0
    shown = 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        step()
        steps += 1
        ok = [torch.equal(o.flat_g, r) for o, r in zip(m.optimizers, ref)]
        if not all(ok):
            events += 1
            if shown < 4:
                shown += 1
                for o, r, net, good in zip(m.optimizers, ref, (m.netT, m.netD, m.netR), ok):      # (optimizer order: T, D, R)
                    if not good:
                        for line in where(o, r, net)[:4]:
                            print('    step %d: %s' % (steps, line), flush=True)
    print('experiment %-8s: %d events in %d steps (%.1f s)' % (e, events, steps, time.time() - t0), flush=True)
    if lags:
        torch.cuda.synchronize()
        ms = sorted(em.elapsed_time(es) * 1e3 for em, es in lags)
        print('    side-stream start of a wide weight gradient behind the compute stream\'s marker at the fork, us: min %.0f, p10 %.0f, median %.0f, p90 %.0f, max %.0f  (%d samples)'
              % (ms[0], ms[len(ms) // 10], ms[len(ms) // 2], ms[len(ms) * 9 // 10], ms[-1], len(ms)), flush=True)
        del lags[:]
