"""`-m gpu`: FULL-WIDTH NEMARModel.optimize_parameters() (ngf = ndf = 64, resnet_9blocks: the production kernel
dispatch — 128x128 wave-specialised tiles, one-round weight-gradient splits, vector loaders) at the BASELINE.json
shapes C2..C5, against fixtures recorded from the reference itself (tests/golden/make_golden.py --only full: the
reference's NEMARModel run in fp32 AND fp64 on the same seeded weights / inputs).

Every compared quantity q obeys   |q_build - q_ref64| <= base(q) + 4 * |q_ref32 - q_ref64| :
the fp64 run is the true value of the reference's algorithm, and the reference's own fp32-vs-fp64 gap measures how
ill-conditioned q is (sign() in the L1 gradient, ReLU / LeakyReLU / max-pool masks, floor() in the sampler).  base(q) is
the fp32 rounding floor of the quantity's class, written below.  Two refinements, both measured on the first full-width
run (gpurun_out/r2a/full_rows.txt: every loss / image / weight-gradient row inside the bound, the build's gradient errors
statistically equal to the reference's own fp32 gaps — R net median 6e-4 vs 1.7e-3):
  * one tensor's own |f32 - f64| is a single draw of a heavy-tailed quantity, so a gradient row uses
    max(own gap, 90th-percentile relative gap of its network) — otherwise a tensor whose reference run happened to land close
    (R's 2-element output bias: 6e-4 where its weight shows 1.8e-3) gets a bound tighter than its conditioning;
  * post-Adam checksums: the first Adam step moves every element by lr * sign(g), so an element whose gradient is at
    rounding distance of zero moves by 2 lr = 4e-4 between ANY two fp32 implementations; three such elements are
    allowed per tensor on top of the relative bound, and biases in front of an InstanceNorm (true gradient exactly zero,
    i.e. pure rounding noise with a random sign) are not compared.
c1_full is BASELINE config 1 at full width (affine STN, resnet_6blocks, 128x128, batch 1), c2_b8 the bench workload at the bench
batch (8: the batched T / D passes launch their kernels at batch 16 / 24, as the timed step does).
The 1024x1024 config has no fp64 run (one fp64 step of the reference needs > 60 GB: it was tried, tests/golden/make_golden.py
died in a 26 GB allocation), so there the build is compared with the reference's FP32 run — both sides carry rounding error —
with the gap per quantity class taken from the 512x512 config's measured relative gaps (90th percentile of the class), tripled
for the gradient classes: the registration net of that config is two levels deeper and sees 4x the pixels."""
import os

import numpy as np
import pytest
import torch

import seeded
from full_record import full_step_record
from step_configs import FULL_CONFIGS, make_opt, hw
from step_parity import load_seeded_into

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# class -> (relative base, absolute base); gradients are relative to the tensor's own norm
BASE = {
    'loss': (1e-4, 1e-6), 'reg': (1e-4, 1e-7), 'mean': (0.0, 2e-5), 'absmean': (2e-5, 2e-6), 'proj': (0.0, 2e-4),
    'crop0': (0.0, 5e-5), 'cropc': (0.0, 5e-5), 'offsets': (2e-5, 2e-6),
    'gradnorm': (3e-3, 0.0), 'gradproj': (3e-3, 0.0), 'gradmax': (1e-2, 0.0), 'psum': (5e-5, 0.0), 'pabs': (5e-5, 0.0),
}


def _class_of(key):
    return key.split('/')[0]


def _rel_gaps_by_class(g):
    """max over the scalar quantities of a class of |f32 - f64| / scale (for fixtures recorded with both runs)."""
    out = {}
    for k in g.files:
        if not k.startswith('f64/'):
            continue
        q = k[4:]
        a, b = g['f32/' + q], g[k]
        cls = _class_of(q)
        if cls.startswith('grad'):
            scale = float(g['f64/gradnorm/' + q.split('/', 1)[1]])
            if scale < 1e-5 * _net_gmax(g, q.split('/')[1], 'f64'):
                continue
        elif cls in ('psum', 'pabs'):
            scale = float(g['f64/pabs/' + q.split('/', 1)[1]])
        else:
            scale = max(float(np.abs(b).max()), 1.0 if cls in ('loss', 'crop0', 'cropc', 'mean', 'proj') else 1e-30)
        out.setdefault(cls, []).append(float(np.abs(a - b).max()) / max(scale, 1e-30))
    return {c: float(np.quantile(v, 0.9)) for c, v in out.items()}


def _f32_tags(g):
    """'f32' and, where the fixture holds them, the fp32 runs of the reference on inputs moved by +-4 ulps ('f32p1', 'f32p2': make_golden.py)"""
    return ['f32'] + sorted({k.split('/')[0] for k in g.files if k.startswith('f32p')})


def _gap(g, q):
    """largest |fp32 run - fp64 run| of quantity q over the reference's fp32 runs"""
    want = g['f64/' + q]
    return max(float(np.abs(g['%s/%s' % (t, q)] - want).max()) for t in _f32_tags(g) if '%s/%s' % (t, q) in g.files)


def _net_grad_gap(g, net):
    """90th percentile over the weight tensors of one network of |gradnorm_f32 - gradnorm_f64| / gradnorm_f64 (largest over the fp32 runs)"""
    pre = 'f64/gradnorm/%s/' % net
    rel = [_gap(g, k[4:]) / max(float(g[k]), 1e-30) for k in g.files if k.startswith(pre) and k.endswith('weight')]
    return float(np.quantile(rel, 0.9)) if rel else 0.0


LR = 2e-4


def _net_gmax(g, net, tag):
    pre = '%s/gradnorm/%s/' % (tag, net)
    return max(float(g[k]) for k in g.files if k.startswith(pre))


def build(name):
    from nemar_amd.models import create_model
    cfg = FULL_CONFIGS[name]
    opt = make_opt(cfg, gpu_ids=[0])
    m = create_model(opt)
    m.setup(opt)
    load_seeded_into(m.netT, cfg['seed'] + 1, cfg.get('overrides_T'))
    load_seeded_into(m.netR, cfg['seed'] + 2, cfg.get('overrides_R'))
    load_seeded_into(m.netD, cfg['seed'] + 3, cfg.get('overrides_D'))
    for i, d in enumerate(m.netD_multiresolution):
        load_seeded_into(d, cfg['seed'] + 10 + i, cfg.get('overrides_D'))
    return m


def compare(name, rec, report=None, fan_in=None):
    g = np.load(os.path.join(GOLD, 'step_%s.npz' % name))
    have64 = any(k.startswith('f64/') for k in g.files)
    truth = 'f64' if have64 else 'f32'
    class_gap = None
    if not have64:
        class_gap = _rel_gaps_by_class(np.load(os.path.join(GOLD, 'step_c4_full.npz')))
    rows = []
    for k in sorted(g.files):
        if not k.startswith(truth + '/'):
            continue
        q = k[len(truth) + 1:]
        if q not in rec:
            continue
        want = g[k]
        got = np.asarray(rec[q], dtype=np.float64)
        cls = _class_of(q)
        rel, ab = BASE[cls]
        if cls.startswith('grad') or cls in ('psum', 'pabs'):
            tail = q.split('/', 1)[1]
            gk = '%s/gradnorm/%s' % (truth, tail)
            if gk in g.files and float(g[gk]) < 1e-5 * _net_gmax(g, tail.split('/')[0], truth):
                continue              # conv biases in front of InstanceNorm: exactly-zero gradient + rounding noise
        if cls.startswith('grad'):
            scale = float(g['%s/gradnorm/%s' % (truth, tail)])
        elif cls in ('psum', 'pabs'):
            scale = float(g['%s/pabs/%s' % (truth, tail)])
            ab = 3 * 2 * LR
            # an nn.Linear in front of a ReLU (the affine STN's regressor): ONE hidden unit whose pre-activation, or whose six-term
            # gradient sum, sits at rounding distance of zero takes the other Adam step with its WHOLE weight row — fan_in elements
            # of 2 lr each.  Two such units are allowed (measured: default_full, one unit, 10.2 of a row sum of 11.1; the reference's
            # three fp32 runs happen to have none)
            if fan_in and tail in fan_in:
                ab += 2 * fan_in[tail] * 2 * LR
        else:
            scale = float(np.abs(want).max())
        if have64:
            gap = _gap(g, q)
            if cls.startswith('grad'):
                gap = max(gap, _net_grad_gap(g, tail.split('/')[0]) * scale)
        else:
            gap = class_gap.get(cls, 0.0) * max(scale, 1.0 if cls in ('loss', 'crop0', 'cropc', 'mean', 'proj') else 0.0)
            if cls.startswith('grad'):     # a projection / max of a tiny tensor moves as much as its norm does
                gap = 3.0 * scale * max(class_gap.get(c, 0.0) for c in ('gradnorm', 'gradproj', 'gradmax'))
        tol = rel * scale + ab + 4.0 * gap
        err = float(np.abs(got - want).max())
        rows.append((q, err, tol, err <= tol))
    if report:
        with open(report, 'a') as f:
            f.write('== %s (%d rows, truth = reference %s)\n' % (name, len(rows), truth))
            for r in rows:
                f.write('%-86s err=%.3e tol=%.1e %s\n' % (r[0], r[1], r[2], 'ok' if r[3] else 'FAIL'))
    return rows


ROUTE_NAMES = {0: 'exact', 1: 'narrow', 2: 'split16', 3: 's16g', 4: 'k7'}


class _RouteLog:
    """Which kernel family every convolution-type C-ABI call of a step took (nemar_last_route()), written into the test report:
    the reference's default geometry (288 x 384) and the non-square unet case land on route combinations no square fixture exercises."""

    def __init__(self):
        from nemar_amd import ops
        self.L, self.rows, self.saved = ops.L, {}, {}

    def __enter__(self):
        L = self.L
        # (the product path calls the _ex entry points — side inputs per call; ConvTranspose2d the plain ones: same argument positions)
        for nm, dims in (('conv2d_fwd', (7, 8, 9, 10, 11, 13, 14)), ('conv2d_bwd_data', (9, 10, 11, 12, 15, 17, 18)),
                         ('conv2d_bwd_weight', (7, 8, 9, 10, 13, 15, 16)), ('conv2d_fwd_ex', (7, 8, 9, 10, 11, 13, 14)),
                         ('conv2d_bwd_data_ex', (9, 10, 11, 12, 15, 17, 18)), ('conv2d_bwd_weight_ex', (7, 8, 9, 10, 13, 15, 16))):
            f = getattr(L, nm)
            self.saved[nm] = f

            def wrapped(*a, _f=f, _nm=nm[:-3] if nm.endswith('_ex') else nm, _dims=dims):
                r = _f(*a)
                key = (_nm,) + tuple(a[i] for i in _dims) + (ROUTE_NAMES.get(L.last_route(), '?'),)
                self.rows[key] = self.rows.get(key, 0) + 1
                return r
            setattr(L, nm, wrapped)
        return self

    def __exit__(self, *exc):
        for nm, f in self.saved.items():
            setattr(self.L, nm, f)

    def routes_of(self, op, **want):
        """the set of routes taken by the calls of `op` whose (N, H, W, K, R, stride, pad) match the given fields"""
        names = ('N', 'H', 'W', 'K', 'R', 'stride', 'pad')
        return {k[8] for k in self.rows if k[0] == op and all(k[1 + names.index(f)] == v for f, v in want.items())}

    def write(self, name, report):
        if not report:
            return
        with open(report, 'a') as f:
            f.write('== %s: routes of the convolution calls (op, N, H, W, K, R|OH, stride, pad -> route x count)\n' % name)
            for k, c in sorted(self.rows.items(), key=lambda kv: (kv[0][0], -kv[0][2] * kv[0][3], kv[0][4])):
                f.write('   %-18s N=%-3d H=%-4d W=%-4d K=%-4d R=%-3d stride=%d pad=%d  -> %-8s x %d\n' % (k[:8] + (k[8], c)))


@pytest.mark.parametrize("name", list(FULL_CONFIGS))
def test_full_width_step_vs_reference(name):
    cfg = FULL_CONFIGS[name]
    m = build(name)
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    with _RouteLog() as routes:
        rec = full_step_record(m, A, B, cfg['seed'])
    torch.cuda.synchronize()
    routes.write(name, os.environ.get('NEMAR_FULL_REPORT'))
    # the kernel families the fixture's shapes are SUPPOSED to exercise (a silent change of dispatch would leave a route unpinned)
    want_routes = {
        'c2_full': [('conv2d_fwd', dict(H=64, W=64, K=256, R=3, stride=1), {'split16'}),
                    ('conv2d_bwd_data', dict(H=64, W=64, K=256, R=3, stride=1), {'split16'}),
                    ('conv2d_bwd_weight', dict(H=64, W=64, K=256, R=3, stride=1), {'split16'}),
                    ('conv2d_fwd', dict(R=7), {'k7'}), ('conv2d_bwd_weight', dict(R=7), {'k7'})],
        # 72 x 96 / 64 x 96 residual-block maps: rows are not a power of two -> off the wide route's forward / data gradient (the general
        # in-kernel-split kernels serve them), its weight gradient still applies; 7x7 layers on the non-aligned strips of conv_k7.hip
        'default_full': [('conv2d_fwd', dict(H=72, W=96, K=256, R=3, stride=1), {'s16g'}),
                         ('conv2d_bwd_data', dict(H=72, W=96, K=256, R=3, stride=1), {'s16g'}),
                         ('conv2d_bwd_weight', dict(H=72, W=96, K=256, R=3, stride=1), {'split16'}),
                         ('conv2d_fwd', dict(R=7), {'k7'}), ('conv2d_bwd_data', dict(R=7), {'k7'}), ('conv2d_bwd_weight', dict(R=7), {'k7'})],
        'c2_256x384': [('conv2d_fwd', dict(H=64, W=96, K=256, R=3, stride=1), {'s16g'}),
                       ('conv2d_bwd_weight', dict(H=64, W=96, K=256, R=3, stride=1), {'split16'}),
                       ('conv2d_fwd', dict(H=256, W=384, K=32, R=3), {'s16g'}),
                       ('conv2d_fwd', dict(R=7), {'k7'}), ('conv2d_bwd_weight', dict(R=7), {'k7'})],
    }
    for op, sel, want in want_routes.get(name, []):
        got = routes.routes_of(op, **sel)
        assert got == want, (name, op, sel, got, want)
    fan_in = {'R/' + k: int(p.shape[1]) for k, p in m.netR.named_parameters() if p.dim() == 2}      # (Linear weights)
    rows = compare(name, rec, report=os.environ.get('NEMAR_FULL_REPORT'), fan_in=fan_in)
    assert len(rows) > (200 if cfg['stn_type'] == 'affine' else 300), len(rows)     # (the affine STN has 8 parameter tensors)
    bad = [r for r in rows if not r[3]]
    assert not bad, (len(bad), bad[:8])


def test_full_width_step_on_the_exact_fp32_route():
    """The same C2 fixture with every 16-bit-pipe kernel switched off (nemar_tune(20, 0) and (24, 0): all convolutions on the
    exact-fp32 MFMA / VALU kernels) — both arithmetic routes stay pinned to the reference."""
    from nemar_amd import ops
    name = 'c2_full'
    cfg = FULL_CONFIGS[name]
    ops.tune(20, 0)
    ops.tune(24, 0)
    try:
        m = build(name)
        A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
        rec = full_step_record(m, A, B, cfg['seed'])
        torch.cuda.synchronize()
    finally:
        ops.tune(20, 1)
        ops.tune(24, 1)
    rows = compare(name, rec)
    bad = [r for r in rows if not r[3]]
    assert len(rows) > 300 and not bad, (len(bad), bad[:8])


def test_registration_submodel_of_config5_vs_fp64():
    """BASELINE config 5's stress path with an fp64 truth of its own: the deep registration net + large-field warp + smoothness at
    1024 x 1024 (tests/golden/regsub_c5_full.npz: the reference's UnetSTN run in fp32 and fp64 by make_golden.py --only regsub), same
    bound as the full-width steps:  |build - ref64| <= base + 4 |ref32 - ref64|, gradients with the net's 90th-percentile gap."""
    from full_record import registration_record
    from nemar_amd import ops
    from nemar_amd.models import stn
    path = os.path.join(GOLD, 'regsub_c5_full.npz')
    if not os.path.exists(path):
        pytest.skip('regsub_c5_full.npz not generated')
    g = np.load(path)
    cfg = FULL_CONFIGS['c5_full']
    net = stn.define_stn(make_opt(cfg, gpu_ids=[0]), 'unet')
    load_seeded_into(net, cfg['seed'] + 2, cfg.get('overrides_R'))
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    rec = registration_record(net, ops.l1_loss, A, B, cfg['seed'], 100.0, cfg['lambda_smooth'])
    torch.cuda.synchronize()
    net_gap = _net_grad_gap(g, 'R')
    gmax = _net_gmax(g, 'R', 'f64')
    base = dict(BASE)
    rows = []
    for k in sorted(g.files):
        if not k.startswith('f64/'):
            continue
        q = k[4:]
        want, got = g[k], np.asarray(rec[q], dtype=np.float64)
        cls = _class_of(q)
        rel, ab = base[cls]
        if cls.startswith('grad'):
            scale = float(g['f64/gradnorm/' + q.split('/', 1)[1]])
            if scale < 1e-5 * gmax:
                continue
        else:
            scale = float(np.abs(want).max())
        gap = float(np.abs(g['f32/' + q] - want).max())
        if cls.startswith('grad'):
            gap = max(gap, net_gap * scale)
        tol = rel * scale + ab + 4.0 * gap
        err = float(np.abs(got - want).max())
        rows.append((q, err, tol, err <= tol))
    report = os.environ.get('NEMAR_FULL_REPORT')
    if report:
        with open(report, 'a') as f:
            f.write('== registration sub-model of c5_full (%d rows, truth = reference f64)\n' % len(rows))
            for r in rows:
                f.write('%-86s err=%.3e tol=%.1e %s\n' % (r[0], r[1], r[2], 'ok' if r[3] else 'FAIL'))
    bad = [r for r in rows if not r[3]]
    assert len(rows) > 150 and not bad, (len(rows), len(bad), bad[:8])


SIDE_STEPS = 50


@pytest.mark.parametrize("name", list(FULL_CONFIGS))
def test_side_stream_equals_single_stream_on_every_config(name):
    """The weight-gradient branch of EVERY convolution runs on a side HIP stream (ops._on_side).  Same bits as the single-stream order, on
    every full-width configuration (different kernel mixes and timings: 128^2 ... 1024^2, affine / unet / deep STN, multi-resolution D,
    non-square maps), over SIDE_STEPS consecutive steps each (the fixtures' option set: no dropout in T): parameters, both Adam moments and the loss trajectory.
    Round 4 saw one kernel pair go wrong here (the 7x7 stem's weight gradient next to grid_sample's grid gradient); the cause was
    a packed-FP32 instruction form that miscomputes next to another kernel's MFMAs — the library has none any more
    (tests/test_abi.py checks the ISA), and this test is the end-to-end guard."""
    from nemar_amd import ops
    cfg = FULL_CONFIGS[name]
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    data = {'A': torch.from_numpy(A), 'B': torch.from_numpy(B), 'A_paths': [''], 'B_paths': ['']}

    def run(side):
        ops.side_stream(side)
        m = build(name)
        ops.manual_seed(1234)            # (the model seeded the dropout generator from torch's seed: the same stream of masks in both runs)
        losses = []
        for _ in range(SIDE_STEPS):
            m.set_input(data)
            m.optimize_parameters()
            losses.append(tuple(sorted(m.get_current_losses().items())))
        torch.cuda.synchronize()
        return losses, [t.detach().cpu().clone() for o in m.optimizers for t in (o.flat_p, o.m, o.v)]

    prev = ops.side_stream(None)
    try:
        want_l, want = run(False)
        got_l, got = run(True)
    finally:
        ops.side_stream(prev)
    first = next((i for i, (x, y) in enumerate(zip(got_l, want_l)) if x != y), None)
    assert first is None, ('losses differ from step %d on' % first, got_l[first], want_l[first])
    for k, (x, y) in enumerate(zip(got, want)):
        assert torch.equal(x, y), ('buffer %d' % k, int((x != y).sum()), float((x - y).abs().max()))


def test_ten_step_trajectory_vs_reference():
    """TEN consecutive free-running steps at the bench configuration's width (c2_full), against the reference's own fp32 and fp64
    trajectories (tests/golden/step_c2_traj10.npz, make_golden.py --only traj): losses, regulariser, deformation-field statistics and
    the translated image per step, parameter sums at the end.  The one-step fixtures start from seeded uniform weights; here the
    fp16 x 3 scales, the packed-weight refresh and the Adam moments run on weights the optimizer has moved, step after step.
    Tolerance per quantity and step: base + 4 x the largest |reference fp32 run - reference fp64 run| AT THAT STEP over the fixture's
    three fp32 runs (the plain one and two on inputs moved by +-4 ulps): a free-running GAN trajectory is chaotic — Adam's first steps
    move every weight by +-lr, so the sign of a gradient element at rounding distance of zero decides — and how fast two correct fp32
    runs drift apart is what those runs measure."""
    from full_record import traj_record
    g = np.load(os.path.join(GOLD, 'step_c2_traj10.npz'))
    steps = 1 + max(int(k.split('/')[1][1:]) for k in g.files if k.split('/')[1].startswith('s'))
    cfg = FULL_CONFIGS['c2_full']
    m = build('c2_full')
    A, B = seeded.seeded_images(cfg['batch'], 3, *hw(cfg), cfg['seed'])
    rec = traj_record(m, A, B, cfg['seed'], steps)
    torch.cuda.synchronize()
    base = {'loss': (1e-4, 1e-6), 'reg': (1e-4, 1e-7), 'offsets': (2e-5, 2e-6), 'absmean': (2e-5, 2e-6), 'proj': (0.0, 2e-4),
            'psum': (0.0, 0.0), 'pabs': (0.0, 0.0)}
    bad, rows = [], []
    for k in sorted(g.files):
        if not k.startswith('f64/'):
            continue
        q = k[4:]
        cls = q.split('/')[1]
        want, got = float(g[k]), float(rec[q])
        gap = _gap(g, q)
        rel, ab = base[cls]
        if cls in ('psum', 'pabs'):
            ab = 3 * 2 * LR * steps * 50        # (a few hundred elements per net may take the other Adam sign step in some of the steps)
        tol = rel * max(abs(want), 1.0 if cls in ('loss', 'proj') else 0.0) + ab + 4.0 * gap
        err = abs(got - want)
        rows.append('%-40s err=%.3e tol=%.1e max|ref32-ref64|=%.1e %s' % (q, err, tol, gap, 'ok' if err <= tol else 'FAIL'))
        if err > tol:
            bad.append(rows[-1])
    report = os.environ.get('NEMAR_FULL_REPORT')
    if report:
        with open(report, 'a') as f:
            f.write('== c2_full, %d-step trajectory (%d rows)\n' % (steps, len(rows)) + '\n'.join(rows) + '\n')
    assert len(rows) >= 13 * steps
    assert not bad, bad[:10]
