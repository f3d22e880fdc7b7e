"""bench.py — NeMAR training-step throughput on MI355X (BASELINE.json metric: train images/sec, 256x256 A/B pairs).

  python bench.py --gpus N --steps K --warmup W        (N>1: one rank per GPU over RCCL — under torch.distributed.run, or, started
                                                        directly, bench.py spawns its N ranks itself: nemar_amd/launch.py)

A step = one NEMARModel.optimize_parameters() (forward, discriminator update, translation+registration update, three
fused Adam steps) on one synthetic batch already resident in HBM.  Workload = BASELINE.json configs[1]:
`--stn_type unet --stn_cfg A`, resnet_9blocks translation net, basic PatchGAN, 256x256, batch 8 per GPU, dropout ON
(the reference default), lambda_smooth 10 — fp32 end to end (the reference's precision; no reduced-precision mode).
N GPUs = N independent batch shards (weak scaling) + gradient all-reduce(avg) over RCCL.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline            dominant kernel = igemm_split16_kernel<2,2,3>: forward and data gradient of the 256->256 3x3 reflect
                      layers of the translation net's residual blocks (36 batch-16 launches per step, 17 % of the step), fp32
                      products as three fp16 partial products on the 16-bit MFMA.  achieved = algorithmic (fp32-equivalent) FLOP /
                      launch duration, measured with HIP events on the launch stream inside the library (nemar_kernel_timer) in
                      two extra steps right after the timed region; peak = 2500 TF dense fp16 MFMA / 3 products;
                      traffic = FETCH_SIZE x 2 + WRITE_SIZE of the same batch-16 launch from separate rocprofv3 --pmc passes
                      (profiles/r6_pmc_traffic.json).  Under this kernel the chip clocks at ~1.75 GHz, not the 2.4 GHz the
                      peak assumes (profiles/r3_clock_trace.txt); a pure MFMA loop on random operands sustains 1730 TFLOP/s,
                      on all-zero operands 2530 (profiles/r4_mfma_peak_modes.txt): `frac_of_sustained_mfma` is priced on the former
  roofline_grid_sample   BASELINE's second metric: grid_sample fwd+bwd algorithmic bytes / event-timed duration vs 8 TB/s
  launch / launch_probe  the timed steps are eager launches (weight-gradient branch on a side stream, nemar_amd/ops.py _on_side) or replays of one
                      captured hipGraph, whichever a short pre-run probe measured faster (--graph on|off forces either)
  cpu_baseline        the CPU oracle (oracle/torch_ref.py, a proven-equal restatement of the reference's step) timed
                      on this box's host cores on a bounded sample (config-2 shape at batch 1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TF = 2500.0    # MI355X_MICROARCH.md: dense FP16/BF16 MFMA peak (32x32x16)
F16_MFMA_SUSTAINED_TF = 1730.0   # measured on the MI355X of this pool: back-to-back v_mfma_f32_32x32x16_f16 on every SIMD, no LDS / memory,
                                 # RANDOM operands, 2 waves / SIMD: the clock settles at ~1.75 GHz.  The same loop on all-zero / all-one
                                 # operands holds 2.37-2.43 GHz = 2450-2530 TFLOP/s — the guide's peak is a trivial-operand figure, the
                                 # power limit depends on the data (tools/probes/mfma_peak_modes.hip, profiles/r4_mfma_peak_modes.txt)
PMC_FILE = "profiles/r6_pmc_traffic.json"


def build_opt(batch, size, extra=()):
    from nemar_amd.options import TrainOptions
    argv = ['--stn_type', 'unet', '--stn_cfg', 'A', '--netG', 'resnet_9blocks', '--img_height', str(size),
            '--img_width', str(size), '--batch_size', str(batch), '--lambda_smooth', '10.0',
            '--checkpoints_dir', '/tmp/nemar_bench_ck', '--name', 'bench', '--gpu_ids', '0', *extra]
    return TrainOptions().parse(argv, quiet=True)


class KernelTimer:
    """HIP-event brackets around selected C-ABI launches (on torch's current stream == the launch stream)."""

    def __init__(self):
        self.spans = {}
        self.enabled = False

    def bracket(self, tag):
        timer = self

        class _Span:
            def __enter__(self_s):
                self_s.on = timer.enabled
                if self_s.on:
                    self_s.e0 = torch.cuda.Event(enable_timing=True)
                    self_s.e1 = torch.cuda.Event(enable_timing=True)
                    self_s.e0.record()

            def __exit__(self_s, *a):
                if self_s.on:
                    self_s.e1.record()
                    timer.spans.setdefault(tag, []).append((self_s.e0, self_s.e1))
        return _Span()

    def summary(self):
        torch.cuda.synchronize()
        return {tag: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v) * 1e-3) for tag, v in self.spans.items()}


def _cpu_model(stn_type, n_blocks, size):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import seeded
    from oracle import torch_ref as R
    from nemar_amd.models import networks, stn
    opt = build_opt(1, size, ['--no_dropout', '--stn_type', stn_type])
    torch.manual_seed(0)
    netT = networks.define_G(3, 3, 64, 'resnet_%dblocks' % n_blocks, 'instance', False, 'normal', 0.02, [])
    netR = stn.define_stn(argparse.Namespace(**{**vars(opt), 'gpu_ids': []}), stn_type)
    netD = networks.define_D(6, 64, 'basic', 3, 'instance', 'normal', 0.02, [])
    sd = lambda n: {k: v.detach().clone() for k, v in n.state_dict().items()}
    m = R.RefModel(sd(netT), sd(netR), sd(netD), n_blocks=n_blocks, stn_type=stn_type, lambda_smooth=10.0)
    A, B = seeded.seeded_images(1, 3, size, size, 1)
    return m, torch.from_numpy(A), torch.from_numpy(B)


def _cpu_time(m, A, B, steps):
    t0 = time.time()
    for _ in range(steps):
        m.optimize_parameters(A, B)
    return (time.time() - t0) / steps


def _physical_cores():
    """(socket, core) pairs of /proc/cpuinfo; falls back to the logical count"""
    try:
        pairs, phys = set(), None
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('physical id'):
                phys = ln.split(':')[1].strip()
            elif ln.startswith('core id'):
                pairs.add((phys, ln.split(':')[1].strip()))
        return len(pairs) or os.cpu_count()
    except OSError:
        return os.cpu_count()


def cpu_baseline():
    """The CPU oracle (a restatement of the reference's step proven equal to it by tests/test_oracle_golden.py) timed on this
    box's host cores, as SURVEY.md §8d asks: BASELINE config 1 (the reference's own CPU-runnable case: affine STN,
    resnet_6blocks, 128x128, batch 1) always, and the config-2 shape (the GPU workload) at batch 1.  Both without dropout
    (the oracle draws no masks; dropout is a negligible share of CPU time) against the GPU's batch 8 with dropout on — the
    CPU path does not get faster per image with a larger batch (measured 0.23 img/s at batch 1 and at batch 2 in round 1).

    torch's default intra-op thread count on a 128-thread host oversubscribes this step (rounds 4-5 reported 0.16 - 0.28 images/s from it,
    below the survey container's 8-thread 0.27): the leg sweeps torch.set_num_threads over {8, 16, 32, 64, 128} (one warm-up + one timed
    step each), keeps the fastest and times it again — `value` is that repeat, `repeat_ratio` how well the two agree."""
    prev = torch.get_num_threads()
    logical, physical = os.cpu_count() or 1, _physical_cores()
    try:
        m, A, B = _cpu_model('unet', 9, 256)
        sweep = {}
        m.optimize_parameters(A, B)                          # (the first steps of a process are slow whatever the thread count: allocator, MKLDNN primitives)
        m.optimize_parameters(A, B)
        for nt in sorted({n for n in (8, 16, 32, 64, 128) if n <= logical} | {min(8, logical)}):
            torch.set_num_threads(nt)
            m.optimize_parameters(A, B)                      # warm-up (thread pool, allocator)
            sweep[nt] = _cpu_time(m, A, B, 1)
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        again = _cpu_time(m, A, B, 3)
        c2 = 1.0 / again
        m1, A1, B1 = _cpu_model('affine', 6, 128)
        m1.optimize_parameters(A1, B1)
        c1 = 1.0 / _cpu_time(m1, A1, B1, 6)
    finally:
        torch.set_num_threads(prev)
    return {"value": c2, "unit": "images/sec", "cores": best, "kind": "port",
            "sample": "3 steps of the config-2 shape (unet cfg A, resnet_9blocks, 256x256) at batch 1, no dropout, oracle/torch_ref.py on "
                      "torch CPU fp32 with the best intra-op thread count of a sweep; host has %d logical / %d physical cores" % (logical, physical),
            "thread_sweep_s_per_step": {str(k): round(v, 3) for k, v in sweep.items()},
            "repeat_ratio": round(sweep[best] / again, 3),
            "physical_cores": physical, "logical_cores": logical,
            "config1": {"value": c1, "unit": "images/sec", "cores": best,
                        "sample": "6 steps of BASELINE config 1 (affine STN, resnet_6blocks, 128x128, batch 1, no dropout)"}}


OTHER_CONFIGS = (
    # (name, batch per GPU, size, steps, warmup, extra flags): the per-GPU shards of the other BASELINE.json configs (parity cases with
    # full-width fixtures in tests/; here only their step time, eager and as a hipGraph replay)
    ("C1 affine, resnet_6blocks, 128x128, batch 1", 1, 128, 20, 5, ['--stn_type', 'affine', '--netG', 'resnet_6blocks']),
    ("C3 shard: C2 + multi-resolution D, 256x256, batch 8", 8, 256, 5, 2, ['--multi_resolution', '2']),
    ("C4 shard: 512x512, bilateral smoothness, multi-resolution regulariser, batch 4", 4, 512, 5, 2,
     ['--stn_bilateral_alpha', '1.5', '--stn_multires_reg', '2']),
    ("C5 shard: 1024x1024, deep stn cfg, batch 1", 1, 1024, 5, 2, ['--stn_cfg', 'deep']),
)


def _make(batch, size, extra, dev, seed=0):
    import contextlib
    import io
    from nemar_amd.models import create_model
    opt = build_opt(batch, size, extra)
    opt.gpu_ids = [dev.index]
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(opt)
        model.setup(opt)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    A = torch.rand(batch, 3, size, size, device=dev, generator=g) * 2 - 1
    B = torch.rand(batch, 3, size, size, device=dev, generator=g) * 2 - 1
    return model, {'A': A, 'B': B, 'A_paths': ['synthetic'], 'B_paths': ['synthetic']}


def _time_steps(model, data, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.set_input(data)
        model.optimize_parameters()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def other_configs(dev):
    """Step time of the other BASELINE shapes on this GPU (their per-GPU shard), eager launches and hipGraph replay."""
    import gc
    out = []
    for name, batch, size, steps, warmup, extra in OTHER_CONFIGS:
        row = {"config": name}
        try:
            model, data = _make(batch, size, extra, dev)
            for _ in range(warmup):
                model.set_input(data)
                model.optimize_parameters()
            row["eager_ms_per_step"] = _time_steps(model, data, steps)
            try:
                model.set_input(data)
                model.enable_step_graph()
                row["graph_ms_per_step"] = _time_steps(model, data, steps)
            except Exception as e:  # noqa: BLE001
                row["graph_error"] = '%s: %s' % (type(e).__name__, e)
            best = min(v for k, v in row.items() if k.endswith('_ms_per_step'))
            row["images_per_sec"] = batch / best * 1e3
        except Exception as e:  # noqa: BLE001  (a side measurement must not take the bench line down)
            row["error"] = '%s: %s' % (type(e).__name__, e)
        finally:
            from nemar_amd import ops
            ops.pin_workspaces(False)
            ops.step_params(False)
            model = data = None
            gc.collect()
            torch.cuda.empty_cache()
        out.append(row)
    return out


def grid_sample_hbm_stress(dev):
    """BASELINE's second metric at the HBM-stress shape of config 5 (8 x 3 x 1024 x 1024; the bench config's 256 x 256 launches are
    8-30 us: latency-sized): the four grid_sample launches of a step — forward of real_A and fake_B, backward with grad_input (fake_B) and
    without (real_A) — through the C ABI on a near-identity field (what the registration net emits early in training) and on a smooth
    3-pixel field, algorithmic bytes (32 / 52 / 40 B per pixel) over event-timed duration."""
    import ctypes
    from nemar_amd import _lib
    lib = _lib.load()
    N, C, H, W = 8, 3, 1024, 1024
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    g = torch.Generator(device=dev).manual_seed(5)
    img = torch.rand(N, C, H, W, device=dev, generator=g) * 2 - 1
    go = torch.randn(N, C, H, W, device=dev, generator=g)
    res, gin = torch.empty_like(img), torch.empty_like(img)
    wsb = lib.grid_sample_bwd_workspace(N, C, H, W)
    ws = torch.zeros(wsb // 4 + 16, device=dev)
    out = {}
    for name, off in (("near_identity", torch.randn(N, 2, H, W, device=dev, generator=g) * 1e-4),
                      ("smooth_3px", torch.nn.functional.interpolate(torch.randn(N, 2, H // 32, W // 32, device=dev, generator=g), size=(H, W),
                                                                     mode='bilinear', align_corners=False) * (6.0 / W))):
        gd = torch.empty_like(off)

        def t(fn, iters=10):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e-3
        fwd = t(lambda: lib.grid_sample_fwd(P(img), P(off), 1, P(res), N, C, H, W, H, W, st()))
        bgi = t(lambda: lib.grid_sample_bwd(P(img), P(off), 1, P(go), P(gin), 0, P(gd), 0, N, C, H, W, H, W, P(ws), wsb, st()))
        bng = t(lambda: lib.grid_sample_bwd(P(img), P(off), 1, P(go), None, 0, P(gd), 0, N, C, H, W, H, W, P(ws), wsb, st()))
        px = N * H * W
        tot_b, tot_t = px * (32 * 2 + 52 + 40), 2 * fwd + bgi + bng
        out[name] = {"fwd_us": fwd * 1e6, "fwd_GBps": px * 32 / fwd / 1e9, "bwd_gin_us": bgi * 1e6, "bwd_gin_GBps": px * 52 / bgi / 1e9,
                     "bwd_nogin_us": bng * 1e6, "bwd_nogin_GBps": px * 40 / bng / 1e9,
                     "step_mix_GBps": tot_b / tot_t / 1e9, "frac_of_hbm_peak": tot_b / tot_t / 1e9 / HBM_PEAK_GBS}
    out["shape"] = [N, C, H, W]
    out["peak"] = HBM_PEAK_GBS
    return out


def exact_route_ab(dev, batch, size, extra):
    """(measurement library) the training step with every 16-bit-pipe kernel off — all convolutions on the exact-fp32 MFMA / VALU kernels —
    beside the default routes: same model and batch, 5 eager steps each."""
    from nemar_amd import ops
    model, data = _make(batch, size, extra, dev)
    for _ in range(3):
        model.set_input(data)
        model.optimize_parameters()
    ms_default = _time_steps(model, data, 5)
    for k in (20, 24):
        ops.tune(k, 0)
    for _ in range(2):
        model.set_input(data)
        model.optimize_parameters()
    ms_exact = _time_steps(model, data, 5)
    return {"ms_per_step_eager": ms_exact, "images_per_sec": batch / ms_exact * 1e3, "default_route_ms_per_step_eager": ms_default,
            "speedup_from_fp16x3_routes": ms_exact / ms_default,
            "how": "child process on libnemar_hip_ab.so (the product library has no switch): same model and batch, nemar_tune 20=0 and 24=0 "
                   "(no fp16 x 3 kernel anywhere), 5 eager steps each"}


def ab_child(kind, a):
    """The side measurements that need a route switch run in a child process bound to the measurement build of the library
    (NEMAR_AB_LIBRARY=1, nemar_amd/_lib.py): this process — the timed one — is on the product library, which has none."""
    import subprocess
    env = dict(os.environ, NEMAR_AB_LIBRARY="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--ab-child", kind, "--batch", str(a.batch), "--size", str(a.size)]
    for o in a.opt:
        cmd.append("--opt=" + o)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("child %s failed (%d): %s" % (kind, r.returncode, (r.stderr or r.stdout)[-400:]))
    return json.loads(lines[-1])


def route_agreement(dev):
    """(measurement library) How far the fp16 x 3 routes move the step's GRADIENTS: one full-width config-2-shaped step (batch 1, no dropout, same seeded weights
    and inputs) on the default routes and with every 16-bit-pipe kernel off (nemar_tune 20=0, 24=0: exact-fp32 MFMA / VALU), compared
    per network as 1 - cos of the whole flat gradient and as relative L2 distance."""
    import gc
    from nemar_amd import ops
    grads = {}
    for tag, keys in (("default", ()), ("exact", (20, 24))):
        for k in keys:
            ops.tune(k, 0)
        try:
            model, data = _make(1, 256, ['--no_dropout'], dev, seed=7)
            model.set_input(data)
            model.optimize_parameters()
            torch.cuda.synchronize()
            grads[tag] = {n: getattr(model, 'optimizer_' + n).flat_g.double().clone() for n in ('T', 'R', 'D')}
        finally:
            for k in keys:
                ops.tune(k, 1)
            model = None
            gc.collect()
    out = {}
    for n in ('T', 'R', 'D'):
        a, b = grads["default"][n], grads["exact"][n]
        out[n] = {"one_minus_cos": float(1.0 - (a * b).sum() / (a.norm() * b.norm())), "rel_l2": float((a - b).norm() / b.norm())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='per-GPU batch (weak scaling)')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the side measurements of the default run (exact-route A/B, the other BASELINE shapes, route agreement)')
    ap.add_argument('--graph', choices=('auto', 'on', 'off'), default='auto',
                    help='replay the step as ONE captured hipGraph in the timed region (NEMARModel.enable_step_graph: every kernel of the '
                         'step, dropout offsets and Adam scalars in device memory; bit-identical to eager launches, tests/test_step_gpu.py). '
                         'auto = on a single GPU, eager if the capture fails; the per-kernel roofline pass after the timed region is eager '
                         'either way (event records cannot sit inside a graph)')
    ap.add_argument('--opt', action='append', default=[], metavar='FLAG',
                    help='extra reference-style option for other BASELINE configs, e.g. --opt=--multi_resolution --opt=2')
    ap.add_argument('--ab-child', choices=('exact_route', 'route_agreement'), default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.ab_child:
        # side measurement on the measurement library (ab_child): one JSON line, nothing else
        torch.cuda.set_device(0)
        from nemar_amd import _lib
        assert _lib.load().has_switches, _lib.load().path
        dev0 = torch.device('cuda', 0)
        print(json.dumps(exact_route_ab(dev0, a.batch, a.size, a.opt) if a.ab_child == 'exact_route' else route_agreement(dev0)))
        return

    from nemar_amd import launch
    if a.gpus > 1 and not launch.under_launcher():
        # started directly (`python bench.py --gpus N`): become N ranks, one per GPU, over RCCL — never a silent 1-rank run
        raise SystemExit(launch.spawn_local_ranks(a.gpus))
    from nemar_amd import distributed as dist
    rank, world, local = dist.init_from_env()
    if world != a.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    if torch.cuda.device_count() <= local:
        raise SystemExit('rank %d: local rank %d has no GPU (%d visible)' % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    from nemar_amd import ops
    from nemar_amd.models import create_model
    import contextlib
    import io
    cpu_base = None
    if world == 1 and not a.no_cpu_baseline:
        # the CPU leg runs FIRST: the GPU phase is then one contiguous block at the end of the run (the driver samples GPU activity)
        with contextlib.redirect_stdout(sys.stderr):     # the net constructors print; stdout carries the JSON line only
            cpu_base = cpu_baseline()
    opt = build_opt(a.batch, a.size, a.opt)
    opt.gpu_ids = [local]
    torch.manual_seed(0)      # identical initial weights on every rank (also broadcast at setup); NEMARModel seeds the
                              # dropout stream with torch.initial_seed() + rank
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(opt)
        model.setup(opt)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    A = torch.rand(a.batch, 3, a.size, a.size, device=dev, generator=g) * 2 - 1
    B = torch.rand(a.batch, 3, a.size, a.size, device=dev, generator=g) * 2 - 1
    data = {'A': A, 'B': B, 'A_paths': ['synthetic'], 'B_paths': ['synthetic']}

    timer = KernelTimer()
    ops.set_kernel_timer(timer)

    def step():
        model.set_input(data)
        model.optimize_parameters()

    for _ in range(a.warmup):
        step()
    graph_on, graph_note = False, None
    if a.graph != 'off' and world == 1 and not dist.is_distributed():
        try:
            model.set_input(data)
            model.enable_step_graph()
            graph_on = True
        except Exception as e:  # noqa: BLE001  (bench only: fall back to eager launches and say so in the JSON line)
            if a.graph == 'on':
                raise
            graph_note = '%s: %s' % (type(e).__name__, e)
            model._graph = None
    elif a.graph == 'on':
        raise SystemExit('--graph on: single GPU only')

    # --graph auto: a hipGraph replay runs the step's kernels as ONE chain (the ROCm graph executor does not overlap the branches a
    # two-stream capture describes), eager launches put the weight-gradient branch on a side stream (ops._on_side) but pay ~900 kernel
    # launches from Python.  Which wins depends on the configuration: time a few steps of each and keep the faster for the timed region.
    launch_probe = None
    if graph_on and a.graph == 'auto':
        def probe(n=4):
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3
        t_graph = probe()
        g_saved, model._graph = model._graph, None           # (eager launches with the step parameters still in device memory)
        t_eager = probe()
        launch_probe = {"graph_ms_per_step": t_graph, "eager_ms_per_step": t_eager}
        if t_eager < 0.99 * t_graph:
            graph_on = False
            graph_note = 'eager launches were faster than the graph replay in the pre-run probe (%.2f vs %.2f ms/step)' % (t_eager, t_graph)
            g_saved = None
            ops.pin_workspaces(False)
        else:
            model._graph = g_saved

    multi = dist.is_distributed()        # world > 1 (or the one-rank RCCL smoke configuration, NEMAR_DIST_SINGLE=1)

    def barrier():
        if multi:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    peak_mem = torch.cuda.max_memory_allocated(dev)
    # per-launch durations of the roofline kernels: HIP events on the launch stream in a SEPARATE pass of two steps, after
    # the timed region (the event records would otherwise sit inside it)
    model._graph = None                  # the roofline pass needs eager launches (step parameters stay in device memory)
    side_prev = ops.side_stream(False)   # ... on ONE stream: beside a side-stream kernel a launch shares the CUs and its duration says nothing about it
    timer.enabled = True
    from nemar_amd import _lib
    lib = _lib.load()
    lib.kernel_timer(1)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    timer.enabled = False
    # peak memory of the single-stream order (the side-stream branch keeps the tensors it reads until the side stream has run it)
    torch.cuda.reset_peak_memory_stats(dev)
    step()
    torch.cuda.synchronize()
    peak_mem_single = torch.cuda.max_memory_allocated(dev)
    ops.side_stream(side_prev)
    import ctypes
    tk_ms_c, tk_fl_c, tk_n_c = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_int(0)
    lib.kernel_timer_read(ctypes.byref(tk_ms_c), ctypes.byref(tk_fl_c), ctypes.byref(tk_n_c))
    lib.kernel_timer(0)
    tk_ms, tk_flop, tk_n = tk_ms_c.value, tk_fl_c.value, tk_n_c.value
    # ... and the same launches as the TIMED step runs them: beside the side stream's weight-gradient kernels (the data gradient's launches
    # share the matrix pipe with them; the forward launches have the chip to themselves)
    tk2 = None
    if side_prev and tk_n:
        lib.kernel_timer(1)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        a_ms, a_fl, a_n = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_int(0)
        lib.kernel_timer_read(ctypes.byref(a_ms), ctypes.byref(a_fl), ctypes.byref(a_n))
        lib.kernel_timer(0)
        if a_n.value:
            tk2 = (a_ms.value, a_fl.value, a_n.value)
    rank_ms = [dt / a.steps * 1e3]
    buckets = sum(len(getattr(model, n).launched) for n in ("sync_T", "sync_D", "sync_R"))
    if multi:
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = dt
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)         # every rank's own wall time
        rank_ms = [float(x) / a.steps * 1e3 for x in t.tolist()]
        dt = float(t.max().item())
    losses = model.get_current_losses()
    bucket_rows = []
    if multi:
        # one more step with events around every bucket's all-reduce (compute stream at launch, side stream at completion): achieved
        # bus bandwidth per bucket and how long finish() had to wait — the part of the exchange that is NOT hidden behind backward
        evs = []
        for nm in ("D", "R", "T"):
            s_ = getattr(model, "sync_" + nm)

            def launch(b, s_=s_, nm=nm, orig=s_._launch):
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
                orig(b)
                e1 = torch.cuda.Event(enable_timing=True)
                if s_._side is not None:
                    with torch.cuda.stream(s_._side):
                        e1.record()
                else:
                    e1.record()
                lo, hi = s_.buckets[b]
                evs.append((nm, b, (hi - lo) * 4, e0, e1))
            s_._launch = launch
        step()
        torch.cuda.synchronize()
        for nm, b, nbytes, e0, e1 in evs:
            ms = e0.elapsed_time(e1)
            bucket_rows.append({"optimizer": nm, "bucket": b, "MB": nbytes / 1e6, "ms_ready_to_reduced": ms,
                                "ring_GBps_per_rank": nbytes * 2.0 * (world - 1) / world / max(ms, 1e-6) / 1e6})
    stn_cfg = opt.stn_cfg
    extras = {}
    if world == 1 and not multi and not a.no_extras:
        # side measurements of the default single-GPU run — after the timed region, never inside it
        import gc
        ops.pin_workspaces(False)
        ops.step_params(False)
        ops.set_kernel_timer(None)
        model = None
        gc.collect()
        torch.cuda.empty_cache()
        for kind in ("exact_route",) + (("route_agreement",) if a.size == 256 and not a.opt else ()):
            try:
                extras[kind] = ab_child(kind, a)
            except Exception as e:  # noqa: BLE001
                extras[kind] = {"error": '%s: %s' % (type(e).__name__, e)}
        if a.size == 256 and not a.opt:
            extras["other_configs"] = other_configs(dev)
            try:
                extras["roofline_grid_sample_1024"] = grid_sample_hbm_stress(dev)
            except Exception as e:  # noqa: BLE001
                extras["roofline_grid_sample_1024"] = {"error": '%s: %s' % (type(e).__name__, e)}
    if multi:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return

    spans = timer.summary()
    out = {
        "metric": "train images/sec (256x256 A/B pairs)", "value": a.batch * world * a.steps / dt,
        "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32",
        "dtype_note": "fp32 tensors and fp32 accumulation throughout; convolutions with >= 5 output channels above 30 M multiply-adds form "
                      "each fp32 product from three fp16 partial products on the 16-bit MFMA (per-sample / per-tile power-of-two scales; "
                      "measured error vs float64 at or below the exact-fp32 MFMA kernels', DESIGN.md 4c/4d); every other kernel is exact fp32",
        "data": "synthetic U[-1,1) A/B pairs resident in HBM; reference-equivalent random init",
        "config": {"workload": "BASELINE configs[1]: --stn_type unet --stn_cfg A, resnet_9blocks T, basic PatchGAN D, "
                               "%dx%d, batch %d per GPU, dropout on, lambda_smooth 10, fp32%s"
                               % (a.size, a.size, a.batch, (" + " + " ".join(a.opt)) if a.opt else ""),
                   "global_batch": a.batch * world, "parallelism": "dp%d" % world, "stn_cfg": stn_cfg,
                   "step": "NEMARModel.optimize_parameters(): fwd + D update + T/R update + 3x Adam"
                           + (" — the timed steps are replays of ONE captured hipGraph of the step" if graph_on else "")},
        "launch": "hipGraph replay" if graph_on else ("eager, weight-gradient branch on a side stream" + (" (%s)" % graph_note if graph_note else "")),
        "launch_probe": launch_probe,
        "losses_finite": all(v == v and abs(v) != float('inf') for v in losses.values()),
        "peak_memory_GB": {"timed_region": peak_mem / 1e9, "single_stream_order": peak_mem_single / 1e9,
                           "note": "torch.cuda.max_memory_allocated; the weight-gradient branch on the side stream keeps the tensors it reads "
                                   "alive until the side stream has finished that branch (released per branch, nemar_amd/ops.py _on_side)"},
        "rank_ms_per_step": rank_ms,                      # one entry per rank: the N > 1 run cannot degrade to one rank unnoticed
        "dist": {"backend": "nccl (RCCL)" if multi else None, "buckets_launched_last_step": buckets, "buckets": bucket_rows or None,
                 "launcher": "self-spawned ranks" if os.environ.get("NEMAR_SPAWNED") else ("torchrun" if multi else None)},
    }
    if os.environ.get("NEMAR_BENCH_DUMP_LOSSES"):       # tests: the losses of the last step + how many gradient buckets went out
        out["losses"] = {k: float(v) for k, v in losses.items()}
        out["dist_buckets_launched"] = buckets
    C = 256
    hw = (a.size // 4) ** 2
    # HBM traffic comes from separate rocprofv3 --pmc passes over the same kernels (tools/profile_evidence.sh); it is
    # only attached when the bench runs the shape those passes measured.
    pmc, std = {}, (a.size == 256 and a.batch == 8)
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_FILE)) as f:
            pmc = json.load(f)
    except OSError:
        pass
    # Dominant kernel: igemm_split16_kernel<2,2,3> — forward AND data gradient of the 256-channel 3x3 resblock convolutions (72 launches
    # per step; the discriminator's 4x4 layer runs a different instantiation and is not timed).  Its launches are timed by HIP events recorded on the launch stream
    # inside the library (nemar_kernel_timer), which also adds up their algorithmic flop (2 N OH OW K C R S, fp32-equivalent).  Every
    # fp32 product is executed as three fp16 partial products, so the matrix pipe does 3x the algorithmic flop: the roofline of
    # this formulation is the dense fp16 MFMA peak / 3.
    if tk_n:
        sec = tk_ms * 1e-3 / tk_n
        flop = tk_flop / tk_n
        peak = F16_MFMA_PEAK_TF / 3.0
        out["roofline"] = {"bound": "mfma", "achieved": flop / sec / 1e12, "peak": peak, "unit": "TFLOP/s",
                           "frac": flop / sec / 1e12 / peak,
                           # PMC passes ran the step's own batch-16 launch (77.3 GFLOP: T's two applications as one batch);
                           # scaled by flop per launch should the timed launches differ
                           "traffic": (pmc["igemm_split16"]["traffic_bytes"] * flop / 77309411328.0) if std and pmc.get("igemm_split16") else None,
                           "algorithmic_bytes": (pmc["igemm_split16"]["algorithmic_bytes"] * flop / 77309411328.0) if std and pmc.get("igemm_split16") else None,
                           "traffic_source": (PMC_FILE + " (batch-16 launch)") if std and pmc.get("igemm_split16") else None,
                           "kernel": "igemm_split16_kernel<2,2,3> (conv2d_fwd / conv2d_bwd_data of the 256->256 3x3 reflect layers @%dx%d, %d images per "
                                     "launch = T's two applications of the batch of %d as one; fp32 operands as fp16 x 3 partial products, fp32 "
                                     "accumulate)" % (a.size // 4, a.size // 4, 2 * a.batch, a.batch),
                           "launches_timed": tk_n, "avg_launch_us": sec * 1e6,
                           "algorithmic_flop_per_launch": flop,
                           "peak_basis": "2500 TFLOP/s dense fp16 MFMA / 3 products per fp32 product",
                           "issued_mfma_TFLOPs": 3.0 * flop / sec / 1e12, "issued_frac_of_fp16_peak": 3.0 * flop / sec / 1e12 / F16_MFMA_PEAK_TF,
                           "vs_fp32_mfma_peak_157": flop / sec / 1e12 / FP32_MFMA_PEAK_TF,
                           # the nominal peak assumes 2.4 GHz; a pure MFMA loop on random operands sustains 1730 TFLOP/s on this chip
                           # (power-limited clock; 2450-2530 on all-zero / all-one operands)
                           "sustained_mfma_TFLOPs_measured": F16_MFMA_SUSTAINED_TF,
                           "frac_of_sustained_mfma": 3.0 * flop / sec / 1e12 / F16_MFMA_SUSTAINED_TF,
                           "sustained_source": "profiles/r4_mfma_peak_modes.txt (f16, random operands, 2 waves/SIMD, long run)"}
        if tk2 is not None:
            sec2 = tk2[0] * 1e-3 / tk2[2]
            out["roofline"]["in_two_stream_step"] = {
                "avg_launch_us": sec2 * 1e6, "achieved": tk2[1] / tk2[2] / sec2 / 1e12, "frac": tk2[1] / tk2[2] / sec2 / 1e12 / peak,
                "launches_timed": tk2[2],
                "note": "the same launches with the weight-gradient branch on the side stream, as the timed step runs them: the data "
                        "gradient's launches share the CUs with the side stream's weight-gradient kernels (forward launches do not), so "
                        "this is a share of the chip, not the kernel's efficiency — `frac` above is that"}
    # Operator level (what the step pays per residual-block layer): the whole C-ABI call — support passes (split / pack / slab sums) + main
    # kernel — event-timed on ONE stream in the same two-step pass; the batch of a call = T's two applications (2 x batch)
    rb = {}
    for tag, nm in (('igemm_fwd_resblock', 'forward'), ('dgrad_resblock', 'data_gradient'), ('wgrad_resblock', 'weight_gradient')):
        if tag in spans:
            n, sec = spans[tag]
            fl = 2.0 * (2 * a.batch) * C * hw * C * 9
            rb[nm] = {"avg_call_us": sec * 1e6, "calls_timed": n, "fp32_equivalent_TFLOPs": fl / sec / 1e12,
                      "frac": fl / sec / 1e12 / (F16_MFMA_PEAK_TF / 3.0)}
    if rb:
        tot = sum(v["avg_call_us"] for v in rb.values())
        rb["all_three"] = {"avg_us_per_layer": tot, "frac": sum(2.0 * (2 * a.batch) * C * hw * C * 9 for _ in range(len(rb))) / (tot * 1e-6) / 1e12 / (F16_MFMA_PEAK_TF / 3.0)}
        rb["peak_basis"] = "2500 TFLOP/s dense fp16 MFMA / 3 products per fp32 product; 256->256 3x3 reflect layer, %d images per call" % (2 * a.batch)
        # Round 6: inside the one-node ResnetBlock the three calls run on producer-written operand planes (no split / max passes inside them
        # any more): what used to be their support passes now sits in the InstanceNorm passes on either side.  The honest per-layer price
        # therefore includes those two producers (one InstanceNorm forward + one InstanceNorm backward per convolution layer).
        prod = {nm: spans[tag][1] * 1e6 for tag, nm in (('in_fwd_planes_resblock', 'instnorm_fwd_planes'), ('in_bwd_planes_resblock', 'instnorm_bwd_planes')) if tag in spans}
        if len(prod) == 2 and len(rb) >= 5:
            tot2 = tot + sum(prod.values())
            rb["with_producers"] = {"avg_call_us": prod, "avg_us_per_layer": tot2,
                                    "frac": 3 * 2.0 * (2 * a.batch) * C * hw * C * 9 / (tot2 * 1e-6) / 1e12 / (F16_MFMA_PEAK_TF / 3.0),
                                    "note": "three convolution calls + the InstanceNorm forward and backward passes that write their operand planes"}
        out["roofline_operator"] = rb
    px = a.batch * a.size * a.size
    gs = {}
    for tag, bpp in (('grid_sample_fwd', 4 * (2 * 3 + 2)), ('grid_sample_bwd_gin', 4 * (3 * 3 + 4)),
                     ('grid_sample_bwd_nogin', 4 * (2 * 3 + 4))):
        if tag in spans:
            n, sec = spans[tag]
            gs[tag] = {"avg_launch_us": sec * 1e6, "GBps": px * bpp / sec / 1e9, "bytes_per_px": bpp, "launches_timed": n}
    if gs:
        tot_b = sum(v["bytes_per_px"] * px for v in gs.values())
        tot_t = sum(v["avg_launch_us"] * 1e-6 for v in gs.values())
        out["roofline_grid_sample"] = {"bound": "hbm", "achieved": tot_b / tot_t / 1e9, "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": tot_b / tot_t / 1e9 / HBM_PEAK_GBS,
                                       "traffic": sum(pmc.get(t, {}).get("traffic_bytes", 0) for t in gs)
                                       if std and pmc else None,
                                       "kernels": gs}
    out.update(extras)
    if cpu_base is not None:
        out["cpu_baseline"] = cpu_base
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
