"""The N>1 data-parallel path on CPU: two gloo processes (127.0.0.1) drive nemar_amd.distributed exactly as the GPU
ranks do over RCCL, plus the sharding identity the design rests on: with equal shards, the mean over
ranks of per-shard gradients equals the full-batch gradient, because every operator is per-sample and every loss is a
batch mean."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import seeded
from step_configs import STEP_CONFIGS


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeOpt:
    """The three attributes distributed.py touches on ops.FlatAdam."""

    def __init__(self, n, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.flat_g = torch.randn(n, generator=g)
        self.flat_p = torch.randn(n, generator=g)
        self.m = torch.randn(n, generator=g)
        self.v = torch.rand(n, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from nemar_amd import distributed as dist
    rk, ws, _ = dist.init_from_env(backend="gloo")
    assert (rk, ws) == (rank, world) and dist.is_distributed() and dist.rank() == rank
    opts = [_FakeOpt(1000, rank), _FakeOpt(37, rank)]
    want = [sum(_FakeOpt(n, r).flat_g for r in range(world)) / world for n in (1000, 37)]
    dist.all_reduce_gradients(opts)
    for o, w in zip(opts, want):
        assert torch.allclose(o.flat_g, w, atol=1e-6)
    dist.broadcast_parameters(opts, src=0)
    ref = _FakeOpt(1000, 0)
    assert torch.equal(opts[0].flat_p, ref.flat_p) and torch.equal(opts[0].m, ref.m) and torch.equal(opts[0].v, ref.v)
    assert dist.shard_range(8) == (rank * 4, rank * 4 + 4)
    with pytest.raises(ValueError):
        dist.shard_range(7)
    out.put(rank)
    torch.distributed.destroy_process_group()


def test_gloo_world_size_2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(out.get(timeout=5) for _ in range(2)) == [0, 1]


def _sync_worker(rank, world, port, out):
    """Real ops.FlatAdam buffers (CPU tensors: construction, views and zero_grad are plain torch) driven through GradSync the way
    the conv backward does: gradients written into the flat buffer's views, ops.grad_ready() per contribution."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from nemar_amd import distributed as dist
    from nemar_amd import ops
    dist.init_from_env(backend="gloo")
    torch.manual_seed(0)                                               # same initial parameters on both ranks
    shapes = [(64, 3, 7, 7), (64,), (128, 64, 3, 3), (128,), (3, 64, 7, 7), (3,), (5,)]
    params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    opt = ops.FlatAdam(params)
    dist.broadcast_parameters([opt])
    sync = dist.GradSync(opt, bucket_bytes=16 << 10)
    assert len(sync.buckets) >= 3 and sync.buckets[0][1] == opt.flat_numel and sync.buckets[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(sync.buckets, sync.buckets[1:]))          # contiguous, walked from the end
    g = torch.Generator().manual_seed(1000 + rank)
    contrib = [[torch.randn(s, generator=g) for s in shapes] for _ in range(2)]      # two passes (T is applied twice)
    opt.zero_grad()
    sync.begin(expected=2)
    launched_at = {}
    for pas in range(2):
        for i in range(len(params) - 1, -1, -1):                       # backward order: last layer first
            params[i].grad.add_(contrib[pas][i])                       # the kernels accumulate into the flat buffer's views
            ops.grad_ready(params[i])
            for b in sync.launched:
                launched_at.setdefault(b, (pas, i))
    # buckets go out in order, the first one as soon as ITS parameters are final — long before the backward pass ends
    assert sync.launched == sorted(sync.launched) and launched_at[0][0] == 1 and launched_at[0][1] > 0
    assert len(sync.launched) == len(sync.buckets)
    sync.finish()
    local = [contrib[0][i] + contrib[1][i] for i in range(len(shapes))]
    gathered = [None] * world
    td = torch.distributed
    td.all_gather_object(gathered, [t.numpy() for t in local])
    for i, p in enumerate(params):
        mean = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
        assert torch.allclose(p.grad, mean, atol=1e-6), i
    with pytest.raises(RuntimeError):                                  # a third contribution after the bucket went out
        sync.begin(expected=1)
        ops.grad_ready(params[-1])
        ops.grad_ready(params[-1])
    out.put(rank)
    td.destroy_process_group()


def test_bucketed_overlapped_gradient_sync_on_flat_adam_buffers():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(out.get(timeout=5) for _ in range(2)) == [0, 1]


def test_single_process_is_a_noop():
    from nemar_amd import distributed as dist
    o = _FakeOpt(10, 0)
    g = o.flat_g.clone()
    dist.all_reduce_gradients([o])
    assert torch.equal(o.flat_g, g) and dist.world_size() == 1 and dist.shard_range(6) == (0, 6)


def test_mean_of_shard_gradients_equals_full_batch_gradient():
    """Oracle-level check of the data-parallel identity on the affine128 step (batch 2 -> two shards of 1)."""
    from test_oracle_golden import build_ref_model
    cfg = STEP_CONFIGS['affine128']
    A, B = seeded.seeded_images(2, 3, cfg['size'], cfg['size'], cfg['seed'])
    A, B = torch.from_numpy(A), torch.from_numpy(B)
    full = build_ref_model('affine128', dtype=torch.float64)
    full.optimize_parameters(A, B)
    shards = []
    for r in range(2):
        m = build_ref_model('affine128', dtype=torch.float64)
        m.optimize_parameters(A[r:r + 1], B[r:r + 1])
        shards.append(m)
    # D's gradient is taken before any update, so it is exactly comparable; T/R gradients are taken against the
    # already-updated D, which differs per shard, so compare D here (the all-reduce happens BEFORE each update in the
    # real step, which is what makes the replicas stay identical)
    for k, g in full.grads_D.items():
        mean = (shards[0].grads_D[k] + shards[1].grads_D[k]) / 2
        np.testing.assert_allclose(mean.numpy(), g.numpy(), rtol=1e-9, atol=1e-12)
