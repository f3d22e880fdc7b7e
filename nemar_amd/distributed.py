"""Data parallelism for the NeMAR step: one process per GPU, batch sharded by rank, gradient averaging over RCCL.

The reference's only multi-GPU mechanism is single-process nn.DataParallel (models/networks.py:108-111,
models/stn/__init__.py:30-35).  Every operator on the path is per-sample (InstanceNorm has no cross-sample
statistics) and every loss is a batch mean, so with equal shards grad_full = mean_r(grad_r).
The MI355X design is therefore: replicas of T/R/D and their Adam state on every rank, and ONE all-reduce(avg) per
optimizer per step over the optimizer's flat gradient buffer (ops.FlatAdam.flat_g) — the D bucket after
backward_D, the R and T buckets after backward_T_and_R.  Buckets are 8-45 MB: over xGMI (7 links x ~153 GB/s) that
is well under a millisecond per step against tens of milliseconds of MFMA work, so the collectives are issued
back-to-back (async) and waited for together rather than interleaved with individual layers.

`backend="nccl"` is RCCL on ROCm; the CPU test tier exercises the same code over gloo with world_size 2.
"""
import os

import torch
import torch.distributed as td


def is_distributed():
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def rank():
    return td.get_rank() if (td.is_available() and td.is_initialized()) else 0


def world_size():
    return td.get_world_size() if (td.is_available() and td.is_initialized()) else 1


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun's
    contract).  Returns (rank, world_size, local_rank).  No-op for a single process."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(lr)
        td.init_process_group(backend=backend, rank=rk, world_size=ws)
    return rk, ws, lr


def shard_range(global_batch, rk=None, ws=None):
    """[begin, end) of this rank's samples; the global batch must divide evenly (equal shards keep mean-of-means
    equal to the full-batch mean)."""
    rk = rank() if rk is None else rk
    ws = world_size() if ws is None else ws
    if global_batch % ws != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, ws))
    per = global_batch // ws
    return rk * per, (rk + 1) * per


def _flat_grads(optimizers):
    return [o.flat_g for o in optimizers]


def all_reduce_gradients(optimizers):
    """Average each optimizer's flat gradient buffer over ranks (in place).  No-op for a single process."""
    if not is_distributed():
        return
    ws = td.get_world_size()
    bufs = _flat_grads(optimizers)
    on_gpu = all(b.is_cuda for b in bufs)
    if on_gpu:
        works = [td.all_reduce(b, op=td.ReduceOp.AVG, async_op=True) for b in bufs]     # RCCL averages in-kernel
        for w in works:
            w.wait()
    else:
        works = [td.all_reduce(b, op=td.ReduceOp.SUM, async_op=True) for b in bufs]     # gloo has no AVG
        for w in works:
            w.wait()
        for b in bufs:
            b.div_(ws)


def broadcast_parameters(optimizers, src=0):
    """Make every rank start from rank `src`'s parameters (and Adam moments)."""
    if not is_distributed():
        return
    for o in optimizers:
        for buf in (o.flat_p, o.m, o.v):
            td.broadcast(buf, src=src)
