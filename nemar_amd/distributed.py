"""Data parallelism for the NeMAR step: one process per GPU, batch sharded by rank, gradient averaging over RCCL.

The reference's only multi-GPU mechanism is single-process nn.DataParallel (models/networks.py:108-111,
models/stn/__init__.py:30-35).  Every operator on the path is per-sample (InstanceNorm has no cross-sample
statistics) and every loss is a batch mean, so with equal shards grad_full = mean_r(grad_r).
The MI355X design is therefore: replicas of T/R/D and their Adam state on every rank, and an all-reduce(avg) of every
optimizer's flat gradient buffer (ops.FlatAdam.flat_g) per step, OVERLAPPED with the backward kernels:
  * each flat buffer is cut into contiguous buckets (~8 MB: enough to run xGMI's 7 x ~153 GB/s links at bandwidth, small
    enough that the last one is short), ordered so that bucket 0 holds the LAST layers' parameters — the ones whose
    gradients are final first;
  * the weight-gradient launches report every finished parameter (ops.grad_ready); when the last expected contribution of a
    bucket has been launched, an event is recorded on the compute stream and the bucket's all-reduce is issued on a side
    stream behind that event — RCCL moves it over xGMI while the remaining data/weight-gradient kernels run;
  * `finish()` makes the compute stream wait for the side stream right before the optimizer step (no host sync).
The discriminator's buckets are on the critical path by data dependence (D.step() needs them, and the first kernels of the
T/R phase evaluate the UPDATED D), so only the part of them that becomes ready before backward_D ends is hidden; the T
and R buckets (54 of the 65 MB) overlap with the rest of backward_T_and_R.

`backend="nccl"` is RCCL on ROCm; the CPU test tier exercises the same code over gloo with world_size 2.
"""
import os

import torch
import torch.distributed as td


def _forced():
    # NEMAR_DIST_SINGLE=1: run the whole data-parallel machinery (RCCL communicator, bucketed side-stream collectives, parameter
    # broadcast) with a world of ONE rank — lets a one-GPU box exercise every torch.distributed / RCCL call of the N > 1 path
    return os.environ.get("NEMAR_DIST_SINGLE", "0") == "1"


def is_distributed():
    return td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or _forced())


def rank():
    return td.get_rank() if (td.is_available() and td.is_initialized()) else 0


def world_size():
    return td.get_world_size() if (td.is_available() and td.is_initialized()) else 1


def init_from_env(backend=None, device=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun's
    contract).  Returns (rank, world_size, local_rank).  No-op for a single process.  `device`: the GPU index this rank uses (default
    LOCAL_RANK) — selected BEFORE the communicator is created, so that no stray context is opened on another GPU."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if (ws > 1 or _forced()) and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(lr if device is None else int(device))
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


BUCKET_BYTES = 8 << 20


class GradSync:
    """Bucketed, overlapped gradient averaging for ONE FlatAdam optimizer (see the module docstring).

    begin(expected) arms it for a backward pass in which every parameter receives `expected` gradient contributions
    (the translation net is applied twice per step, the discriminator three times in its own phase); ops.grad_ready(p)
    counts them; finish() issues whatever is left and orders the compute stream behind the collectives."""

    def __init__(self, opt, bucket_bytes=None):
        self.opt = opt
        bucket_bytes = BUCKET_BYTES if bucket_bytes is None else bucket_bytes
        # contiguous parameter ranges of the flat buffer, walked from the END (the last layers finish first)
        self.buckets = []                 # (lo, hi) element ranges of flat_g, in launch order
        self.bucket_of = {}               # id(param) -> bucket index
        ends = [o + (p.numel() + 3) // 4 * 4 for p, o in zip(opt.params, opt.offsets)]
        hi, members = opt.flat_numel, []
        for i in range(len(opt.params) - 1, -1, -1):
            members.append(i)
            lo = opt.offsets[i]
            if (hi - lo) * 4 >= bucket_bytes or i == 0:
                b = len(self.buckets)
                self.buckets.append((lo, hi))
                for j in members:
                    self.bucket_of[id(opt.params[j])] = b
                hi, members = lo, []
        assert ends[-1] == opt.flat_numel
        self.n_params = [0] * len(self.buckets)
        for p in opt.params:
            self.n_params[self.bucket_of[id(p)]] += 1
        self.active = False
        self.counting = False
        self.uses = {}
        self.works = []
        self.launched = []
        self._side = None
        self._events = {}
        for p in opt.params:
            p._grad_sync = self

    def count_uses(self):
        """Start counting how often each parameter is applied in the forward passes that follow (ops.conv2d & co. call
        note_use): begin() without an argument then expects exactly that many gradient contributions per parameter — no
        constants that mirror the model's call pattern."""
        self.uses = {}
        self.counting = is_distributed()

    def note_use(self, param):
        if self.counting:
            k = id(param)
            self.uses[k] = self.uses.get(k, 0) + 1

    def begin(self, expected=None):
        # (counting stays on: the discriminator's forward passes run inside backward_D(), after begin(); every forward use
        # precedes the first gradient contribution of its pass)
        if not is_distributed():
            return
        self.active = True
        self.expected = None if expected is None else int(expected)
        self.seen = {}
        self.remaining = list(self.n_params)
        self.works = []
        self.launched = []

    def ready(self, param):
        """One gradient contribution of `param` has been launched on the current stream."""
        if not self.active:
            return
        k = id(param)
        c = self.seen.get(k, 0) + 1
        self.seen[k] = c
        expected = self.uses.get(k, 0) if self.expected is None else self.expected
        if c > expected:
            raise RuntimeError("GradSync: parameter received %d gradient contributions, %d were announced — its bucket has "
                               "already been all-reduced" % (c, expected))
        if c == expected:
            b = self.bucket_of[k]
            self.remaining[b] -= 1
            if self.remaining[b] == 0:
                self._launch(b)

    def _launch(self, b):
        lo, hi = self.buckets[b]
        buf = self.opt.flat_g[lo:hi]
        self.launched.append(b)
        if buf.is_cuda:
            # a bucket may hold gradients written from BOTH lanes of the backward pass (the weight-gradient branch runs on a side
            # stream, ops._on_side): whatever moves the bucket — RCCL on its own stream or the staged test path — starts behind both
            from . import ops
            ops.order_current_after_both(buf.device)
        if buf.is_cuda and td.get_backend() != 'nccl':
            # test configuration only (two gloo ranks sharing one GPU): staged through the host, synchronously
            host = buf.detach().cpu()
            td.all_reduce(host, op=td.ReduceOp.SUM)
            buf.copy_(host / td.get_world_size())
        elif buf.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=buf.device)
            if b not in self._events:                     # one event per bucket for the life of the optimizer, re-recorded every step
                self._events[b] = torch.cuda.Event()
            ev = self._events[b]
            from . import ops
            ops.order_current_after_both(buf.device)      # weight gradients are issued on a side stream (ops._on_side)
            ev.record()                                   # everything that wrote this bucket is ahead of this point
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                self.works.append(td.all_reduce(buf, op=td.ReduceOp.AVG, async_op=True))     # RCCL averages in-kernel
        else:
            w = td.all_reduce(buf, op=td.ReduceOp.SUM, async_op=True)                        # gloo has no AVG
            self.works.append((w, buf))

    def finish(self):
        """Issue the buckets that never filled up (parameters without a gradient this pass), then make the compute stream
        wait for every collective of this pass."""
        self.counting = False
        if not self.active:
            return
        for b in range(len(self.buckets)):
            if b not in self.launched:
                self._launch(b)
        ws = td.get_world_size()
        for w in self.works:
            if isinstance(w, tuple):
                w[0].wait()
                w[1].div_(ws)
            else:
                w.wait()                                   # stream-level wait for NCCL work: no host sync
        self.active = False


def grad_sync_for(optimizers):
    return [GradSync(o) for o in optimizers]


def all_reduce_gradients(optimizers):
    """Average each optimizer's flat gradient buffer over ranks (in place), without overlap: the simple form, used when no
    GradSync is armed.  No-op for a single process."""
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
            if buf.is_cuda and td.get_backend() != 'nccl':       # gloo test configuration: through the host
                host = buf.detach().cpu()
                td.broadcast(host, src=src)
                buf.copy_(host)
            else:
                td.broadcast(buf, src=src)
