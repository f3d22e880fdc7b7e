"""Autograd glue: one torch.autograd.Function per fused operator, each a thin call into the gfx950 C-ABI library
(include/nemar_hip.h) on torch's current HIP stream.  PyTorch is used here for device memory, stream ordering and
the autograd graph only — every number is produced by the hand-written kernels.  There is no fallback path: the
library is loaded at import and a missing/failed build raises.

Conventions
  * tensors are made contiguous NCHW fp32 before their pointer is taken;
  * weight/bias gradients are ACCUMULATED by the kernels straight into `param.grad` (a view of the owning
    optimizer's flat gradient buffer, see FlatAdam) and the Function returns None for them — no per-parameter
    autograd accumulation kernels, and the flat buffer is what a data-parallel all-reduce consumes;
  * loss Functions return 0-dim tensors already multiplied by their lambda.
"""
import ctypes
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

L = _lib.load()          # raises NemarHipError when the extension is missing: no CPU / eager fallback



class _SizeQueries:
    """The library's workspace-size queries are pure functions of their integer arguments (and, in the measurement build, of the switch
    state: ops.tune() forgets the memo) — ~700 ctypes round trips per step otherwise."""

    def __init__(self, lib):
        self._lib, self._memo = lib, {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        memo = self._memo

        def q(*a):
            key = (name, a)
            v = memo.get(key)
            if v is None:
                v = memo[key] = fn(*a)
            return v
        self.__dict__[name] = q
        return q


Q = _SizeQueries(L)

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
GRID_EXPLICIT, GRID_UNET, GRID_AFFINE = 0, 1, 2
GAN_MODES = {"vanilla": 0, "lsgan": 1, "wgangp": 2}


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)      # (the raw handle without building a torch.cuda.Stream object:
                                                                         # 768 calls per step at ~9 us each were 1.2 ms of host time)


def _stream():
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError("nemar_amd ops are fp32 only, got %s" % t.dtype)
    if not t.is_cuda:
        raise RuntimeError("nemar_amd ops need a GPU tensor (there is no CPU path in the product)")
    return t.contiguous()


import collections
import contextlib

_timer = None


def set_kernel_timer(timer):
    """Optional measurement hook (bench.py): `timer.bracket(tag)` returns a context manager that records HIP events
    on the launch stream around one C-ABI launch.  None disables it (default)."""
    global _timer
    _timer = timer


def _span(tag):
    return _timer.bracket(tag) if _timer is not None else contextlib.nullcontext()


_ws_cache = {}
_pinned = [False]      # a captured hipGraph holds the addresses of the buffers below: they may still grow, the old ones are kept alive
_retired = []


def pin_workspaces(on):
    """While a captured step graph is live (NEMARModel.enable_step_graph) a regrown workspace / scratch arena must not FREE the buffer
    the graph's launches point at: the replaced buffers are parked instead of released."""
    _pinned[0] = bool(on)
    if not on:
        _retired.clear()


def _retire(buf):
    if _pinned[0] and buf is not None:
        _retired.append(buf)


_lane = [0]            # 0: the compute stream; 1: the side stream of the weight-gradient branch (its own workspace and arena)


def _workspace(nbytes, device):
    """Grow-only scratch buffer per device and stream lane.  Kernels are stream-ordered, so one buffer serves consecutive ops."""
    key = (device, _lane[0])
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        _retire(buf)
        buf = torch.empty(max(int(nbytes) // 4 + 64, 1 << 20), dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


# ---- the weight-gradient branch on a side stream ---------------------------------------------------------------------------------
# Backward of a convolution = data gradient (feeds the rest of the chain) + weight gradient (feeds only the optimizer).  The second is
# issued on a side HIP stream: its HBM-bound support passes, slab sums and bias reductions run beside the matrix-pipe-bound kernels of
# the main chain (tools/overlap_probe.py: 500 -> 450 us per residual-block layer).  Rules that keep it race-free:
#   * the side stream waits for everything issued so far (g, x and the gy planes exist) before a branch starts;
#   * it has its own workspace / arena lane; the gy-planes buffers written by the main stream are a ring guarded by events;
#   * every tensor a branch reads is kept alive until the side stream has FINISHED that branch — the caching allocator would otherwise
#     hand a block freed on the main stream to a later main-stream allocation while the side stream still reads it.  Each branch ends
#     with an event; the next fork drops the tensors of every branch whose event has completed (a host-side query, no wait), so the
#     activations / gradients of the pass are released layer by layer as without the side stream (at most the side stream's backlog is
#     held; bench.py reports the peak).  Inside a graph capture nothing can be queried: there the tensors stay until join_side();
#   * a weight's gradient accumulations stay on ONE lane within a pass (read-modify-write of the same buffer from two unordered
#     streams otherwise): a weight that was accumulated on the side stream and is then applied by a layer without a data gradient
#     (which would take the compute lane) makes the compute stream wait for the side stream first;
#   * join_side() (before an optimizer step / at the end of a phase) makes the main stream wait for the side stream.
_side_on = [os.environ.get("NEMAR_SIDE_STREAM", "1") != "0"]
_SIDE_K7 = os.environ.get("NEMAR_SIDE_K7", "1") != "0"
_KEEP_ALL = os.environ.get("NEMAR_SIDE_KEEP_ALL", "0") == "1"     # (diagnostic: kept tensors released at the join only, as in round 4)
_side_streams = {}
_side_keep = collections.deque()     # (event recorded on the side stream behind a branch | None inside a capture, the tensors it reads)
_side_touched = set()                # id() of the gradient buffers the side lane has accumulated into since the last join
_side_busy = [False]
_event_pool = []                     # completed branch events, reused (a fresh torch.cuda.Event per backward convolution is a hipEventCreate each)
_main_streams = {}     # device -> the compute stream the last branch forked from
_side_cb = [False]     # join_side is queued as an end-of-backward callback of the running autograd pass


def side_stream(on=None):
    """Switch the side-stream weight-gradient branch on / off (returns the previous setting); joins first."""
    prev = _side_on[0]
    if on is not None:
        join_side()
        _side_on[0] = bool(on)
    return prev


def _side_stream_of(device):
    st = _side_streams.get(device)
    if st is None:
        # (NEMAR_SIDE_PRIORITY: tools/prio_experiment.py — HIP queue priorities, two levels on this stack, move the step by <= 1 %)
        prio = os.environ.get("NEMAR_SIDE_PRIORITY")
        st = _side_streams[device] = torch.cuda.Stream(device=device, priority=int(prio)) if prio is not None else torch.cuda.Stream(device=device)
    return st


class _on_side:
    """`with _on_side(device, keep...)`: the launches inside go to the side stream (lane 1) after everything issued so far"""

    def __init__(self, device, *keep):
        self.device, self.keep = device, keep

    def __enter__(self):
        if not _side_on[0]:
            self.ctx = None
            return self
        side = _side_stream_of(self.device)
        main = torch.cuda.current_stream(self.device)
        _main_streams[self.device] = main
        side.wait_stream(main)
        if not torch.cuda.is_current_stream_capturing() and not _KEEP_ALL:
            while _side_keep and _side_keep[0][0] is not None and _side_keep[0][0].query():
                _event_pool.append(_side_keep.popleft()[0])        # that branch has run: its tensors may go back to the allocator
        self.kept = [t for t in self.keep if t is not None]
        _side_busy[0] = True
        if not _side_cb[0]:
            # the pass that issued a branch joins it when it ends: whoever reads .grad after backward() sees a single stream again
            torch.autograd.Variable._execution_engine.queue_callback(join_side)
            _side_cb[0] = True
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        _lane[0] = 1
        return self

    def __exit__(self, *a):
        if self.ctx is not None:
            ev = None
            if not torch.cuda.is_current_stream_capturing():
                ev = _event_pool.pop() if _event_pool else torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
            _side_keep.append((ev, self.kept))
            _lane[0] = 0
            self.ctx.__exit__(*a)


def _main_lane_grad(buf, device):
    """A weight-gradient accumulation about to be issued on the COMPUTE lane into `buf`: if the side lane has accumulated into the same
    buffer since the last join (a weight applied twice in one pass, once with and once without a data gradient), order the compute
    stream behind the side stream first."""
    if _side_busy[0] and id(buf) in _side_touched:
        side = _side_streams.get(device)
        if side is not None:
            torch.cuda.current_stream(device).wait_stream(side)


def order_current_after_both(device):
    """Called on either lane: the current stream is ordered after everything issued so far on the compute stream AND on the side stream
    (GradSync records its bucket-complete event right after: a bucket may hold gradients written from both)."""
    if not _side_busy[0]:
        return
    side = _side_streams.get(device)
    main = _main_streams.get(device)
    if side is None or main is None:
        return
    cur = torch.cuda.current_stream(device)
    other = main if cur == side else side
    cur.wait_stream(other)


def join_side():
    """The compute stream waits for the side stream's branches; the tensors kept alive for them are released."""
    if _side_busy[0]:
        for dev, side in _side_streams.items():
            torch.cuda.current_stream(dev).wait_stream(side)
        _side_busy[0] = False
        for ring in _garena.values():          # every consumer is behind the compute stream now: the ring's guards are moot (and an
            for slot in ring:                  # event recorded outside a graph capture must not be waited for inside one)
                slot[1] = None
    _side_cb[0] = False
    _side_keep.clear()
    _side_touched.clear()


_scratch_need = {}     # layer shape -> nemar_conv2d_scratch bytes (depends on the nemar_tune switches: ops.tune clears it)
_arena = {}            # device -> the transient scratch arena handed to the wide-layer calls (nemar_conv_extras.scratch)


def _conv_scratch(N, H, W, K, C, R, S, stride, pad, device):
    """-> the scratch arena (tensor) this layer's fp16 x 3 route wants (split source planes of the wide 3x3 layers, csrc/conv_split16.hip),
    or None.  Grow-only per device; stream-ordered like _workspace.  Handed to the library WITH THE CALL (nemar_conv2d_*_ex): nothing is
    registered process-wide."""
    key = (N, H, W, K, C, R, S, stride, pad)
    need = _scratch_need.get(key)
    if need is None:                      # (a host-bound small config pays for every ctypes call: ask once per shape)
        need = _scratch_need[key] = L.conv2d_scratch(N, H, W, K, C, R, S, stride, pad)
    if not need:
        return None
    akey = (device, _lane[0])             # (the side stream's weight-gradient branch has its own arena)
    buf = _arena.get(akey)
    if buf is None or buf.numel() * 4 < need:
        _retire(buf)
        buf = torch.empty(int(need) // 4 + 64, dtype=torch.float32, device=device)
        _arena[akey] = buf
    return buf


_garena = {}           # device -> ring of [buffer, event of the side-stream consumer or None] for the gy planes
_garena_next = {}
_gplanes_need = {}
GY_RING = int(os.environ.get("NEMAR_GY_RING", "4"))
_GY_HANDOVER = os.environ.get("NEMAR_GY_HANDOVER", "1") == "1"      # (NEMAR_GY_HANDOVER=0: A/B of the schedule — the side stream splits gy itself)


def _gy_planes(N, C, H, W, K, R, S, stride, pad, pad_mode, device):
    """-> (buffer, ring slot) in which the data-gradient call of this layer leaves the operand planes of gy its weight-gradient call
    takes (nemar_conv_extras.gy_planes_out -> .src2_planes: gy is split once for both), or (None, None).  A ring of GY_RING buffers per
    device: the weight gradient may run on the side stream while the compute stream writes the next layer's planes; a slot is handed out
    again only after the compute stream has waited for the event its last consumer recorded."""
    if _side_on[0] and not _GY_HANDOVER:
        # (A/B of the schedule.)  The round-5 build ran for a while WITHOUT the hand-over beside the side stream: with it, ~1 % of the steps of
        # config 3 showed one 32 x 32 tile of one wide layer's weight gradient, for one tap row, off by 0.5 - 3 % of the tensor's maximum.  The
        # cause was not the hand-over but WHEN it lets wgrad_split16_kernel start — next to the compute stream's split_dual_kernel: that kernel,
        # staged by LDS-DMA, misread fragments while an LDS-active workgroup of another kernel shared its CU.  It stages through registers now
        # (csrc/conv_split16_wgrad.hip XREG, csrc/common.h) and the hand-over is back on.
        return None, None
    key = (N, C, H, W, K, R, S, stride, pad, pad_mode)
    need = _gplanes_need.get(key)
    if need is None:
        need = _gplanes_need[key] = L.conv2d_gy_planes_bytes(N, C, H, W, K, R, S, stride, pad, pad_mode)
    if not need:
        return None, None
    ring = _garena.setdefault(device, [[None, None] for _ in range(GY_RING)])
    k = _garena_next.get(device, 0)
    _garena_next[device] = (k + 1) % GY_RING
    slot = ring[k]
    if slot[1] is not None:
        torch.cuda.current_stream(device).wait_event(slot[1])
        slot[1] = None
    if slot[0] is None or slot[0].numel() * 4 < need:
        _retire(slot[0])
        slot[0] = torch.empty(int(need) // 4 + 64, dtype=torch.float32, device=device)
    return slot[0], slot


def _extras(arena=None, src_max=None, src2_max=None, planes=None, gy_out=None, src2_planes=None, addend=None, out_max=None, bias_partials=None):
    """nemar_conv_extras for one call: the side inputs of the wide-layer route, or None when there are none"""
    if arena is None and src_max is None and src2_max is None and planes is None and addend is None:
        return None
    e = _lib.ConvExtras()
    if addend is not None:
        e.addend = addend.data_ptr()
    if bias_partials is not None:
        e.bias_partials = bias_partials.data_ptr()
    if out_max is not None:
        e.out_max_words = out_max.data_ptr()
    if arena is not None:
        e.scratch, e.scratch_bytes = arena.data_ptr(), arena.numel() * 4
    if src_max is not None:
        e.src_max_words, e.src_max_count = src_max.data_ptr(), src_max.numel()
    if src2_max is not None:
        e.src2_max_words, e.src2_max_count = src2_max.data_ptr(), src2_max.numel()
    if planes is not None:
        e.src_planes = planes.data_ptr()
    if gy_out is not None:
        e.gy_planes_out, e.gy_planes_bytes = gy_out.data_ptr(), gy_out.numel() * 4
    if src2_planes is not None:
        e.src2_planes = src2_planes.data_ptr()
    return ctypes.byref(e)


_absmax_pool = {}      # device -> [zero-filled int32 tensor, next free word]: nemar_absmax wants its output word zero on entry


MAX_PARTIALS = 2048      # include/nemar_hip.h NEMAR_MAX_PARTIALS


def _max_words(n, device):
    """a NEMAR_MAX_WORDS(n) buffer for a producer's per-sample maxima: [n results | n x 2048 partial words], no initialisation needed"""
    return torch.empty(n * (1 + MAX_PARTIALS), dtype=torch.int32, device=device)


# producers (InstanceNorm forward / backward, dropout) publish the per-sample maxima of what they write when the consumer is likely to
# be one of the wide fp16 x 3 layers: the tensor carries the words as `_nemar_absmax` (Python attributes survive autograd in both
# directions as long as the tensor itself is handed on), and the convolution takes them instead of running a max pass
def _wants_max(t):
    return t.dim() == 4 and 128 <= t.shape[1] <= MAX_PARTIALS and t.shape[1] % 16 == 0 and t.shape[0] <= 256


def _tag_max(t, words, lazy=False):
    # (with the tensor's version: autograd may accumulate another gradient INTO this tensor in place — the words are then stale)
    # lazy: the producer ran under _lazy_max() — the words hold the marker, not the maxima (csrc/max_words.h): a tag of its own that only
    # the consumers which reduce the partial words themselves (the two InstanceNorm producers of the residual blocks) ask for
    if lazy:
        t._nemar_absmax_lazy = (words[:t.shape[0]], t._version)
    else:
        t._nemar_absmax = (words[:t.shape[0]], t._version)


def _absmax_word(t, lazy_ok=False):
    lazy = getattr(t, '_nemar_absmax_lazy', None)
    if lazy is not None and not (lazy[1] == t._version and lazy[0].numel() == t.shape[0]):
        lazy = None
    if lazy_ok and lazy is not None:
        return lazy[0]
    have = getattr(t, '_nemar_absmax', None)
    if have is not None and have[1] == t._version and have[0].numel() == t.shape[0]:
        return have[0]
    if lazy is not None:
        # a producer left the reduction of its partial words to whoever needs the maxima (the InstanceNorm passes publish them "in case the
        # consumer is a wide-route convolution": most are not): reduce them now, once — nemar_max_words_finalize leaves finalized words alone
        L.max_words_finalize(_p(lazy[0]), int(t.shape[0]), _stream())
        t._nemar_absmax = lazy
        return lazy[0]
    return _absmax_word_compute(t)


_LAZY_MAX = os.environ.get("NEMAR_LAZY_MAX", "1") != "0"      # (0: every producer finalizes its words — for A/B runs)


class _lazy_max:
    """`with _lazy_max(on)`: producers of per-sample maxima called inside leave the reduction of their partial words to the consumer
    (nemar_set_max_words_lazy; one launch less per call on the chain the step waits for).  Only where the consumer is known to be
    nemar_instnorm_fwd_planes / nemar_instnorm_bwd_planes."""

    def __init__(self, on=True):
        self.on = on and _LAZY_MAX

    def __enter__(self):
        if self.on:
            L.set_max_words_lazy(1)
        return self

    def __exit__(self, *a):
        if self.on:
            L.set_max_words_lazy(0)


def _absmax_word_compute(t):
    """max |t| PER SAMPLE as the N-word tensor nemar_absmax_hint takes (the fp16 split of the wide 3x3 layers scales every sample by
    a power of two derived from its own maximum).  Computed once per tensor and shared by the calls that take it as a source.
    Words come out of a pre-zeroed pool: one fill launch per 4096 of them instead of one per tensor."""
    n = int(t.shape[0])
    if _step_params["on"]:
        # device-parameter mode (a step that is, or may become, a captured graph): the words must be zeroed IN-STREAM by every
        # execution — a pooled word zeroed once at allocation would carry a running maximum from replay to replay
        word = torch.zeros(n, dtype=torch.int32, device=t.device)
        L.absmax_samples(_p(t), n, t.numel() // n, _p(word), _stream())
        return word
    pool = _absmax_pool.get(t.device)
    if pool is None or pool[1] + n > pool[0].numel():
        pool = [torch.zeros(4096, dtype=torch.int32, device=t.device), 0]
        _absmax_pool[t.device] = pool
    word = pool[0][pool[1]:pool[1] + n]
    pool[1] += n
    L.absmax_samples(_p(t), n, t.numel() // n, _p(word), _stream())
    return word


_zero_ws_cache = {}


def _zeroed_workspace(nbytes, key):
    """Buffer whose users keep a prefix ALL-ZERO between calls (grid_sample backward's fixed-point accumulator returns it
    zero-filled, so it is memset exactly once, when allocated).  How long that prefix is depends on the problem shape, so
    there is one buffer per (device, shape) `key` — a shared grow-only buffer would hand scratch bytes of a small problem
    to a larger one as "zeroed"."""
    buf = _zero_ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.zeros(int(nbytes) // 4 + 64, dtype=torch.float32, device=key[0])
        _zero_ws_cache[key] = buf
    return buf


# ---- packed-weight cache ---------------------------------------------------------------------------------------------
# The MFMA kernels read weights in a packed [reduction][channel] layout.  Packing is a tiny kernel, but a step applies
# every weight tensor several times (T: 2 forward + 2 backward passes, D: 5), so each (weight, direction) keeps its own
# packed workspace until the values change: FlatAdam.step() bumps the epoch of ITS OWN parameters (the discriminator's
# update does not re-pack the translation / registration nets), invalidate_packed_weights() bumps the global one, in-place
# torch ops bump tensor._version.
class PackEpoch:
    """Value generation of a group of weights: bumped when their values change outside torch's version counter.  `plan`: the
    weight-pack plan of the optimizer that owns them (None: none)."""
    __slots__ = ('n', 'plan')

    def __init__(self):
        self.n = 0
        self.plan = None


_param_epoch = PackEpoch()      # everything (re-initialisation through .data, checkpoint loads)
_pack_cache = {}

# ---- weight-pack plans (include/nemar_hip.h: nemar_pack_plan_*) ---------------------------------------------------------------------
# Lazily, every (weight, direction) re-packs itself at its first use after its optimizer stepped: ~240 launches of 4-6 us per training
# step.  With plans the pack jobs of an optimizer's weights are RECORDED the first time they run (the library notes the launches its
# convolution entry points issue while nemar_pack_plan_record is in effect) and from then on FlatAdam.step() re-runs all of them right
# behind the Adam kernel in <= 5 launches; the convolutions then find their images ready (prepacked = 1).  The packed images are the
# same bits either way.  NEMAR_PACK_PLAN=0 switches plans off (A/B, tests).
_plans_on = [os.environ.get("NEMAR_PACK_PLAN", "1") != "0"]
_plan_gen = [0]                 # bumped when every plan is forgotten (route switches, re-initialised weights): recorded entries lapse
_plan_owners = weakref.WeakSet()
_plan_ids = [0]


def pack_plans(on):
    """Switch weight-pack plans on / off (off: every weight packs itself at first use, as before); forgets the recorded plans."""
    _plans_on[0] = bool(on)
    _reset_plans()


def _reset_plans():
    _plan_gen[0] += 1
    for o in list(_plan_owners):
        L.pack_plan_reset(o._plan)


def invalidate_packed_weights():
    _param_epoch.n += 1
    _reset_plans()


def tune(key, value):
    """nemar_tune through the packed-weight cache: several switches (tile family, split-16 route) change the packed image."""
    L.tune(key, value)
    Q._memo.clear()
    _scratch_need.clear()
    _gplanes_need.clear()            # (the gy-planes hand-over depends on the route switches too)
    _fusable.clear()
    _xplanes_need.clear()
    invalidate_packed_weights()


class _PackEntry:
    __slots__ = ('buf', 'token', 'ref', 'plan_gen')

    def __init__(self, buf, token, ref):
        self.buf, self.token, self.ref, self.plan_gen = buf, token, ref, -1


class _record:
    """`with _record(plan):` — the pack launches of the library call inside are recorded as jobs of `plan` (None: nothing happens)"""
    __slots__ = ('plan',)

    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        if self.plan is not None:
            L.pack_plan_record(self.plan)

    def __exit__(self, *a):
        if self.plan is not None:
            L.pack_plan_record(-1)


def _packed(weight, kind, nbytes):
    """-> (workspace tensor, prepacked flag, plan to record the pack into or None) for `weight` used in direction `kind`."""
    key = (id(weight), kind)
    own = getattr(weight, '_pack_epoch', None)
    # (the library's config epoch: which kernel family a shape routes to — and so the format of its packed image — depends on the
    # nemar_tune switches and on the registered arena, whoever changed them)
    static = (_param_epoch.n, weight._version, weight.data_ptr(), tuple(weight.shape), L.config_epoch())
    token = (static, own.n if own is not None else 0)
    ent = _pack_cache.get(key)
    # id() and device addresses are recycled once a tensor dies: an entry is only valid for the very object it was
    # made for (weak reference), with unchanged values (epoch, _version) at an unchanged address
    mine = ent is not None and ent.ref() is weight and ent.buf.numel() * 4 >= nbytes
    if mine and ent.token == token:
        return ent.buf, 1, None
    # an image that is a job of its optimizer's plan was rebuilt right behind the optimizer's last step
    if mine and ent.plan_gen == _plan_gen[0] and ent.token[0] == static:
        return ent.buf, 1, None
    buf = ent.buf if mine else torch.empty(int(nbytes) // 4 + 64, dtype=torch.float32, device=weight.device)
    new = _PackEntry(buf, token, weakref.ref(weight, lambda _r, k=key: _pack_cache.pop(k, None)))
    plan = None
    # record the pack as a plan job: once per (image buffer, plan generation), not while a captured graph holds an older image of the
    # plan (its replays would not rebuild the newcomer), not for weights without an optimizer
    if _plans_on[0] and own is not None and own.plan is not None and not _pinned[0]:
        if mine and ent.plan_gen == _plan_gen[0]:
            new.plan_gen = ent.plan_gen                     # (already a job: same buffer)
        else:
            plan = own.plan
            new.plan_gen = _plan_gen[0]
    _pack_cache[key] = new
    return buf, 0, plan


def grad_ready(param):
    """A gradient contribution of `param` has been launched (weight / bias gradient kernels accumulate straight into
    param.grad): lets nemar_amd.distributed.GradSync start the all-reduce of a bucket as soon as it is complete."""
    gs = getattr(param, '_grad_sync', None)
    if gs is not None:
        gs.ready(param)


def _note_use(param):
    """`param` is applied once more in a forward pass (GradSync counts how many gradient contributions to expect)."""
    gs = getattr(param, '_grad_sync', None)
    if gs is not None and param is not None:
        gs.note_use(param)


def _grad_buffer(param):
    """param.grad as an accumulation target.  Parameters owned by a FlatAdam accumulate into their view of the optimizer's
    flat gradient buffer: if user code re-seated or cleared `.grad` (`p.grad = None`, `zero_grad(set_to_none=True)` of some
    wrapper), the view is put back — the kernels must never write into a tensor the optimizer and the all-reduce do not see.
    Stand-alone parameters get a zero-filled tensor on first use."""
    view = getattr(param, '_flat_grad', None)
    if view is not None:
        g = param.grad
        if g is None or g.data_ptr() != view.data_ptr():
            if g is not None:
                view.add_(g)                 # keep whatever was accumulated outside (normally nothing)
            param.grad = view
        return view
    if param.grad is None:
        param.grad = torch.zeros_like(param, memory_format=torch.contiguous_format)
    return param.grad


# ------------------------------------------------------------------------------------------------------
class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, x2, weight, bias, stride, pad, pad_mode, act, slope, wshape, with_skip=False):
        x, x2, w, b = _c(x), _c(x2), _c(weight), _c(bias)
        ctx.with_skip = bool(with_skip)
        if with_skip:
            ctx.set_materialize_grads(False)
        if wshape is not None:
            w = w.view(wshape)          # e.g. nn.Linear's [out,in] seen as a 1x1 conv; gradients keep the param's shape
        N, C0, H, W = x.shape
        C1 = 0 if x2 is None else x2.shape[1]
        K, C, R, S = w.shape
        if C != C0 + C1:
            raise ValueError("conv2d: weight expects %d input channels, got %d+%d" % (C, C0, C1))
        OH = (H + 2 * pad - R) // stride + 1
        OW = (W + 2 * pad - S) // stride + 1
        y = torch.empty((N, K, OH, OW), dtype=torch.float32, device=x.device)
        wsb = Q.conv2d_fwd_workspace(N, H, W, K, C, R, S, stride, pad)
        ws, hit, plan = _packed(weight, ('fwd', stride, pad, N, H, W), wsb)
        arena = _conv_scratch(N, H, W, K, C, R, S, stride, pad, x.device) if x2 is None else None
        tag = 'igemm_fwd_resblock' if (K == 256 and C == 256 and R == 3 and pad_mode == PAD_REFLECT) else None
        with (_span(tag) if tag else contextlib.nullcontext()):
            xmax = ready = None
            if arena is not None:     # max |x| once: this call and the weight gradient in backward both scale x by it
                ready = _planes_of(x) if (pad_mode == PAD_REFLECT and R == 3 and S == 3 and stride == 1 and pad == 1) else None
                xmax = ready[1] if ready is not None else _absmax_word(x)     # (planes from the producer: scaled by its a-priori bound words)
            with _record(plan):
                L.conv2d_fwd_ex(_p(x), C0, _p(x2), C1, _p(w), _p(b), _p(y), N, H, W, K, R, S, stride, pad, pad_mode, act,
                                slope, _p(ws), wsb, hit, _stream(), _extras(arena, xmax, None, ready[0] if ready is not None else None))
        ctx.xmax = xmax
        # the producer's pixel-major planes of x (same scale words as `ready`): the weight gradient then skips its split pass over x
        xp = _xplanes_of(x) if ready is not None else None
        ctx.xplanes = xp[0] if (xp is not None and xp[1] is ready[1]) else None
        ctx.grad_from = int(getattr(x, '_nemar_grad_from', 0)) if x2 is None else 0
        ctx.save_for_backward(x, x2, w, y if act != ACT_NONE else None)
        ctx.weight, ctx.bias = weight, bias
        _note_use(weight)
        if bias is not None:
            _note_use(bias)
        ctx.cfg = (stride, pad, pad_mode, act, slope)
        if with_skip:
            # a second handle of the input for the caller's skip connection (ResnetBlock: out = x + conv_block(x)): its gradient comes back
            # to THIS node and is added inside the data gradient's last pass where the route has one (nemar_conv2d_bwd_data_addend_ok)
            return y, x.view_as(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gskip=None):
        x, x2, w, y = ctx.saved_tensors
        stride, pad, pad_mode, act, slope = ctx.cfg
        ctx.gskip = _c(gskip) if (ctx.with_skip and ctx.needs_input_grad[0]) else None
        if gy is None:                          # (only the skip handle was used)
            return (ctx.gskip,) + (None,) * 10
        gy = _c(gy)
        N, C0, H, W = x.shape
        C1 = 0 if x2 is None else x2.shape[1]
        K, C, R, S = w.shape
        OH, OW = gy.shape[2:]
        st = _stream()
        if act != ACT_NONE:
            g = torch.empty_like(gy)
            L.act_bwd(_p(gy), _p(y), _p(g), gy.numel(), act, slope, st)
        else:
            g = gy
        need_x, need_x2, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], \
            ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        gmax = _absmax_word(g) if ctx.xmax is not None else None      # wide layer: max |gy| once for the data and the weight gradient
        return _Conv2d._backward_body(ctx, x, x2, w, g, N, C0, C1, H, W, K, C, R, S, OH, OW, stride, pad, pad_mode, st,
                                      need_x, need_x2, need_w, need_b, gmax)

    @staticmethod
    def _backward_body(ctx, x, x2, w, g, N, C0, C1, H, W, K, C, R, S, OH, OW, stride, pad, pad_mode, st, need_x, need_x2, need_w,
                       need_b, gmax):
        gx = gx2 = gpl = gslot = None
        if need_x or need_x2:
            gx = torch.empty_like(x) if need_x else None
            gx2 = torch.empty_like(x2) if (need_x2 and x2 is not None) else None
            if x2 is not None and gx2 is None:
                # kernel splits channels [0,C0) | [C0,C); a missing second half still needs a destination
                gx2 = torch.empty_like(x2)
            wsb = Q.conv2d_bwd_data_workspace(N, C, H, W, K, R, S, stride, pad, pad_mode)
            if pad_mode == PAD_REFLECT and pad > 0 and x2 is not None:
                raise NotImplementedError("reflect-padded conv over a concatenated input has no data-gradient kernel")
            # the packed image depends on which source halves are differentiated (channel skip) and on the geometry
            n0 = getattr(ctx, 'grad_from', 0)
            if 0 < n0 < N:
                # only samples [n0:] of the input are differentiated (ops.grad_from: T's batch [real_A ; R(real_A)]): the data
                # gradient of the others is never read — run the kernel on the sub-batch
                _zero(gx[:n0])
                Nd, gd, gxd = N - n0, g[n0:], gx[n0:]
            else:
                Nd, gd, gxd = N, g, gx
            wsb = Q.conv2d_bwd_data_workspace(Nd, C, H, W, K, R, S, stride, pad, pad_mode)
            ws, hit, plan = _packed(ctx.weight, ('dgrad', stride, pad, pad_mode, need_x, Nd, H, W), wsb)
            arena = _conv_scratch(Nd, H, W, K, C, R, S, stride, pad, g.device)
            if arena is not None and need_w and Nd == N and gmax is not None and x2 is None:
                # wide layer whose weight gradient follows: the pass that splits gy for this call leaves its planes for that one too
                gpl, gslot = _gy_planes(N, C, H, W, K, R, S, stride, pad, pad_mode, g.device)
            rb = K == 256 and C == 256 and R == 3 and pad_mode == PAD_REFLECT         # (bench.py: operator-level roofline of the residual blocks)
            gskip = getattr(ctx, 'gskip', None)
            # (layers the wide route never takes: their data gradient ends with a fold pass or it does not — no arena-dependent routing)
            ride = (gskip is not None and gx is not None and gx2 is None and Nd == N and arena is None
                    and Q.conv2d_scratch(N, H, W, K, C, R, S, stride, pad) == 0
                    and Q.conv2d_bwd_data_addend_ok(N, C, H, W, K, R, S, stride, pad, pad_mode) == 1)
            with _record(plan), (_span('dgrad_resblock') if rb else contextlib.nullcontext()):
                L.conv2d_bwd_data_ex(_p(gd), _p(w), None, ACT_NONE, 0.0, _p(gxd), C0, _p(gx2), C1, Nd, H, W, K, OH, OW, R, S,
                                     stride, pad, pad_mode, _p(ws), wsb, hit, st,
                                     _extras(arena, gmax[n0:] if (gmax is not None and Nd != N) else gmax, gy_out=gpl,
                                             addend=gskip if ride else None))
            if gskip is not None and gx is not None and not ride:
                L.add2(_p(gx), _p(gskip), _p(gx), gx.numel(), st)       # (a route without a last pass to ride in: one more launch)
            if gpl is not None and not L.last_gy_planes():
                gpl = None
            if not need_x2:
                gx2 = None
        want_b = need_b and ctx.bias is not None
        if need_w:
            # the weight-gradient branch feeds only the optimizer: side stream (after everything issued so far).  Every layer, the 7x7
            # stem / head included: what round 4 saw go wrong with those two on the side queue (grid_sample's grid gradient, on the
            # compute stream at that moment, came out wrong in lanes 48..63) was a packed-FP32 instruction form of THAT kernel
            # miscomputing next to another kernel's MFMAs — the library is built without such instructions now.
            # NEMAR_SIDE_K7=0 keeps the 7x7 layers on the compute stream (A/B of the schedule only).
            _use = R != 7 or _SIDE_K7
            # a layer without a data gradient (the first layer of a net: nothing left for the compute stream to do) keeps its weight
            # gradient there — it is the last one of the pass, and the compute stream would only wait for it at the join
            if not (need_x or need_x2):
                _use = False
            # schedule hint of the model (NLayerDiscriminator: the side lane is the longer one of the D phase, the compute stream idles at
            # its join): this layer's weight gradient stays on the compute stream.  Same kernels, same accumulation order: same bits.
            if getattr(ctx.weight, '_nemar_wgrad_main', False):
                _use = False
            gwbuf = _grad_buffer(ctx.weight)
            gbbuf = _grad_buffer(ctx.bias) if want_b else None       # (the bias gradient rides along with the weight gradient: same lane)
            if _use and _side_on[0]:
                _side_touched.add(id(gwbuf))
                if gbbuf is not None:
                    _side_touched.add(id(gbbuf))
            else:
                _main_lane_grad(gwbuf, g.device)
                if gbbuf is not None:
                    _main_lane_grad(gbbuf, g.device)
            rbw = K == 256 and C == 256 and R == 3 and pad_mode == PAD_REFLECT
            xpl = getattr(ctx, 'xplanes', None)
            with (_on_side(g.device, x, x2, g, gmax, ctx.xmax, gpl, xpl) if _use else contextlib.nullcontext()), \
                    (_span('wgrad_resblock') if rbw else contextlib.nullcontext()):
                gb = _grad_buffer(ctx.bias) if want_b else None      # bias gradient rides along in the same pass
                wsb = Q.conv2d_bwd_weight_workspace(N, C, H, W, K, OH, OW, R, S, stride, pad)
                arena = _conv_scratch(N, H, W, K, C, R, S, stride, pad, g.device)
                L.conv2d_bwd_weight_ex(_p(x), C0, _p(x2), C1, _p(g), _p(gwbuf), _p(gb), N, H, W, K, OH, OW,
                                       R, S, stride, pad, pad_mode, _p(_workspace(wsb, g.device)), wsb, _stream(),
                                       _extras(arena, ctx.xmax, gmax, planes=xpl, src2_planes=gpl))
                if gslot is not None and _side_on[0]:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(g.device))
                    gslot[1] = ev
                grad_ready(ctx.weight)
                if want_b:
                    grad_ready(ctx.bias)
        elif want_b:
            gbbuf = _grad_buffer(ctx.bias)
            _main_lane_grad(gbbuf, g.device)          # (a bias the side lane has accumulated into in this pass: order behind it first)
            _bias_grad(g, gbbuf, N, K, OH * OW, st)
            grad_ready(ctx.bias)
        if gx is None and getattr(ctx, 'gskip', None) is not None:
            gx = ctx.gskip
        return gx, gx2, None, None, None, None, None, None, None, None, None


def _bias_grad(g, gb, N, C, HW, st):
    wsb = Q.bias_grad_workspace(N, C, HW)
    L.bias_grad(_p(g), _p(gb), N, C, HW, _p(_workspace(wsb, g.device)), wsb, st)


def conv2d(x, weight, bias=None, stride=1, pad=0, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.2, x2=None, wshape=None):
    """act(conv2d(pad(cat(x, x2)), weight) + bias).  pad_mode PAD_REFLECT == nn.ReflectionPad2d(pad) + conv.
    `weight` must be the leaf Parameter (its .grad is the accumulation target); `wshape` reinterprets it as 4-D."""
    return _Conv2d.apply(x, x2, weight, bias, stride, pad, pad_mode, act, slope, wshape)


def conv2d_with_skip(x, weight, bias=None, stride=1, pad=0, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.2):
    """-> (conv2d(x, ...), a second handle of x for the caller's skip connection).  The skip's gradient returns to the convolution's node
    and is added in the last pass of its data gradient (one launch less than a separate sum; ops.fork where autograd is off)."""
    if not (_own_nodes and torch.is_grad_enabled() and x.requires_grad):
        return _Conv2d.apply(x, None, weight, bias, stride, pad, pad_mode, act, slope, None), x
    y, skip = _Conv2d.apply(x, None, weight, bias, stride, pad, pad_mode, act, slope, None, True)
    for name in _TAGS:
        v = getattr(x, name, None)
        if v is not None:
            setattr(skip, name, v)
    return y, skip


class _ConvTranspose2d(Function):
    """nn.ConvTranspose2d(Ci -> Co, k, stride, pad, output_padding): forward is the data-gradient kernel of the
    conv whose weight tensor is weight[Ci][Co][R][S]; backward-data is that conv's forward; backward-weight is its
    weight gradient with the roles of input and output-gradient swapped."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, out_pad, act, slope):
        x, w, b = _c(x), _c(weight), _c(bias)
        N, Ci, H, W = x.shape
        _, Co, R, S = w.shape
        Ho = (H - 1) * stride - 2 * pad + R + out_pad
        Wo = (W - 1) * stride - 2 * pad + S + out_pad
        y = torch.empty((N, Co, Ho, Wo), dtype=torch.float32, device=x.device)
        wsb = Q.conv2d_bwd_data_workspace(N, Co, Ho, Wo, Ci, R, S, stride, pad, PAD_ZERO)
        ws, hit, plan = _packed(weight, ('convT_fwd', stride, pad, N, Ho, Wo), wsb)
        with _record(plan):
            L.conv2d_bwd_data(_p(x), _p(w), _p(b), act, slope, _p(y), Co, None, 0, N, Ho, Wo, Ci, H, W, R, S, stride, pad,
                              PAD_ZERO, _p(ws), wsb, hit, _stream())
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.weight, ctx.bias = weight, bias
        _note_use(weight)
        if bias is not None:
            _note_use(bias)
        ctx.cfg = (stride, pad, act, slope)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, act, slope = ctx.cfg
        gy = _c(gy)
        N, Ci, H, W = x.shape
        _, Co, R, S = w.shape
        Ho, Wo = gy.shape[2:]
        st = _stream()
        if act != ACT_NONE:
            g = torch.empty_like(gy)
            L.act_bwd(_p(gy), _p(y), _p(g), gy.numel(), act, slope, st)
        else:
            g = gy
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            wsb = Q.conv2d_fwd_workspace(N, Ho, Wo, Ci, Co, R, S, stride, pad)
            ws, hit, plan = _packed(ctx.weight, ('convT_bwd', stride, pad, N, Ho, Wo), wsb)
            with _record(plan):
                L.conv2d_fwd(_p(g), Co, None, 0, _p(w), None, _p(gx), N, Ho, Wo, Ci, R, S, stride, pad, PAD_ZERO, ACT_NONE,
                             0.0, _p(ws), wsb, hit, st)
        if ctx.needs_input_grad[1] or (ctx.needs_input_grad[2] and ctx.bias is not None):
            if _side_on[0] and ctx.needs_input_grad[1]:
                _side_touched.add(id(_grad_buffer(ctx.weight)))
            with _on_side(g.device, g, x):                 # the weight-gradient branch: side stream (see _Conv2d)
                if ctx.needs_input_grad[1]:
                    wsb = Q.conv2d_bwd_weight_workspace(N, Co, Ho, Wo, Ci, H, W, R, S, stride, pad)
                    L.conv2d_bwd_weight(_p(g), Co, None, 0, _p(x), _p(_grad_buffer(ctx.weight)), None, N, Ho, Wo, Ci, H, W, R,
                                        S, stride, pad, PAD_ZERO, _p(_workspace(wsb, g.device)), wsb, _stream())
                    grad_ready(ctx.weight)
                if ctx.needs_input_grad[2] and ctx.bias is not None:
                    _bias_grad(g, _grad_buffer(ctx.bias), N, Co, Ho * Wo, _stream())
                    grad_ready(ctx.bias)
        return gx, None, None, None, None, None, None, None


class _Activation(Function):
    """Stand-alone activation (only where no conv / InstanceNorm epilogue can carry it)."""

    @staticmethod
    def forward(ctx, x, act, slope):
        x = _c(x)
        y = torch.empty_like(x)
        L.act_fwd(_p(x), _p(y), x.numel(), act, slope, _stream())
        ctx.save_for_backward(y)
        ctx.act, ctx.slope = act, slope
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = _c(gy)
        gx = torch.empty_like(gy)
        L.act_bwd(_p(gy), _p(y), _p(gx), gy.numel(), ctx.act, ctx.slope, _stream())
        return gx, None, None


def activation(x, act, slope=0.2):
    return x if act == ACT_NONE else _Activation.apply(x, act, slope)


def conv_transpose2d(x, weight, bias=None, stride=2, pad=1, out_pad=1, act=ACT_NONE, slope=0.2):
    return _ConvTranspose2d.apply(x, weight, bias, stride, pad, out_pad, act, slope)


# ------------------------------------------------------------------------------------------------------
def _planes_ok(x, residual):
    """InstanceNorm output -> fp16 x 3 planes in the same pass (norm_planes.hip): shapes the kernel covers and the wide-layer route takes"""
    N, C, H, W = x.shape
    return _planes_on[0] and _wants_max(x) and W % 4 == 0 and H >= 4 and H * W <= 4096 and _conv_scratch(N, H, W, C, C, 3, 3, 1, 1, x.device) is not None


_planes_on = [os.environ.get("NEMAR_PLANES", "1") != "0"]       # A/B switch for the fused producer (NEMAR_PLANES=0, ops.tune_planes)


def tune_planes(on):
    _planes_on[0] = bool(on)


def _planes_of(t):
    have = getattr(t, '_nemar_planes', None)
    return have if have is not None and have[2] == t._version else None


def _xplanes_of(t):
    have = getattr(t, '_nemar_xplanes', None)
    return have if have is not None and have[2] == t._version else None


_xplanes_on = [os.environ.get("NEMAR_XPLANES", "1") != "0"]      # A/B: the forward producers also write the weight gradient's X planes
_xplanes_need = {}


def _x_planes_buffer(N, C, H, W, device):
    """-> a fresh buffer for the weight gradient's X planes of a [N, C, H, W] activation (3x3 reflect consumer), or None where the
    producer has no such output.  A tensor of its own (not an arena): it lives until the layer's backward pass has read it."""
    if not _xplanes_on[0]:
        return None
    key = (N, C, H, W)
    need = _xplanes_need.get(key)
    if need is None:
        need = _xplanes_need[key] = L.conv2d_x_planes_bytes(N, C, H, W, 3)
    if not need:
        return None
    return torch.empty(int(need), dtype=torch.uint8, device=device)


class _InstanceNorm(Function):
    @staticmethod
    def forward(ctx, x, residual, act, slope, eps, planes, drop_p):
        x, residual = _c(x), _c(residual)
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((N * C, 2), dtype=torch.float32, device=x.device)
        ctx.drop = None
        if planes and _planes_ok(x, residual):
            # the consumer is a 3x3 reflect convolution of the fp16 x 3 route: write its operand planes here, scaled by the a-priori
            # bound sqrt(HW) [/ (1-p)] [+ max |residual|]; dropout drawn in the same pass
            words = _max_words(N, x.device)
            scale_words = torch.empty(N, dtype=torch.int32, device=x.device)
            buf = torch.empty(2 * N * (C // 8) * (H + 4) * (W + 4) * 16, dtype=torch.uint8, device=x.device)
            seed = off = 0
            if drop_p > 0.0:
                _dropout_state["offset"] = (_dropout_state["offset"] + 1) & 0xFFFFFFFF
                seed, off = _dropout_state["seed"], _dropout_state["offset"]
                ctx.drop = (drop_p, seed, off)
            # the weight gradient's pixel-major planes from the same pass (not in no_grad / inference forward passes)
            xbuf = _x_planes_buffer(N, C, H, W, x.device) if any(ctx.needs_input_grad[:2]) else None
            L.instnorm_fwd_planes(_p(x), _p(residual), _p(_absmax_word(residual)) if residual is not None else None, _p(y), _p(stats),
                                  N, C, H, W, eps, act, slope, drop_p, seed, off, _p(buf), _p(scale_words), _p(words), _p(xbuf), _stream())
            _tag_max(y, words)
            y._nemar_planes = (buf, scale_words, y._version)
            if xbuf is not None:
                y._nemar_xplanes = (xbuf, scale_words, y._version)
        else:
            if _wants_max(x) and drop_p <= 0.0:
                words = _max_words(N, x.device)
                with _lazy_max():                 # (the reduction of the partial words: on demand, ops._absmax_word)
                    L.instnorm_fwd_max(_p(x), _p(residual), _p(y), _p(stats), N * C, H * W, eps, act, slope, _p(words), C, _stream())
                _tag_max(y, words, lazy=_LAZY_MAX)
            else:
                L.instnorm_fwd(_p(x), _p(residual), _p(y), _p(stats), N * C, H * W, eps, act, slope, _stream())
            if drop_p > 0.0:
                _dropout_state["offset"] = (_dropout_state["offset"] + 1) & 0xFFFFFFFF
                ctx.drop = (drop_p, _dropout_state["seed"], _dropout_state["offset"])
                _dropout_launch(y, y, *ctx.drop)                       # in place: nothing else has seen y
        ctx.save_for_backward(x, stats)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, stats = ctx.saved_tensors
        act, slope = ctx.cfg
        gy = _c(gy)
        N, C, H, W = x.shape
        gx = None
        if ctx.needs_input_grad[0]:
            g = gy
            if ctx.drop is not None:              # the dropout between the activation and the consumer: mask regenerated
                if ctx.needs_input_grad[1]:
                    raise RuntimeError("instance_norm: dropout and a residual cannot be fused in one call")
                g = torch.empty_like(gy)
                L.dropout(_p(gy), _p(g), gy.numel(), ctx.drop[0], ctx.drop[1], ctx.drop[2], _stream())
            gx = torch.empty_like(x)
            if _wants_max(x):
                words = _max_words(N, x.device)
                with _lazy_max():
                    L.instnorm_bwd_max(_p(x), _p(stats), _p(g), _p(gx), N * C, H * W, act, slope, _p(words), C, _stream())
                _tag_max(gx, words, lazy=_LAZY_MAX)
            else:
                L.instnorm_bwd(_p(x), _p(stats), _p(g), _p(gx), N * C, H * W, act, slope, _stream())
        gres = gy if ctx.needs_input_grad[1] else None
        return gx, gres, None, None, None, None, None


def grad_from(x, n0):
    """Declare that only samples [n0:] of the batch `x` are differentiated (the others are inputs): the first convolution that takes
    `x` computes its data gradient on that sub-batch only."""
    x._nemar_grad_from = int(n0)
    return x


def instance_norm(x, act=ACT_NONE, slope=0.2, residual=None, eps=1e-5, planes=False, dropout_p=0.0):
    """(residual +) dropout(act(InstanceNorm2d(x))) with affine=False, track_running_stats=False.  planes=True: the consumer is a
    3x3 / pad-1 reflect convolution — where the wide-layer fp16 x 3 route applies, its operand planes are written in the same pass."""
    return _InstanceNorm.apply(x, residual, act, slope, eps, bool(planes), float(dropout_p))


# ---- one ResnetBlock of the wide route as ONE autograd node (round 6) -------------------------------------------------------------------
# x + IN(conv3x3(reflect(drop(relu(IN(conv3x3(reflect(x)))))))) — reference models/networks.py:418-446.  As separate nodes the backward pass of
# a block costs, besides its four matrix-pipe kernels: two InstanceNorm backward passes that write fp32 gradients, two split passes that
# read them back (split_dual_kernel), two split passes over the saved activations (split_wgrad_x_kernel), a dropout backward, two bias
# reductions over the fp32 gradients and autograd's add of the skip gradient.  As one node:
#   forward   conv1 -> [IN + ReLU + dropout: planes of conv2's operand AND of its weight gradient's, no fp32 tensor] -> conv2 ->
#             [IN + skip: fp32 output, planes for the next block's conv1 and its weight gradient]
#   backward  [IN2 backward: operand planes of conv2's two gradient calls + bias sums, no fp32 tensor] -> conv2 data gradient (epilogue
#             publishes its per-sample maxima) -> [dropout + ReLU + IN1 backward: planes + bias sums] -> conv1 data gradient
#             (epilogue adds the skip gradient and publishes the maxima the previous block's IN2 backward scales by)
#   the two weight gradients run on the side stream from producer-written planes on both operands: no split pass at all.
_fused_block_on = [os.environ.get("NEMAR_FUSED_BLOCK", "1") != "0"]
_fusable = {}


def fused_blocks(on=None):
    """Switch the one-node ResnetBlock on / off (A/B; returns the previous setting)"""
    prev = _fused_block_on[0]
    if on is not None:
        _fused_block_on[0] = bool(on)
    return prev


def _block_fusable(x, C):
    if not (_fused_block_on[0] and _planes_on[0] and _xplanes_on[0] and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32):
        return False
    N, Cx, H, W = x.shape
    if Cx != C:
        return False
    key = (N, C, H, W, L.config_epoch())
    ok = _fusable.get(key)
    if ok is None:
        ok = (_wants_max(x) and W % 8 == 0 and H >= 4 and H * W <= 4096
              and L.conv2d_bwd_data_fusable(N, C, H, W, C, 3, 3, 1, 1, PAD_REFLECT) == 1
              and L.conv2d_x_planes_bytes(N, C, H, W, 3) > 0
              and L.conv2d_scratch(N, H, W, C, C, 3, 3, 1, 1) > 0)
        _fusable[key] = bool(ok)
    return ok


def _chan_planes_buffer(N, C, H, W, device):
    return torch.empty(2 * N * (C // 8) * (H + 4) * (W + 4) * 16, dtype=torch.uint8, device=device)


class _ResBlock(Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, drop_p, feeds_block, eps):
        x = _c(x)
        N, C, H, W = x.shape
        dev = x.device
        st = _stream()
        arena = _conv_scratch(N, H, W, C, C, 3, 3, 1, 1, dev)
        wsb = Q.conv2d_fwd_workspace(N, H, W, C, C, 3, 3, 1, 1)

        def conv(src_key, planes, words, weight, bias, y):
            ws, hit, plan = _packed(weight, ('fwd', 1, 1, N, H, W), wsb)
            with _span('igemm_fwd_resblock'), _record(plan):
                L.conv2d_fwd_ex(src_key, C, None, 0, _p(weight), _p(bias), _p(y), N, H, W, C, 3, 3, 1, 1, PAD_REFLECT, ACT_NONE, 0.2,
                                _p(ws), wsb, hit, st, _extras(arena, words, None, planes))

        # conv1: operand planes from the producer of x where it wrote them (the previous block / the down-sampling stage's InstanceNorm)
        ready = _planes_of(x)
        xp0 = _xplanes_of(x)
        if ready is not None:
            x_words = ready[1]
            if xp0 is not None and xp0[1] is not ready[1]:
                xp0 = None
        else:
            x_words, xp0 = _absmax_word(x), None
        y1 = torch.empty_like(x)
        conv(_p(x), ready[0] if ready is not None else None, x_words, w1, b1, y1)
        # IN1 + ReLU + dropout: planes only
        stats1 = torch.empty((N * C, 2), dtype=torch.float32, device=dev)
        p1 = _chan_planes_buffer(N, C, H, W, dev)
        # (the weight gradient's planes only when a backward pass can follow: inference / no_grad forward passes skip 78 MB per activation)
        train = any(ctx.needs_input_grad[:5])
        xp1 = _x_planes_buffer(N, C, H, W, dev) if train else None
        scale1 = torch.empty(N, dtype=torch.int32, device=dev)
        seed = off = 0
        if drop_p > 0.0:
            _dropout_state["offset"] = (_dropout_state["offset"] + 1) & 0xFFFFFFFF
            seed, off = _dropout_state["seed"], _dropout_state["offset"]
        with _span('in_fwd_planes_resblock'):
            L.instnorm_fwd_planes(_p(y1), None, None, None, _p(stats1), N, C, H, W, eps, ACT_RELU, 0.2, drop_p, seed, off, _p(p1), _p(scale1),
                                  None, _p(xp1), st)
        # conv2 reads the planes; the (never written) fp32 tensor they stand for is only a key: the planes' own address serves
        y2 = torch.empty_like(x)
        conv(_p(p1), p1, scale1, w2, b2, y2)
        del p1
        # IN2 + skip
        out = torch.empty_like(x)
        stats2 = torch.empty((N * C, 2), dtype=torch.float32, device=dev)
        words = _max_words(N, dev)
        if feeds_block:
            pout = _chan_planes_buffer(N, C, H, W, dev)
            xpout = _x_planes_buffer(N, C, H, W, dev) if train else None
            scale_out = torch.empty(N, dtype=torch.int32, device=dev)
            # (the next block's IN2 + skip producer is the one consumer of these maxima: it reduces the partial words itself)
            with _span('in_fwd_planes_resblock'), _lazy_max():
                L.instnorm_fwd_planes(_p(y2), _p(x), _p(_absmax_word(x, lazy_ok=True)), _p(out), _p(stats2), N, C, H, W, eps, ACT_NONE, 0.2, 0.0, 0, 0,
                                      _p(pout), _p(scale_out), _p(words), _p(xpout), st)
            out._nemar_planes = (pout, scale_out, out._version)
            if xpout is not None:
                out._nemar_xplanes = (xpout, scale_out, out._version)
        else:
            L.instnorm_fwd_max(_p(y2), _p(x), _p(out), _p(stats2), N * C, H * W, eps, ACT_NONE, 0.2, _p(words), C, st)
        _tag_max(out, words, lazy=feeds_block and _LAZY_MAX)
        ctx.save_for_backward(y1, stats1, y2, stats2, w1, w2)
        # conv1's weight gradient takes the producer's X planes of x where they exist, else x itself
        ctx.x0 = None if xp0 is not None else x
        ctx.xp0 = xp0[0] if xp0 is not None else None
        ctx.x_words, ctx.xp1, ctx.scale1 = x_words, xp1, scale1
        ctx.params = (w1, b1, w2, b2)
        ctx.drop = (drop_p, seed, off)
        for prm in (w1, b1, w2, b2):
            if prm is not None:
                _note_use(prm)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        y1, stats1, y2, stats2, w1v, w2v = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        g_out = _c(g_out)
        N, C, H, W = g_out.shape
        dev = g_out.device
        st = _stream()
        need_x = ctx.needs_input_grad[0]
        need = [ctx.needs_input_grad[1], ctx.needs_input_grad[2] and b1 is not None, ctx.needs_input_grad[3],
                ctx.needs_input_grad[4] and b2 is not None]
        drop_p, seed, off = ctx.drop
        gbytes = _gplanes_need.get((N, C, H, W, C, 3, 3, 1, 1, PAD_REFLECT))
        if gbytes is None:
            gbytes = _gplanes_need[(N, C, H, W, C, 3, 3, 1, 1, PAD_REFLECT)] = L.conv2d_gy_planes_bytes(N, C, H, W, C, 3, 3, 1, 1, PAD_REFLECT)
        arena = _conv_scratch(N, H, W, C, C, 3, 3, 1, 1, dev)
        dwsb = Q.conv2d_bwd_data_workspace(N, C, H, W, C, 3, 3, 1, 1, PAD_REFLECT)

        def norm_bwd(xin, stats, g, gwords, act, p, sd, of, want_d, want_g, want_bias):
            d = _chan_planes_buffer(N, C, H, W, dev) if want_d else None
            gp = torch.empty(int(gbytes), dtype=torch.uint8, device=dev) if want_g else None
            scale = torch.empty(N, dtype=torch.int32, device=dev)
            bsum = torch.empty((N, C), dtype=torch.float32, device=dev) if want_bias else None
            with _span('in_bwd_planes_resblock'):
                L.instnorm_bwd_planes(_p(xin), _p(stats), _p(g), _p(gwords), N, C, H, W, act, 0.2, p, sd, of, 1, None, _p(d), _p(gp), _p(scale),
                                      _p(bsum), st)
            return d, gp, scale, bsum

        def dgrad(d, scale, weight, dst, addend, out_words):
            ws, hit, plan = _packed(weight, ('dgrad', 1, 1, PAD_REFLECT, True, N, H, W), dwsb)
            with _record(plan), _span('dgrad_resblock'):
                L.conv2d_bwd_data_ex(_p(d), _p(weight), None, ACT_NONE, 0.0, _p(dst), C, None, 0, N, H, W, C, H, W, 3, 3, 1, 1, PAD_REFLECT,
                                     _p(ws), dwsb, hit, st, _extras(arena, scale, planes=d, addend=addend, out_max=out_words))

        def wgrad(x_t, xpl, x_words, gp, g_scale, weight, bias, bsum, want_w, want_b):
            """side stream: the weight gradient from planes on both operands; the bias gradient from the producer's per-plane sums"""
            if not (want_w or want_b):
                return
            gw = _grad_buffer(weight) if want_w else None
            gb = _grad_buffer(bias) if want_b else None
            if _side_on[0]:
                for buf in (gw, gb):
                    if buf is not None:
                        _side_touched.add(id(buf))
            else:
                for buf in (gw, gb):
                    if buf is not None:
                        _main_lane_grad(buf, dev)
            with _on_side(dev, x_t, xpl, x_words, gp, g_scale, bsum), _span('wgrad_resblock'):
                if want_w:
                    wsb = Q.conv2d_bwd_weight_workspace(N, C, H, W, C, H, W, 3, 3, 1, 1)
                    side_arena = _conv_scratch(N, H, W, C, C, 3, 3, 1, 1, dev)
                    key = x_t if x_t is not None else xpl           # (with planes the fp32 operand is only a key)
                    # (the bias gradient — the sum over the batch of the backward producer's per-plane sums — rides in the launch that sums the
                    # weight gradient's slabs: nemar_conv_extras.bias_partials)
                    L.conv2d_bwd_weight_ex(_p(key), C, None, 0, _p(gp), _p(gw), _p(gb) if want_b else None, N, H, W, C, H, W, 3, 3, 1, 1, PAD_REFLECT,
                                           _p(_workspace(wsb, dev)), wsb, _stream(),
                                           _extras(side_arena, x_words, g_scale, planes=xpl, src2_planes=gp, bias_partials=bsum if want_b else None))
                    grad_ready(weight)
                    if want_b:
                        grad_ready(bias)
                elif want_b:
                    L.bias_from_partials(_p(bsum), N, C, _p(gb), _stream())
                    grad_ready(bias)

        # IN2 backward (no activation, the skip path passes g_out on unchanged)
        gwords = _absmax_word(g_out, lazy_ok=True)
        d2, gp2, scale2, bsum2 = norm_bwd(y2, stats2, g_out, gwords, ACT_NONE, 0.0, 0, 0, True, need[2], need[3])
        gmid = torch.empty_like(g_out)
        mid_words = _max_words(N, dev)
        with _lazy_max():                      # (consumed by this block's own IN1 backward producer, below)
            dgrad(d2, scale2, w2v, gmid, None, mid_words)
        del d2
        wgrad(None, ctx.xp1, ctx.scale1, gp2, scale2, w2, b2, bsum2, need[2], need[3])
        # dropout + ReLU + IN1 backward
        d1, gp1, scale1b, bsum1 = norm_bwd(y1, stats1, gmid, mid_words[:N], ACT_RELU, drop_p, seed, off, need_x or not need[0], need[0], need[1])
        gin = None
        if need_x:
            gin = torch.empty_like(g_out)
            in_words = _max_words(N, dev)
            # the block's input came with producer planes = from another block's end: gin goes to THAT block's backward producer and to nothing else
            lazy_in = ctx.xp0 is not None and _LAZY_MAX
            with _lazy_max(lazy_in):
                dgrad(d1, scale1b, w1v, gin, g_out, in_words)      # + the skip gradient, in the epilogue
            _tag_max(gin, in_words, lazy=lazy_in)
        del d1
        wgrad(ctx.x0, ctx.xp0, ctx.x_words, gp1, scale1b, w1, b1, bsum1, need[0], need[1])
        return gin, None, None, None, None, None, None, None


def resnet_block(x, w1, b1, w2, b2, dropout_p=0.0, feeds_block=False, eps=1e-5):
    """One reflect-padded, instance-normalised ResnetBlock as a single autograd node where the wide-layer route takes it
    (None: the caller composes it from conv2d / instance_norm)."""
    if not _block_fusable(x, w1.shape[0]) or tuple(w1.shape) != (x.shape[1], x.shape[1], 3, 3) or tuple(w2.shape) != tuple(w1.shape):
        return None
    return _ResBlock.apply(x, w1, b1, w2, b2, float(dropout_p), bool(feeds_block), float(eps))


class _MaxPool2(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        L.maxpool2_fwd(_p(x), _p(y), N * C, H, W, _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = _c(gy)
        N, C, H, W = x.shape
        gx = torch.empty_like(x)
        L.maxpool2_bwd(_p(x), _p(gy), None, _p(gx), N * C, H, W, _stream())
        return gx


def max_pool2(x):
    return _MaxPool2.apply(x)


class _MaxPool2Skip(Function):
    """(MaxPool2d(2)(x), a second handle of x) — the U-Net encoder's pooled tensor and its skip connection (reference models/stn/layers.py
    :174-185).  The skip's gradient is the `addend` of the pooling's backward kernel: gx = g_skip + unpool(gy), one launch."""
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        L.maxpool2_fwd(_p(x), _p(y), N * C, H, W, _stream())
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gskip):
        (x,) = ctx.saved_tensors
        if gy is None:
            return gskip
        gy, gskip = _c(gy), _c(gskip)
        N, C, H, W = x.shape
        gx = torch.empty_like(x)
        L.maxpool2_bwd(_p(x), _p(gy), _p(gskip), _p(gx), N * C, H, W, _stream())
        return gx


def max_pool2_with_skip(x):
    if not (_own_nodes and torch.is_grad_enabled() and x.requires_grad):
        return _MaxPool2.apply(x), x
    y, skip = _MaxPool2Skip.apply(x)
    for name in _TAGS:
        v = getattr(x, name, None)
        if v is not None:
            setattr(skip, name, v)
    return y, skip


class _Bilinear(Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        x = _c(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
        L.bilinear_fwd(_p(x), _p(y), N * C, H, W, Ho, Wo, _stream())
        ctx.shape = (N, C, H, W, Ho, Wo)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        N, C, H, W, Ho, Wo = ctx.shape
        gy = _c(gy)
        gx = torch.empty((N, C, H, W), dtype=torch.float32, device=gy.device)
        L.bilinear_bwd(_p(gy), _p(gx), N * C, H, W, Ho, Wo, _stream())
        return gx, None, None


def resize_bilinear(x, Ho, Wo):
    """F.interpolate(x, (Ho, Wo), mode='bilinear', align_corners=False)."""
    if x.shape[2] == Ho and x.shape[3] == Wo:
        return x
    return _Bilinear.apply(x, int(Ho), int(Wo))


_dropout_state = {"seed": 0x5EED5EED, "offset": 0}

# ---- step parameters in device memory (hipGraph replay: nemar_amd/models/nemar_model.py enable_step_graph) -----------------------
# A captured graph replays the launches of ONE step with frozen arguments.  What changes from step to step — the dropout offsets
# and Adam's bias-corrected step size — then has to be read from device memory: `step_params(True)` switches the dropout launches to
# "offset within the step (frozen) + a device word the host rewrites before every step" and FlatAdam.step() to nemar_adam_step_dev.
# The same arithmetic as the by-value forms: an eager run in this mode and a graph replay are bit-identical.
_step_params = {"on": False, "base": None, "capturing": False, "step": 0}
DROPOUT_STEP_STRIDE = 4096           # offsets per step (dropout launches per step must stay below it)


def step_params(on, device=None):
    _step_params["on"] = bool(on)
    if on:
        if _step_params["base"] is None:
            _step_params["base"] = torch.zeros(1, dtype=torch.int32, device=device)
        L.set_dropout_base(_p(_step_params["base"]))
    else:
        L.set_dropout_base(None)


def begin_step():
    """Before every step in step_params mode (eager or graph replay, NOT inside a capture): the dropout offsets of this step are
    base + 1 .. base + calls, base = step index x DROPOUT_STEP_STRIDE in the device word."""
    if not _step_params["on"]:
        return
    if _dropout_state["offset"] >= DROPOUT_STEP_STRIDE:
        raise RuntimeError("more than %d dropout launches in one step" % DROPOUT_STEP_STRIDE)
    _step_params["step"] += 1
    base = (_step_params["step"] * DROPOUT_STEP_STRIDE) & 0x7FFFFFFF
    # (the value travels as a kernel argument: an asynchronous copy from a temporary, pageable host tensor may read its source AFTER the
    # tensor has been freed and its memory reused — seen as forward and backward masks of one step drawn from different bases)
    L.store_words(_p(_step_params["base"]), ctypes.byref(ctypes.c_uint32(base)), 1, _stream())
    _dropout_state["offset"] = 0


def manual_seed(seed):
    """Seed of the counter-based dropout generator.  NEMARModel seeds it from torch.initial_seed() + rank, so that
    torch.manual_seed() governs it and data-parallel ranks draw different masks (as torch's per-process RNG does in the
    reference)."""
    _dropout_state["seed"] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _dropout_state["offset"] = 0


def _dropout_launch(x, y, p, seed, off):
    n = x.shape[0] if x.dim() == 4 else 0
    if n and _wants_max(x) and (x.numel() // n) % 4 == 0:
        words = _max_words(n, x.device)
        L.dropout_max(_p(x), _p(y), n, x.numel() // n, p, seed, off, _p(words), _stream())
        _tag_max(y, words)
    else:
        L.dropout(_p(x), _p(y), x.numel(), p, seed, off, _stream())


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, p):
        x = _c(x)
        y = torch.empty_like(x)
        _dropout_state["offset"] = (_dropout_state["offset"] + 1) & 0xFFFFFFFF
        ctx.key = (p, _dropout_state["seed"], _dropout_state["offset"])
        _dropout_launch(x, y, p, ctx.key[1], ctx.key[2])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        gy = _c(gy)
        gx = torch.empty_like(gy)
        p, seed, off = ctx.key
        _dropout_launch(gy, gx, p, seed, off)
        return gx, None


def dropout(x, p=0.5, training=True):
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(x, float(p))


# ---- batch concatenation / batch slices / fan-out of a tensor, with the library's own kernels in BOTH directions ----------------------
# The batched passes (T on [real_A ; R(real_A)], D on [real ; fakes]) concatenate along the batch and slice the result; tensors with two
# consumers (a ResnetBlock's input, a U-Net skip, a generated image read by a loss and by D, the deformation field) have their gradients
# added.  Left to autograd these are ATen kernels — cat, zero-fill + copy + add_ per slice, add per fan-out: 49 launches per step of the
# bench configuration (tools/aten_on_path.py), the only arithmetic on the path that is not this library's, and built WITH the packed-FP32
# instruction forms that miscompute beside another kernel's MFMAs.  As autograd nodes of their own they run
# nemar_concat_pieces / nemar_add2 instead, and a slice no loss reads costs nothing (no zero-filled gradient is materialised).
_TAGS = ('_nemar_absmax', '_nemar_absmax_lazy', '_nemar_planes', '_nemar_xplanes', '_nemar_grad_from')
_own_nodes = os.environ.get("NEMAR_OWN_NODES", "1") != "0"       # 0: autograd's own cat / slices / accumulation (A/B of the schedule only)


def _zero(t):
    """t[...] = 0 for a contiguous fp32 tensor, by the library's kernel (a NULL piece of nemar_concat_pieces)"""
    if t.numel():
        _concat_launch([None], [t.numel()], t)
    return t


def _concat_launch(pieces, counts, dst):
    k = len(pieces)
    ptrs = (ctypes.c_void_p * k)(*[None if t is None else t.data_ptr() for t in pieces])
    cnts = (ctypes.c_longlong * k)(*counts)
    L.concat_pieces(ptrs, cnts, k, _p(dst), _stream())


class _CatBatch(Function):
    @staticmethod
    def forward(ctx, *ts):
        ts = [_c(t) for t in ts]
        ctx.sizes = [t.shape[0] for t in ts]
        out = torch.empty((sum(ctx.sizes),) + tuple(ts[0].shape[1:]), dtype=torch.float32, device=ts[0].device)
        _concat_launch(ts, [t.numel() for t in ts], out)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _c(g)
        out, o = [], 0
        for i, n in enumerate(ctx.sizes):
            out.append(g[o:o + n] if ctx.needs_input_grad[i] else None)       # (views: contiguous pieces of the batch)
            o += n
        return tuple(out)


def cat_batch(tensors):
    """torch.cat(tensors, 0) of contiguous fp32 tensors of one trailing shape"""
    tensors = list(tensors)
    if len(tensors) == 1:
        return tensors[0]
    if not _own_nodes:
        return torch.cat(tensors, 0)
    if len(tensors) > 8:
        return cat_batch([cat_batch(tensors[:8])] + tensors[8:])
    return _CatBatch.apply(*tensors)


class _SplitBatch(Function):
    @staticmethod
    def forward(ctx, x, k):
        n = x.shape[0] // k
        ctx.k, ctx.n, ctx.shape = k, n, tuple(x.shape)
        ctx.set_materialize_grads(False)
        return tuple(x[i * n:(i + 1) * n] for i in range(k))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return None, None
        gs = [_c(g) for g in gs]
        ref = next(g for g in gs if g is not None)
        out = torch.empty(ctx.shape, dtype=torch.float32, device=ref.device)
        per = ref.numel()
        _concat_launch(gs, [per] * ctx.k, out)
        return out, None


def split_batch(x, k):
    """(x[0:n], x[n:2n], ...) for k equal pieces of the batch; the backward pass assembles the pieces' gradients in one launch"""
    if k == 1:
        return (x,)
    if x.shape[0] % k or k > 8:
        raise ValueError("split_batch: %d pieces of a batch of %d" % (k, x.shape[0]))
    if not _own_nodes:
        n = x.shape[0] // k
        return tuple(x[i * n:(i + 1) * n] for i in range(k))
    return _SplitBatch.apply(_c(x), int(k))


class _Fork(Function):
    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        acc = None
        for g in gs:
            if g is None:
                continue
            g = _c(g)
            if acc is None:
                acc = g
            else:
                out = torch.empty_like(acc)
                L.add2(_p(acc), _p(g), _p(out), acc.numel(), _stream())
                acc = out
        return acc, None


def fork(x, n=2):
    """n handles of one tensor for n consumers: the consumers' gradients are added by nemar_add2 (in consumer order) instead of
    autograd's accumulation.  The handles keep the producer's side information (maximum words, operand planes)."""
    if not (_own_nodes and torch.is_grad_enabled() and x.requires_grad) or n < 2:
        return (x,) * n
    outs = _Fork.apply(x, int(n))
    for name in _TAGS:
        v = getattr(x, name, None)
        if v is not None:
            for o in outs:
                setattr(o, name, v)
    return outs


# ------------------------------------------------------------------------------------------------------
_ggs_alloc = torch.empty_like      # (a seam for tools/diag_lost_stores.py: where the grid gradient's buffer comes from)


class _Warp(Function):
    """grid_sample(img, grid(grid_src)) for every image in `imgs` with ONE shared grid source; the gradient w.r.t.
    the grid source is accumulated across the images inside the kernels."""

    @staticmethod
    def forward(ctx, grid_src, mode, Ho, Wo, *imgs):
        gs = _c(grid_src)
        imgs = [_c(i) for i in imgs]
        outs = []
        st = _stream()
        for img in imgs:
            N, C, H, W = img.shape
            out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=img.device)
            with _span('grid_sample_fwd'):
                L.grid_sample_fwd(_p(img), _p(gs), mode, _p(out), N, C, H, W, Ho, Wo, st)
            outs.append(out)
        ctx.save_for_backward(gs, *imgs)
        ctx.cfg = (mode, Ho, Wo)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        gs, *imgs = ctx.saved_tensors
        mode, Ho, Wo = ctx.cfg
        st = _stream()
        need_gs = ctx.needs_input_grad[0]
        ggs = _ggs_alloc(gs)
        first = True
        gimgs = []
        for k, (img, go) in enumerate(zip(imgs, gouts)):
            need_img = ctx.needs_input_grad[4 + k]
            if go is None or (not need_img and not need_gs):
                gimgs.append(None)
                continue
            go = _c(go)
            N, C, H, W = img.shape
            gin = torch.empty_like(img) if need_img else None
            wsb = Q.grid_sample_bwd_workspace(N, C, H, W)
            ws = _zeroed_workspace(wsb, (img.device, N, C, H, W))    # leading part: all-zero in, all-zero out; rest: scratch
            with _span('grid_sample_bwd_gin' if need_img else 'grid_sample_bwd_nogin'):
                L.grid_sample_bwd(_p(img), _p(gs), mode, _p(go), _p(gin), 0, _p(ggs), 0 if first else 1, N, C, H, W,
                                  Ho, Wo, _p(ws), wsb, st)
            first = False
            gimgs.append(gin)
        if first:
            ggs.zero_()
        return (ggs if need_gs else None, None, None, None, *gimgs)


def warp_unet(offsets, imgs):
    """UnetSTN sampling: grid = linspace identity + offsets [N,2,H,W]; returns the list of warped images."""
    Ho, Wo = offsets.shape[2:]
    return list(_Warp.apply(offsets, GRID_UNET, int(Ho), int(Wo), *imgs))


def warp_affine(dtheta, imgs):
    """AffineSTN sampling: theta = dtheta + I, F.affine_grid(align_corners=False) at each image's own size."""
    outs = []
    # one grid source, possibly different image sizes -> group by size (normally a single group)
    by_size = {}
    for i, img in enumerate(imgs):
        by_size.setdefault(tuple(img.shape[2:]), []).append(i)
    res = [None] * len(imgs)
    for (H, W), idxs in by_size.items():
        o = _Warp.apply(dtheta, GRID_AFFINE, int(H), int(W), *[imgs[i] for i in idxs])
        for i, t in zip(idxs, o):
            res[i] = t
    outs.extend(res)
    return outs


def grid_sample(img, grid):
    """F.grid_sample(img, grid, 'bilinear', 'zeros', align_corners=False) with an explicit [N,Ho,Wo,2] grid."""
    return _Warp.apply(grid, GRID_EXPLICIT, int(grid.shape[1]), int(grid.shape[2]), img)[0]


class _Smoothness(Function):
    @staticmethod
    def forward(ctx, d, img, alpha, factor):
        d, img = _c(d), _c(img)
        N, _, H, W = d.shape
        Ci = 0 if img is None else img.shape[1]
        loss = torch.empty((1,), dtype=torch.float32, device=d.device)
        wsb = Q.smoothness_workspace(N, H, W)
        ws = _workspace(wsb, d.device)
        L.smoothness_fwd(_p(d), _p(img), Ci, alpha, factor, _p(loss), 0, _p(ws), wsb, N, H, W, _stream())
        ctx.save_for_backward(d, img)
        ctx.cfg = (alpha, factor, Ci)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        d, img = ctx.saved_tensors
        alpha, factor, Ci = ctx.cfg
        N, _, H, W = d.shape
        gd = torch.empty_like(d)
        L.smoothness_bwd(_p(d), _p(img), Ci, alpha, _p(_c(g)), factor, _p(gd), 0, N, H, W, _stream())
        return gd, None, None, None


def smoothness(d, img=None, alpha=0.0, factor=1.0):
    """factor * smoothness_loss(d, img, alpha); img carries no gradient."""
    if img is not None:
        img = img.detach()
    return _Smoothness.apply(d, img, float(alpha), float(factor))


class _L1(Function):
    @staticmethod
    def forward(ctx, a, b, weight):
        a, b = _c(a), _c(b)
        loss = torch.empty((1,), dtype=torch.float32, device=a.device)
        wsb = Q.loss_workspace()
        ws = _workspace(wsb, a.device)
        L.l1_loss_fwd(_p(a), _p(b), a.numel(), weight, _p(loss), 0, _p(ws), wsb, _stream())
        ctx.save_for_backward(a, b)
        ctx.weight = weight
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        L.l1_loss_bwd(_p(a), _p(b), a.numel(), _p(_c(g)), ctx.weight, _p(ga), 0, _stream())
        return ga, None, None


def l1_loss(a, b=None, weight=1.0):
    """weight * mean|a - b| (b None: weight * mean|a|); b is treated as a constant."""
    if b is not None:
        b = b.detach()
    return _L1.apply(a, b, float(weight))


class _GanLoss(Function):
    @staticmethod
    def forward(ctx, x, mode, real, weight):
        x = _c(x)
        loss = torch.empty((1,), dtype=torch.float32, device=x.device)
        wsb = Q.loss_workspace()
        ws = _workspace(wsb, x.device)
        L.gan_loss_fwd(_p(x), x.numel(), mode, real, weight, _p(loss), 0, _p(ws), wsb, _stream())
        ctx.save_for_backward(x)
        ctx.cfg = (mode, real, weight)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        mode, real, weight = ctx.cfg
        gx = torch.empty_like(x)
        L.gan_loss_bwd(_p(x), x.numel(), mode, real, _p(_c(g)), weight, _p(gx), _stream())
        return gx, None, None, None


def gan_loss(x, target_is_real, mode="vanilla", weight=1.0):
    if mode not in GAN_MODES:
        raise NotImplementedError('gan mode %s not implemented' % mode)
    return _GanLoss.apply(x, GAN_MODES[mode], 1 if target_is_real else 0, float(weight))


# ------------------------------------------------------------------------------------------------------
class FlatAdam:
    """torch.optim.Adam(params, lr, betas) semantics over ONE flat fp32 buffer per optimizer: parameters and their
    gradients are re-seated as views of flat buffers, so a step is a single fused kernel launch, zero_grad is one
    memset, and the gradient buffer is a single all-reduce bucket for data parallelism."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: empty parameter list")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.param_groups = [{"params": self.params, "lr": self.lr, "betas": self.betas, "eps": self.eps}]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        # 16-byte align every parameter inside the flat buffers (float4 kernels)
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.numel, self.flat_numel = n, total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_count = 0
        self._epoch = PackEpoch()
        _plan_ids[0] += 1
        self._plan, self._plan_buf = _plan_ids[0], None
        self._epoch.plan = self._plan
        _plan_owners.add(self)
        invalidate_packed_weights()
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                p._pack_epoch = self._epoch
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
                p._flat_grad = p.grad
        self.offsets = offs

    def zero_grad(self, set_to_none=False):
        join_side()                    # (a weight-gradient branch of an earlier pass must not land after the fill)
        if self.flat_g.is_cuda:
            _zero(self.flat_g)
        else:
            self.flat_g.zero_()        # (host-resident buffers: the gloo tests of the gradient all-reduce; no kernel runs on them)

    def step(self):
        join_side()                    # every gradient contribution issued on the side stream is behind the compute stream from here
        self._epoch.n += 1
        self.step_count += 1
        g = self.param_groups[0]
        if _step_params["on"]:
            # the two step-dependent scalars in device memory (written by prepare_step() before a graph replay; here when eager)
            if not _step_params["capturing"]:
                self.prepare_step(self.step_count)
            L.adam_step_dev(_p(self.flat_p), _p(self.flat_g), _p(self.m), _p(self.v), self.flat_numel, _p(self.hyper),
                            self.betas[0], self.betas[1], self.eps, _stream())
            self._repack()
            return
        L.adam_step(_p(self.flat_p), _p(self.flat_g), _p(self.m), _p(self.v), self.flat_numel, float(g["lr"]),
                    self.betas[0], self.betas[1], self.eps, self.step_count, _stream())
        self._repack()

    def _repack(self):
        """Right behind the Adam kernel: every recorded pack job of this optimizer's weights, in <= 5 launches (weight-pack plans)."""
        if not _plans_on[0] or L.pack_plan_jobs(self._plan) == 0:
            return
        if L.pack_plan_dirty(self._plan):
            if _step_params["capturing"]:
                raise RuntimeError("a weight was packed for the first time inside a graph capture: run the step eagerly first")
            need = int(L.pack_plan_bytes(self._plan))
            if self._plan_buf is None or self._plan_buf.numel() < need:
                _retire(self._plan_buf)
                self._plan_buf = torch.empty(need + 4096, dtype=torch.uint8, device=self.flat_p.device)
            L.pack_plan_commit(self._plan, _p(self._plan_buf), self._plan_buf.numel(), _stream())
        L.pack_plan_run(self._plan, _stream())

    def prepare_step(self, step):
        """hyper <- (lr / (1 - beta1^step), sqrt(1 - beta2^step)) for the step about to run: nemar_adam_step's own arithmetic (double,
        rounded to float once)."""
        if getattr(self, 'hyper', None) is None:
            self.hyper = torch.zeros(2, dtype=torch.float32, device=self.flat_p.device)
        lr = float(self.param_groups[0]["lr"])
        bc1 = 1.0 - self.betas[0] ** step
        bc2 = 1.0 - self.betas[1] ** step
        words = (ctypes.c_float * 2)(lr / bc1, bc2 ** 0.5)            # (double -> float, rounded once; as kernel arguments: see begin_step)
        L.store_words(_p(self.hyper), ctypes.byref(words), 2, _stream())

    def replayed_step(self):
        """book-keeping of a step that ran inside a graph replay (the launches were the graph's)"""
        self._epoch.n += 1
        self.step_count += 1

    def state_dict(self):
        return {"step": self.step_count, "m": self.m.clone(), "v": self.v.clone()}

    def load_state_dict(self, sd):
        invalidate_packed_weights()
        self.step_count = int(sd["step"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
