"""Timeline of the data-parallel gradient exchange of ONE training step: for every bucket of the three optimizers, when it was launched
relative to the last weight-gradient kernel of its phase and how long its all-reduce took — i.e. how much of the collective hides behind
backward.  Runs with any world size: `python tools/overlap_timeline.py` on one GPU exercises the machinery with a one-rank RCCL
communicator (NEMAR_DIST_SINGLE=1: every call of the N > 1 path, zero bytes over xGMI); under
`python -m torch.distributed.run --nproc-per-node N tools/overlap_timeline.py` it prints the real thing, with achieved GB/s per bucket
(ring all-reduce moves 2 (N - 1) / N of the bucket per rank)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if int(os.environ.get("WORLD_SIZE", "1")) == 1:
    os.environ.setdefault("NEMAR_DIST_SINGLE", "1")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch
import bench
from nemar_amd import distributed as dist
from nemar_amd.models import create_model

rank, world, local = dist.init_from_env()
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
opt = bench.build_opt(8, 256)
opt.gpu_ids = [local]
torch.manual_seed(0)
model = create_model(opt); model.setup(opt)
g = torch.Generator(device=dev).manual_seed(1234 + rank)
data = {'A': torch.rand(8, 3, 256, 256, device=dev, generator=g) * 2 - 1, 'B': torch.rand(8, 3, 256, 256, device=dev, generator=g) * 2 - 1,
        'A_paths': [''], 'B_paths': ['']}
for _ in range(3):
    model.set_input(data); model.optimize_parameters()
# instrument: events at bucket launch (compute stream), at all-reduce begin / end (side stream), at finish() (compute stream)
rows = []
for name in ('D', 'R', 'T'):
    s = getattr(model, 'sync_' + name)
    orig_launch, orig_finish = s._launch, s.finish

    def launch(b, s=s, name=name, orig=orig_launch):
        e_ready = torch.cuda.Event(enable_timing=True); e_ready.record()
        orig(b)
        side = s._side
        e_done = torch.cuda.Event(enable_timing=True)
        if side is not None:
            with torch.cuda.stream(side):
                e_done.record()
        else:
            e_done.record()
        lo, hi = s.buckets[b]
        rows.append([name, b, (hi - lo) * 4, e_ready, e_done])

    def finish(s=s, name=name, orig=orig_finish):
        e = torch.cuda.Event(enable_timing=True); e.record()
        orig()
        e2 = torch.cuda.Event(enable_timing=True); e2.record()
        rows.append([name, 'finish', 0, e, e2])
    s._launch, s.finish = launch, finish
t0 = torch.cuda.Event(enable_timing=True); t0.record()
model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
if rank == 0:
    print('world %d; times in ms from the start of the step' % world)
    for name, b, nbytes, e0, e1 in rows:
        a, z = t0.elapsed_time(e0), t0.elapsed_time(e1)
        if b == 'finish':
            print('  %s finish(): compute stream reached it at %7.2f, waited until %7.2f (exposed %.2f ms)' % (name, a, z, z - a))
        else:
            gbs = (nbytes * 2.0 * (world - 1) / world) / max(z - a, 1e-6) / 1e6 if world > 1 else 0.0
            print('  %s bucket %d  %6.2f MB  ready %7.2f  reduced %7.2f  (%.2f ms%s)' %
                  (name, b, nbytes / 1e6, a, z, z - a, ', %.0f GB/s per rank on the ring' % gbs if world > 1 else ''))
if world > 1 or os.environ.get("NEMAR_DIST_SINGLE") == "1":
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
