"""How long does the compute stream wait at each join of the side stream (the exposed tail of the weight-gradient branch)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nemar_amd import ops
from nemar_amd.models import create_model
dev = torch.device('cuda:0')
opt = bench.build_opt(8, 256)
model = create_model(opt); model.setup(opt)
g = torch.Generator(device=dev).manual_seed(0)
data = {'A': torch.rand(8, 3, 256, 256, device=dev, generator=g) * 2 - 1, 'B': torch.rand(8, 3, 256, 256, device=dev, generator=g) * 2 - 1,
        'A_paths': [''], 'B_paths': ['']}
for _ in range(5):
    model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
rows = []
orig = ops.join_side


def join():
    if ops._side_busy[0]:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); orig(); e1.record()
        rows.append((e0, e1))
    else:
        orig()


ops.join_side = join
# (the end-of-backward callback captured the original function object: patch the queueing too)
import torch.autograd
steps = 6
s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
ops_on_side_enter = ops._on_side.__enter__


def enter(self):
    r = ops_on_side_enter(self)
    return r


s0.record()
for _ in range(steps):
    model.set_input(data); model.optimize_parameters()
s1.record(); torch.cuda.synchronize()
print('step %.2f ms' % (s0.elapsed_time(s1) / steps))
per = len(rows) // steps if rows else 0
for i, (a, b) in enumerate(rows[-per:] if per else []):
    print('join %d of the step: compute stream waited %.3f ms' % (i, a.elapsed_time(b)))
