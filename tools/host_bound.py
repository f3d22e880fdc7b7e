"""Is the eager step host-bound?  Host time to ISSUE a step (no synchronisation) against the GPU time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
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
n = 12
t0 = time.perf_counter(); marks = []
for _ in range(n):
    model.set_input(data); model.optimize_parameters()
    marks.append(time.perf_counter())
t_issue = marks[-1] - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('host issue time %.2f ms/step   wall (GPU done) %.2f ms/step   queue depth at the end: %.1f steps' % (
    t_issue / n * 1e3, t_all / n * 1e3, (t_all - t_issue) / (t_all / n)))
