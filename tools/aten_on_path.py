"""Which ATen / runtime kernels still run inside optimize_parameters() of the bench configuration, and from where?  One profiled step
(torch.profiler, with_stack): per ATen op that launched a device kernel, its count and the innermost nemar_amd / autograd frame.
   python tools/aten_on_path.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
import bench  # noqa: E402
from nemar_amd.models import create_model  # noqa: E402

dev = torch.device('cuda:0')
opt = bench.build_opt(8, 256)
model = create_model(opt); model.setup(opt)
g = torch.Generator(device=dev).manual_seed(0)
data = {'A': torch.rand(8, 3, 256, 256, device=dev, generator=g) * 2 - 1, 'B': torch.rand(8, 3, 256, 256, device=dev, generator=g) * 2 - 1,
        'A_paths': [''], 'B_paths': ['']}
for _ in range(4):
    model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    model.set_input(data); model.optimize_parameters()
    torch.cuda.synchronize()
rows = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 or not ev.kernels:
        continue
    where = 'autograd engine (gradient accumulation)'
    for fr in ev.stack:
        if 'nemar_amd' in fr or 'bench.py' in fr:
            where = fr.strip()
            break
    shapes = str(ev.input_shapes)[:60]
    rows[(ev.name, where, shapes)] += 1
tot = 0
for (name, where, shapes), n in sorted(rows.items(), key=lambda kv: (-kv[1], kv[0])):
    tot += n
    print('%3d x %-16s %-62s %s' % (n, name, shapes, where[-110:]))
print('%d ATen ops with device kernels in one step' % tot)
# every device kernel of the profiled step (the rocprofv3 per-step figures of profiles/ divide a whole process — model set-up included:
# ~360 one-time init / parameter-copy kernels — by its 11 steps)
from torch.autograd import DeviceType  # noqa: E402
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type == DeviceType.CUDA:
        kern[ev.name.split('(')[0][-60:]] += 1
print('%d device kernels / copies in the step; by name:' % sum(kern.values()))
for name, n in kern.most_common(12):
    print('   %4d  %s' % (n, name))
