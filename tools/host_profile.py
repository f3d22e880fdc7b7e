"""cProfile of the host side of the eager step (where do the ~21 ms of issue time per step go?).  usage: python tools/host_profile.py [steps]"""
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nemar_amd.models import create_model
dev = torch.device('cuda:0')
opt = bench.build_opt(8, 256)
model = create_model(opt); model.setup(opt)
g = torch.Generator(device=dev).manual_seed(0)
data = {'A': torch.rand(8, 3, 256, 256, device=dev, generator=g) * 2 - 1, 'B': torch.rand(8, 3, 256, 256, device=dev, generator=g) * 2 - 1, 'A_paths': [''], 'B_paths': ['']}
for _ in range(4):
    model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    model.set_input(data); model.optimize_parameters()
pr.disable(); torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28); print(s.getvalue()[:6000])
