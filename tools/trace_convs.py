"""List every conv-family C-ABI call of one optimize_parameters() step (BASELINE config 2) with its shape and count.
Writes JSON lines: {"op", "C0", "C1", "K", "R", "stride", "pad", "pad_mode", "H", "W", "N", "count"}."""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nemar_amd import ops
from nemar_amd.models import create_model

calls = collections.Counter()
L = ops.L
_f, _d, _w = L.conv2d_fwd_ex, L.conv2d_bwd_data_ex, L.conv2d_bwd_weight_ex          # the entry points nemar_amd/ops.py calls


def fwd(x0, C0, x1, C1, w, b, y, N, H, W, K, R, S, stride, pad, pm, *rest):
    calls[("fwd", C0, C1, K, R, stride, pad, pm, H, W, N, rest[0])] += 1
    return _f(x0, C0, x1, C1, w, b, y, N, H, W, K, R, S, stride, pad, pm, *rest)


def dgrad(gy, w, b, act, slope, gx0, C0, gx1, C1, N, H, W, K, OH, OW, R, S, stride, pad, pm, *rest):
    calls[("dgrad", C0, C1, K, R, stride, pad, pm, H, W, N, act)] += 1
    return _d(gy, w, b, act, slope, gx0, C0, gx1, C1, N, H, W, K, OH, OW, R, S, stride, pad, pm, *rest)


def wgrad(x0, C0, x1, C1, gy, gw, gb, N, H, W, K, OH, OW, R, S, stride, pad, pm, *rest):
    calls[("wgrad", C0, C1, K, R, stride, pad, pm, H, W, N, 0)] += 1
    return _w(x0, C0, x1, C1, gy, gw, gb, N, H, W, K, OH, OW, R, S, stride, pad, pm, *rest)


L.conv2d_fwd_ex, L.conv2d_bwd_data_ex, L.conv2d_bwd_weight_ex = fwd, dgrad, wgrad
opt = bench.build_opt(8, 256)
model = create_model(opt); model.setup(opt)
g = torch.Generator(device='cuda').manual_seed(0)
data = {'A': torch.rand(8, 3, 256, 256, device='cuda', generator=g) * 2 - 1,
        'B': torch.rand(8, 3, 256, 256, device='cuda', generator=g) * 2 - 1, 'A_paths': [''], 'B_paths': ['']}
model.set_input(data); model.optimize_parameters()          # warm (allocations)
calls.clear()
model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
for k, c in sorted(calls.items(), key=lambda kv: (kv[0][0], -kv[0][8], kv[0][1])):
    print(json.dumps(dict(zip(("op", "C0", "C1", "K", "R", "stride", "pad", "pad_mode", "H", "W", "N", "act"), k), count=c)))
