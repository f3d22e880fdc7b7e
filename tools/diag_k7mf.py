"""Does the 7x7 many-to-few data gradient (T's stem) depend on what its workspace / the LDS held before the call?  (diagnostic)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import _lib
lib = _lib.load(); dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
N, C, K, H, W = 8, 3, 64, 256, 256
torch.manual_seed(0)
gy = torch.randn(N, K, H, W, device=dev); w = torch.randn(K, C, 7, 7, device=dev) * 0.05
wsb = lib.conv2d_bwd_data_workspace(N, C, H, W, K, 7, 7, 1, 3, 1)
outs = []
for fill in (0.0, float('nan'), 1e30, 0.0):
    ws = torch.full((wsb // 4 + 64,), fill, device=dev)
    gx = torch.full((N, C, H, W), float('nan'), device=dev)
    # (some LDS garbage: run an unrelated big-LDS kernel first)
    lib.conv2d_bwd_data(P(gy), P(w), None, 0, 0.0, P(gx), C, None, 0, N, H, W, K, H, W, 7, 7, 1, 3, 1, P(ws), wsb, 0, S())
    torch.cuda.synchronize()
    print('fill %-8s route %d  finite %s' % (fill, lib.last_route(), bool(torch.isfinite(gx).all())))
    outs.append(gx.clone())
for i in range(1, len(outs)):
    d = (outs[i] != outs[0])
    print('vs fill 0: differing %d  (nan-aware)' % int((d & ~(torch.isnan(outs[i]) & torch.isnan(outs[0]))).sum()))
