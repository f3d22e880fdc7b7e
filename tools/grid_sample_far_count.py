import ctypes, sys, os
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)
sys.path.insert(0, '/root/repo')
import torch
from nemar_amd import _lib
lib = _lib.load(); dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
N, C, H, W = 8, 3, 1024, 1024
torch.manual_seed(0)
img = torch.rand(N, C, H, W, device=dev) * 2 - 1; go = torch.randn(N, C, H, W, device=dev); gin = torch.empty_like(img)
wsb = lib.grid_sample_bwd_workspace(N, C, H, W); gws = torch.zeros(wsb // 4 + 16, device=dev)
tiles = (W // 64) * (H // 16)
acc_b = 8 * N * C * H * W; dirty_b = 4 * N * tiles
zero = (acc_b + dirty_b + 15) // 16 * 16
wgc_off = zero + 16
for name, off in (('identity', torch.zeros(N, 2, H, W, device=dev)),
                  ('smooth3px', torch.nn.functional.interpolate(torch.randn(N, 2, H // 32, W // 32, device=dev), size=(H, W), mode='bilinear', align_corners=False) * (6.0 / W)),
                  ('smooth3px/64', torch.nn.functional.interpolate(torch.randn(N, 2, H // 64, W // 64, device=dev), size=(H, W), mode='bilinear', align_corners=False) * (6.0 / W)),
                  ('smooth10px/128', torch.nn.functional.interpolate(torch.randn(N, 2, H // 128, W // 128, device=dev), size=(H, W), mode='bilinear', align_corners=False) * (20.0 / W))):
    gd = torch.empty_like(off)
    for variant in (0, 32):
        lib.grid_sample_tune(variant)
        for _ in range(3):
            lib.grid_sample_bwd(P(img), P(off), 1, P(go), P(gin), 0, P(gd), 0, N, C, H, W, H, W, P(gws), wsb, st())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            lib.grid_sample_bwd(P(img), P(off), 1, P(go), P(gin), 0, P(gd), 0, N, C, H, W, H, W, P(gws), wsb, st())
        b.record(); torch.cuda.synchronize()
        cnt = gws.view(torch.int32)[wgc_off // 4: wgc_off // 4 + N * tiles].sum().item()
        print('%-16s %s: %7.1f us, far list entries %9d of %d pixels (%.1f %%)' % (name, 'follow ' if variant == 0 else 'centred', a.elapsed_time(b) * 100, cnt, N * H * W, 100.0 * cnt / (N * H * W)))
    lib.grid_sample_tune(0)
