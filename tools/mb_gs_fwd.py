"""grid_sample forward: 1 pixel per lane (default) vs 4 pixels per lane (nemar_grid_sample_tune(64))"""
import ctypes, os, sys
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import _lib
from tools.microbench import timeit
lib = _lib.load(); dev = torch.device('cuda:0')
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (N, C, H, W) in ((8, 3, 256, 256), (16, 3, 256, 256), (8, 3, 1024, 1024)):
    img = torch.rand(N, C, H, W, device=dev) * 2 - 1; res = torch.empty_like(img)
    for name, sigma in (('identity', 0.0), ('1px', 2.0 / W), ('smooth3px', None)):
        off = (torch.nn.functional.interpolate(torch.randn(N, 2, H // 32, W // 32, device=dev), size=(H, W), mode='bilinear') * (6.0 / W)
               if sigma is None else torch.randn(N, 2, H, W, device=dev) * sigma)
        row = []
        for v in (0, 64):
            lib.grid_sample_tune(v)
            t = timeit(lambda: lib.grid_sample_fwd(P(img), P(off), 1, P(res), N, C, H, W, H, W, st()), 50)
            row.append('%s %7.1f us %5.0f GB/s' % ('vec4' if v else 'vec1', t * 1e6, N * H * W * 32 / t / 1e9))
        lib.grid_sample_tune(0)
        print('%-20s %-10s %s' % ((N, C, H, W), name, ' | '.join(row)))
