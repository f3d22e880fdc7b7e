"""InstanceNorm / dropout with and without the per-sample max word (the fp16 x 3 route's scale source), at the bench's hot shape."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from nemar_amd import _lib
L = _lib.load()
dev = torch.device('cuda:0')
p = lambda t: t.data_ptr() if t is not None else None
st = torch.cuda.current_stream().cuda_stream

def timeit(fn, it=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it

for (N, C, H, W) in ((8, 256, 64, 64), (16, 256, 64, 64), (8, 128, 128, 128)):
    x = torch.randn(N, C, H, W, device=dev); y = torch.empty_like(x); g = torch.randn_like(x)
    stats = torch.empty(N * C, 2, device=dev); words = torch.zeros(N, dtype=torch.int32, device=dev)
    mb = x.numel() * 4 / 1e6
    t0 = timeit(lambda: L.instnorm_fwd(p(x), None, p(y), p(stats), N * C, H * W, 1e-5, 1, 0.0, st))
    t1 = timeit(lambda: L.instnorm_fwd_max(p(x), None, p(y), p(stats), N * C, H * W, 1e-5, 1, 0.0, p(words), C, st))
    t2 = timeit(lambda: L.instnorm_bwd(p(x), p(stats), p(g), p(y), N * C, H * W, 1, 0.0, st))
    t3 = timeit(lambda: L.instnorm_bwd_max(p(x), p(stats), p(g), p(y), N * C, H * W, 1, 0.0, p(words), C, st))
    t4 = timeit(lambda: L.dropout(p(x), p(y), x.numel(), 0.5, 1234, 7, st))
    t5 = timeit(lambda: L.dropout_max(p(x), p(y), N, x.numel() // N, 0.5, 1234, 7, p(words), st))
    print('%s  %.1f MB | IN fwd %.1f / max %.1f us | IN bwd %.1f / max %.1f us | dropout %.1f / max %.1f us  (%.2f TB/s plain fwd)'
          % ((N, C, H, W), mb, t0, t1, t2, t3, t4, t5, 2 * mb / t0 / 1e6 * 1e6 / 1e6))
