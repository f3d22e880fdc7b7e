"""instnorm_fwd_planes (norm_planes.hip) stand-alone at the residual blocks' shape: which of its parts cost what."""
import sys, os
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from nemar_amd import _lib
L = _lib.load()
dev = torch.device('cuda:0')
p = lambda t: t.data_ptr() if t is not None else None
st = torch.cuda.current_stream().cuda_stream

def timeit(fn, it=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it

C, H, W = 256, 64, 64
for N in (8, 16):
    # rotate over several buffers so that nothing stays in the 256 MB infinity cache between launches
    K = 6
    xs = [torch.randn(N, C, H, W, device=dev) for _ in range(K)]
    rs = [torch.randn(N, C, H, W, device=dev) for _ in range(K)]
    ys = [torch.empty(N, C, H, W, device=dev) for _ in range(K)]
    pls = [torch.empty(2 * N * (C // 8) * (H + 4) * (W + 4) * 16, dtype=torch.uint8, device=dev) for _ in range(K)]
    stats = torch.empty(N * C, 2, device=dev); sw = torch.zeros(N, dtype=torch.int32, device=dev)
    mw = torch.zeros(N * 2049, dtype=torch.int32, device=dev)
    xbytes = L.conv2d_x_planes_bytes(N, C, H, W, 3)
    xws = [torch.empty(xbytes, dtype=torch.uint8, device=dev) for _ in range(K)]
    rmax = torch.full((N,), 5.0, device=dev)
    i = [0]
    def run(res, y, drop, maxw, xw=0):
        k = i[0] = (i[0] + 1) % K
        L.instnorm_fwd_planes(p(xs[k]), p(rs[k]) if res else None, p(rmax) if res else None, p(ys[k]) if y else None, p(stats), N, C, H, W,
                              1e-5, 1, 0.0, 0.5 if drop else 0.0, 1234, 7, p(pls[k]), p(sw), p(mw) if maxw else None, p(xws[k]) if xw else None, st)
    def plain(res):
        k = i[0] = (i[0] + 1) % K
        L.instnorm_fwd(p(xs[k]), p(rs[k]) if res else None, p(ys[k]), p(stats), N * C, H * W, 1e-5, 1, 0.0, st)
    mb = xs[0].numel() * 4 / 1e6
    print('N=%d (%.0f MB per tensor)' % (N, mb))
    print('  plain instnorm_fwd            %6.1f us   with residual %6.1f us' % (timeit(lambda: plain(False)), timeit(lambda: plain(True))))
    for bits in (1, 2, 4, 8, 5, 13):
        L.tune(31, bits)
        print('  ablation bits %2d (1 no plane stores, 2 no transpose, 4 no statistics, 8 no fp32 stores): y+max %6.1f us' % (bits, timeit(lambda: run(0, 1, 0, 1))))
    L.tune(31, 0)
    for res, y, drop, maxw, xw in ((0, 1, 0, 1, 0), (0, 1, 1, 1, 0), (1, 1, 0, 1, 0), (0, 0, 0, 0, 0), (0, 0, 0, 1, 0), (0, 1, 0, 0, 0),
                                   (0, 0, 1, 0, 1), (1, 1, 0, 1, 1), (0, 0, 0, 0, 1)):
        print('  planes: residual %d  fp32 y %d  dropout %d  max words %d  X planes %d   %6.1f us' % (res, y, drop, maxw, xw, timeit(lambda: run(res, y, drop, maxw, xw))))
    # the step's two forms under the ablation bits: mid-block (dropout, planes + X planes only) and block end (residual, fp32 y, planes + X planes)
    for bits in (0, 1, 2, 4, 8):
        L.tune(31, bits)
        print('  ablation bits %2d: mid-block %6.1f us   block end %6.1f us' % (bits, timeit(lambda: run(0, 0, 1, 0, 1)), timeit(lambda: run(1, 1, 0, 1, 1))))
    L.tune(31, 0)
