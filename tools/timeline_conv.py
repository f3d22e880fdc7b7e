"""Per-stage cycle timeline of workgroup 0 of the resblock conv (s_memtime stamps): where do the cycles go?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import _lib
lib = _lib.load(); dev = torch.device('cuda:0')
for k, v in zip(sys.argv[1::2], sys.argv[2::2]):
    lib.tune(int(k), int(v))
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
N, C, K, H = 8, 256, 256, 64
x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.05; b = torch.randn(K, device=dev)
y = torch.empty(N, K, H, H, device=dev)
wsb = lib.conv2d_fwd_workspace(K, C, 3, 3); ws = torch.empty(wsb // 4 + 16, device=dev)
tl = torch.zeros(8 * 64 * 8, dtype=torch.int64, device=dev)
for it in range(3):
    lib.tune_ptr(P(tl) if it == 2 else None)
    lib.conv2d_fwd(P(x), C, None, 0, P(w), P(b), P(y), N, H, H, K, 3, 3, 1, 1, 1, 1, 0.2, P(ws), wsb, 0, st())
torch.cuda.synchronize()
lib.tune_ptr(None)
t = tl.cpu()[:8 * 20].view(8, 4, 5)
base = int(t[:, 0, 0].min())
for wv in range(8):
    print('wave', wv)
    for st_ in range(4):
        r = t[wv, st_]
        nxt = t[wv, st_ + 1, 0] if st_ < 3 else None
        print('  stage %d: start@%6d issue %5d  lds+wait %5d  mfma-issue %5d  vmwait %5d %s' % (
            40 + st_, r[0] - base, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3],
            ('barrier+loop %5d | total %5d' % (nxt - r[4], nxt - r[0])) if nxt is not None else ''))
