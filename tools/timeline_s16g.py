"""Cycle stamps of one workgroup of the general 16-bit-pipe forward kernel (conv_s16g.hip), per chunk and wave:
top -> [max + exchange barrier] -> [weights DMA issue + rescale + conversion] -> [DMA wait] -> [next loads issue + barrier] -> [tap loop].
Needs tools/build_timeline_lib.py.  usage: timeline_s16g.py C K H R stride [N]"""
import ctypes, os, sys
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import _lib
lib = _lib.load(os.environ.get('NEMAR_TL_LIB')); dev = torch.device('cuda:0')
C, K, H, R, s = (int(a) for a in sys.argv[1:6]); N = int(sys.argv[6]) if len(sys.argv) > 6 else 8
p = 1 if R > 1 else 0
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
OH = (H + 2 * p - R) // s + 1
x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, R, R, device=dev) * 0.05
y = torch.empty(N, K, OH, OH, device=dev)
wsb = lib.conv2d_fwd_workspace(N, H, H, K, C, R, R, s, p); ws = torch.empty(wsb // 4 + 16, device=dev)
tl = torch.zeros(4 * 8 * 8, dtype=torch.int64, device=dev)
call = lambda pre: lib.conv2d_fwd(P(x), C, None, 0, P(w), None, P(y), N, H, H, K, R, R, s, p, 0, 0, 0.2, P(ws), wsb, pre, st())
call(0)
for _ in range(5): call(1)
lib.tune_ptr(P(tl)); call(1); torch.cuda.synchronize(); lib.tune_ptr(None)
assert lib.last_route() == 3
t = tl.cpu().view(4, 8, 8)
names = ["max+bar1", "dma+conv", "dma wait", "loads+bar2", "taps", "(to next top)"]
for wv in range(4):
    for ch in range(min(8, (C + 15) // 16)):
        r = t[wv, ch]
        d = [int(r[i + 1] - r[i]) for i in range(5)]
        nxt = int(t[wv, ch + 1, 0] - r[5]) if ch + 1 < min(8, (C + 15) // 16) else 0
        print("wave %d chunk %d: " % (wv, ch) + "  ".join("%s %5d" % (n, v) for n, v in zip(names, d + [nxt])))
