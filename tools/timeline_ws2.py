"""Per-stage s_memtime timeline of the wave-specialised igemm's MFMA waves (workgroup (0,0), stages 40..43) on the
T resblock conv.  Needs a build with -DNEMAR_TIMELINE (the probe is compiled out by default: its s_memtime waits perturb the
loop).  usage: timeline_ws2.py [key=value ...]"""
import ctypes, os, sys
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import _lib
lib = _lib.load(os.environ.get('NEMAR_TL_LIB')); dev = torch.device('cuda:0')
for kv in sys.argv[1:]:
    k, v = kv.split('='); lib.tune(int(k), int(v))
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
N, C, K, H, R, s, p, pm = 8, 256, 256, 64, 3, 1, 1, 1
x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, R, R, device=dev) * 0.05; b = torch.randn(K, device=dev)
y = torch.empty(N, K, H, H, device=dev)
wsb = lib.conv2d_fwd_workspace(N, H, H, K, C, R, R, 1, 1); ws = torch.empty(wsb // 4 + 16, device=dev)
tl = torch.zeros(8 * 24, dtype=torch.int64, device=dev)
call = lambda pre: lib.conv2d_fwd(P(x), C, None, 0, P(w), P(b), P(y), N, H, H, K, R, R, s, p, pm, 1, 0.2, P(ws), wsb, pre, st())
call(0)
for _ in range(5): call(1)
lib.tune_ptr(P(tl)); call(1); torch.cuda.synchronize(); lib.tune_ptr(None)
allt = tl.cpu().view(8, 4, 6)
t = allt[:4]
names = ["reads(g1) issue", "MFMA blk1 issue", "lgkmcnt(0)", "barrier", "reads(g0')+MFMA blk2"]
for wv in range(4):
    print("wave %d:" % wv)
    for sidx in range(4):
        r = t[wv, sidx]
        d = [int(r[i + 1] - r[i]) for i in range(5)]
        nxt = int(t[wv, sidx + 1, 0] - r[5]) if sidx < 3 else 0
        print("  stage %d: %s | total %d | to next top %d" % (40 + sidx, "  ".join("%s %5d" % (n, v) for n, v in zip(names, d)), int(r[5] - r[0]), nxt))

base = int(t[0, 0, 0])
for ld in range(2):
    r = allt[4 + ld]
    print("loader %d:" % ld)
    for sidx in range(4):
        a = [int(r[sidx, j]) for j in range(4)]
        nxt = int(r[sidx + 1, 0]) - a[3] if sidx < 3 else 0
        print("  stage %d: wait-for-landing %5d  barrier %5d  issue %5d | to next %d | top at %+d vs MFMA wave 0 top" % (
            40 + sidx, a[1] - a[0], a[2] - a[1], a[3] - a[2], nxt, a[0] - int(t[0, sidx, 0])))
