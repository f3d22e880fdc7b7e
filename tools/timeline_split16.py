"""Cycle stamps of workgroup 0 of the pipelined split-16 convolution (taps 40..47): loader waves (wait-for-landing | barrier |
issue) and MFMA waves (MFMA blocks | lgkmcnt(0) | barrier).  Needs tools/build_timeline_lib.py.  usage: timeline_split16.py [k=v ...]"""
import ctypes, os, sys
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import _lib
from tests.side_inputs import SideInputs
lib = SideInputs(_lib.load(os.environ.get('NEMAR_TL_LIB'))); dev = torch.device('cuda:0')
VARIANT = int(os.environ.get('SPLIT16_VARIANT', '4'))
lib.tune(21, VARIANT)
for kv in sys.argv[1:]:
    k, v = kv.split('='); lib.tune(int(k), int(v))
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
N, C, K, H = 8, 256, 256, 64
x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.05
y = torch.empty(N, K, H, H, device=dev)
wsb = lib.conv2d_fwd_workspace(N, H, H, K, C, 3, 3, 1, 1); ws = torch.empty(wsb // 4 + 16, device=dev)
need = lib.conv2d_scratch(N, H, H, K, C, 3, 3, 1, 1); arena = torch.empty(need // 4 + 16, device=dev); lib.set_scratch(P(arena), need)
tl = torch.zeros(6 * 8 * 8, dtype=torch.int64, device=dev)
call = lambda pre: lib.conv2d_fwd(P(x), C, None, 0, P(w), None, P(y), N, H, H, K, 3, 3, 1, 1, 1, 0, 0.2, P(ws), wsb, pre, st())
call(0)
for _ in range(5): call(1)
lib.tune_ptr(P(tl)); call(1); torch.cuda.synchronize(); lib.tune_ptr(None)
t = tl.cpu().view(6, 8, 8)
if True:
    for wv in range(4):
        print("wave %d: " % wv + " | ".join("T%d slots %5d vmcnt %4d lgkm %4d barrier %4d period %5d" % (
            40 + i, int(t[wv, i, 1] - t[wv, i, 0]), int(t[wv, i, 2] - t[wv, i, 1]), int(t[wv, i, 3] - t[wv, i, 2]),
            int(t[wv, i, 4] - t[wv, i, 3]), int(t[wv, i + 1, 0] - t[wv, i, 0]) if i < 7 else 0) for i in range(8)))
