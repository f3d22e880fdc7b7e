"""Run one conv shape repeatedly (for rocprofv3 --pmc passes).  usage: pmc_conv.py [fwd|dgrad|wgrad] [iters] [key=value ...]
The scratch arena is registered (NEMAR_ARENA=0: not), i.e. the resblock shape runs on the split-16 kernels as in the product."""
import ctypes, os, sys
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import _lib
from tests.side_inputs import SideInputs
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lib = SideInputs(_lib.load()); dev = torch.device('cuda:0')
for kv in sys.argv[3:]:          # nemar_tune switches as key=value
    k, v = kv.split('='); lib.tune(int(k), int(v))
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
N, C, K, H, R, s, p, pm = 8, 256, 256, 64, 3, 1, 1, 1
if os.environ.get('NEMAR_PMC_SHAPE'):          # "N,C,K,H,R,stride,pad,pad_mode" (square maps; default: the residual blocks' layer at batch 8)
    N, C, K, H, R, s, p, pm = (int(v) for v in os.environ['NEMAR_PMC_SHAPE'].split(','))
OH = (H + 2 * p - R) // s + 1
x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, R, R, device=dev) * 0.05; b = torch.randn(K, device=dev)
y = torch.empty(N, K, OH, OH, device=dev); gy = torch.randn(N, K, OH, OH, device=dev); gx = torch.empty_like(x); gw = torch.zeros_like(w)
wsb = max(lib.conv2d_fwd_workspace(N, H, H, K, C, R, R, s, p), lib.conv2d_bwd_data_workspace(N, C, H, H, K, R, R, s, p, pm)); ws = torch.empty(wsb // 4 + 16, device=dev)
arena = None
if os.environ.get('NEMAR_ARENA', '1') == '1':
    need = lib.conv2d_scratch(N, H, H, K, C, R, R, s, p)
    arena = torch.empty(need // 4 + 16, device=dev); lib.set_scratch(P(arena), need)
act = 0 if arena is not None else 1
wwb = lib.conv2d_bwd_weight_workspace(N, C, H, H, K, OH, OH, R, R, s, p); ws3 = torch.empty(wwb // 4 + 16, device=dev)
for _ in range(iters):
    if which == 'fwd':
        lib.conv2d_fwd(P(x), C, None, 0, P(w), P(b), P(y), N, H, H, K, R, R, s, p, pm, act, 0.2, P(ws), wsb, 0, st())
    elif which == 'dgrad':
        lib.conv2d_bwd_data(P(gy), P(w), None, 0, 0.0, P(gx), C, None, 0, N, H, H, K, OH, OH, R, R, s, p, pm, P(ws), wsb, 0, st())
    else:
        lib.conv2d_bwd_weight(P(x), C, None, 0, P(gy), P(gw), P(b), N, H, H, K, OH, OH, R, R, s, p, pm, P(ws3), wwb, st())
torch.cuda.synchronize()
