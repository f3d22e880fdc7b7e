"""Stand-alone time of the wide weight-gradient call (own split passes + wgrad_split16_kernel + slab sum) on a chosen build of the library —
used with tools/gpu_lib_ab.sh-style variant builds (round 6: a variant with 1/9 of the slab stores is 2-3 us faster of 334: the strided
slab epilogue is not worth a tap-major layout).   DIAG_LIB=path/to/lib.so python tools/wg_time.py"""
import os, sys, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from nemar_amd import _lib
_lib.DEFAULT_PATH = os.path.abspath(os.environ['DIAG_LIB'])
from nemar_amd._lib import ConvExtras
lib = _lib.load(); dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
N, C, K, H, W = 16, 256, 256, 64, 64
x = torch.randn(N, C, H, W, device=dev); gy = torch.randn(N, K, H, W, device=dev)
gw = torch.zeros(K, C, 3, 3, device=dev)
need = lib.conv2d_scratch(N, H, W, K, C, 3, 3, 1, 1); arena = torch.empty(need // 4 + 64, device=dev)
wsb = lib.conv2d_bwd_weight_workspace(N, C, H, W, K, H, W, 3, 3, 1, 1); ws = torch.empty(wsb // 4 + 64, device=dev)
xw = torch.zeros(N, dtype=torch.int32, device=dev); gwd = torch.zeros(N, dtype=torch.int32, device=dev)
lib.absmax_samples(P(x), N, C * H * W, P(xw), S()); lib.absmax_samples(P(gy), N, K * H * W, P(gwd), S())
def call():
    e = ConvExtras(); e.scratch, e.scratch_bytes = arena.data_ptr(), need
    e.src_max_words, e.src_max_count = xw.data_ptr(), N; e.src2_max_words, e.src2_max_count = gwd.data_ptr(), N
    lib.conv2d_bwd_weight_ex(P(x), C, None, 0, P(gy), P(gw), None, N, H, W, K, H, W, 3, 3, 1, 1, 1, P(ws), wsb, S(), ctypes.byref(e))
for _ in range(5): call()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(50): call()
b.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ['DIAG_LIB']), 'wgrad call (split x, split g, kernel, slab sum): %.1f us' % (a.elapsed_time(b) * 1e3 / 50))
