"""Both gradient calls of a wide 3x3 layer the way nemar_amd/ops.py issues them (producer max words for x and gy, gy split once for both:
nemar_conv_extras.gy_planes_out -> .src2_planes), repeated, for rocprofv3 --pmc passes.  usage: pmc_bwd_pair.py [iters]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import _lib
from nemar_amd._lib import ConvExtras
lib = _lib.load(); dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
N, C, K, H, W = 16, 256, 256, 64, 64
x = torch.randn(N, C, H, W, device=dev); gy = torch.randn(N, K, H, W, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.05
gx = torch.empty_like(x); gw = torch.zeros_like(w); gb = torch.zeros(K, device=dev)
need = lib.conv2d_scratch(N, H, W, K, C, 3, 3, 1, 1); arena = torch.empty(need // 4 + 64, device=dev)
wsd_b = lib.conv2d_bwd_data_workspace(N, C, H, W, K, 3, 3, 1, 1, 1); wsd = torch.empty(wsd_b // 4 + 64, device=dev)
wsw_b = lib.conv2d_bwd_weight_workspace(N, C, H, W, K, H, W, 3, 3, 1, 1); wsw = torch.empty(wsw_b // 4 + 64, device=dev)
gpb = lib.conv2d_gy_planes_bytes(N, C, H, W, K, 3, 3, 1, 1, 1); gpl = torch.empty(gpb // 4 + 64, device=dev)
gwords = torch.zeros(N, dtype=torch.int32, device=dev); xwords = torch.zeros(N, dtype=torch.int32, device=dev)
lib.absmax_samples(P(gy), N, K * H * W, P(gwords), S()); lib.absmax_samples(P(x), N, C * H * W, P(xwords), S())
for i in range(iters):
    e = ConvExtras(); e.scratch, e.scratch_bytes = arena.data_ptr(), need
    e.src_max_words, e.src_max_count = gwords.data_ptr(), N
    e.gy_planes_out, e.gy_planes_bytes = gpl.data_ptr(), gpb
    lib.conv2d_bwd_data_ex(P(gy), P(w), None, 0, 0.0, P(gx), C, None, 0, N, H, W, K, H, W, 3, 3, 1, 1, 1, P(wsd), wsd_b, 1 if i else 0, S(), ctypes.byref(e))
    assert lib.last_gy_planes() == 1
    e = ConvExtras(); e.scratch, e.scratch_bytes = arena.data_ptr(), need
    e.src_max_words, e.src_max_count = xwords.data_ptr(), N
    e.src2_max_words, e.src2_max_count = gwords.data_ptr(), N
    e.src2_planes = gpl.data_ptr()
    lib.conv2d_bwd_weight_ex(P(x), C, None, 0, P(gy), P(gw), P(gb), N, H, W, K, H, W, 3, 3, 1, 1, 1, P(wsw), wsw_b, S(), ctypes.byref(e))
torch.cuda.synchronize()
