"""Can the data-gradient and the weight-gradient branch of a wide layer overlap on two HIP streams?  Main stream: the layer's backward chain
(InstanceNorm backward -> conv2d_bwd_data [split pass + igemm_split16]); side stream: conv2d_bwd_weight [x split + wgrad_split16 + slab sum +
bias gradient], each with its own arena / workspace.  Prints serial vs concurrent time per layer (batch 16, 256 -> 256 @ 64 x 64, reflect)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import _lib
from nemar_amd._lib import ConvExtras

lib = _lib.load(); dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
N, C, K, H, W = 16, 256, 256, 64, 64
x = torch.randn(N, C, H, W, device=dev); gy = torch.randn(N, K, H, W, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.05
gx = torch.empty_like(x); gw = torch.zeros_like(w); gb = torch.zeros(K, device=dev)
stats = torch.randn(N * K, 2, device=dev).abs() + 0.5; gin = torch.randn(N, K, H, W, device=dev); xin = torch.randn(N, K, H, W, device=dev)
need = lib.conv2d_scratch(N, H, W, K, C, 3, 3, 1, 1)
arena_d = torch.empty(need // 4 + 64, device=dev); arena_w = torch.empty(need // 4 + 64, device=dev)
wsd_b = lib.conv2d_bwd_data_workspace(N, C, H, W, K, 3, 3, 1, 1, 1); wsd = torch.empty(wsd_b // 4 + 64, device=dev)
wsw_b = lib.conv2d_bwd_weight_workspace(N, C, H, W, K, H, W, 3, 3, 1, 1); wsw = torch.empty(wsw_b // 4 + 64, device=dev)
gpb = lib.conv2d_gy_planes_bytes(N, C, H, W, K, 3, 3, 1, 1, 1)
gpl = [torch.empty(gpb // 4 + 64, device=dev) for _ in range(4)]
words = torch.zeros(N, dtype=torch.int32, device=dev)
lib.absmax_samples(P(gy), N, K * H * W, P(words), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
xw = torch.zeros(N, dtype=torch.int32, device=dev)
lib.absmax_samples(P(x), N, C * H * W, P(xw), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
S = lambda s: ctypes.c_void_p(s.cuda_stream)


def dgrad(st, i, pre):
    e = ConvExtras(); e.scratch, e.scratch_bytes = arena_d.data_ptr(), need
    e.src_max_words, e.src_max_count = words.data_ptr(), N
    e.gy_planes_out, e.gy_planes_bytes = gpl[i % 4].data_ptr(), gpb
    lib.conv2d_bwd_data_ex(P(gy), P(w), None, 0, 0.0, P(gx), C, None, 0, N, H, W, K, H, W, 3, 3, 1, 1, 1, P(wsd), wsd_b, pre, S(st), ctypes.byref(e))


def wgrad(st, i):
    e = ConvExtras(); e.scratch, e.scratch_bytes = arena_w.data_ptr(), need
    e.src_max_words, e.src_max_count = xw.data_ptr(), N
    e.src2_max_words, e.src2_max_count = words.data_ptr(), N
    e.src2_planes = gpl[i % 4].data_ptr()
    lib.conv2d_bwd_weight_ex(P(x), C, None, 0, P(gy), P(gw), P(gb), N, H, W, K, H, W, 3, 3, 1, 1, 1, P(wsw), wsw_b, S(st), ctypes.byref(e))


def norm_bwd(st):
    lib.instnorm_bwd(P(gin), P(xin), P(stats), P(gy), N, K, H * W, 0, 0.2, None, 0, S(st)) if False else None


def run(concurrent, layers=18):
    evs = [torch.cuda.Event() for _ in range(layers)]
    for i in range(layers):
        dgrad(main, i, 1)
        if concurrent:
            evs[i].record(main); side.wait_event(evs[i]); wgrad(side, i)
        else:
            wgrad(main, i)
    if concurrent:
        main.wait_stream(side)


dgrad(main, 0, 0); wgrad(main, 0); torch.cuda.synchronize()
for mode in (False, True, False, True):
    run(mode); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(3): run(mode)
    e1.record(main); torch.cuda.synchronize()
    print('%-10s %.1f us per layer (dgrad call + wgrad call)' % ('two streams' if mode else 'one stream', e0.elapsed_time(e1) * 1e3 / 3 / 18))
