"""The forward producer (nemar_instnorm_fwd_planes) of a VARIANT build of the library against the product library's: every output buffer
bit for bit, and the launch time of both.   python tools/diag_np_variant.py nemar_amd/lib/libnemar_hip_<name>.so"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nemar_amd import _lib  # noqa: E402

A = _lib.Library(_lib.PRODUCT_PATH)
B = _lib.Library(os.path.abspath(sys.argv[1]))
dev = torch.device('cuda:0')
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
MAXP = 2048


def run(lib, x, res, resw, act, drop, N, C, H, W, want_y=True):
    y = torch.full_like(x, float('nan')) if want_y else None
    stats = torch.full((N * C, 2), float('nan'), device=dev)
    planes = torch.full((2 * N * (C // 8) * (H + 4) * (W + 4) * 16,), 0xAB, dtype=torch.uint8, device=dev)
    xb = lib.raw('nemar_conv2d_x_planes_bytes')(N, C, H, W, 3)
    xw = torch.full((int(xb),), 0xCD, dtype=torch.uint8, device=dev)
    scale = torch.zeros(N, dtype=torch.int32, device=dev)
    words = torch.zeros(N * (1 + MAXP), dtype=torch.int32, device=dev)
    call = lambda: lib.instnorm_fwd_planes(P(x), P(res), P(resw), P(y), P(stats), N, C, H, W, 1e-5, act, 0.2, drop, 1234, 7, P(planes), P(scale),
                                           P(words), P(xw), st)
    call()
    torch.cuda.synchronize()
    outs = [t.clone() for t in (y if want_y else stats, stats, planes, xw, scale, words[:N], words[N:N + N * (C // 8)])]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        call()
    a.record()
    for _ in range(40):
        call()
    b.record()
    torch.cuda.synchronize()
    return outs, a.elapsed_time(b) * 1e3 / 40


g = torch.Generator(device=dev).manual_seed(1)
for (N, C, H, W, act, drop, use_res, want_y) in ((16, 256, 64, 64, 1, 0.5, False, False), (16, 256, 64, 64, 0, 0.0, True, True), (8, 256, 64, 64, 1, 0.5, False, True),
                                                   (2, 64, 32, 32, 2, 0.0, True, True), (3, 128, 8, 32, 1, 0.5, False, True)):
    x = torch.randn(N, C, H, W, device=dev, generator=g)
    res = torch.randn(N, C, H, W, device=dev, generator=g) if use_res else None
    resw = None
    if use_res:
        resw = torch.zeros(N, dtype=torch.int32, device=dev)
        A.absmax_samples(P(res), N, C * H * W, P(resw), st)
    oa, ta = run(A, x, res, resw, act, drop, N, C, H, W, want_y)
    ob, tb = run(B, x, res, resw, act, drop, N, C, H, W, want_y)
    same = [bool(torch.equal(p.view(torch.uint8), q.view(torch.uint8))) for p, q in zip(oa, ob)]
    print('N=%d C=%d %dx%d act=%d drop=%.1f res=%d y=%d: product %.1f us, variant %.1f us; bitwise equal (y, stats, planes, xplanes, scale, max, partial max): %s'
          % (N, C, H, W, act, drop, use_res, want_y, ta, tb, same), flush=True)
