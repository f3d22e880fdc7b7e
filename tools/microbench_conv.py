"""Conv-family micro-benchmark: achieved fp32 TFLOP/s per layer shape of BASELINE config 2 (B=8, 256x256)."""
import argparse
import ctypes
import json
import os
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nemar_amd import _lib
from tests.side_inputs import SideInputs
from tools.microbench import timeit

SHAPES = [  # name, C0, C1, K, R, stride, pad, pad_mode, H
    ("T.resblock 256->256 k3 refl @64", 256, 0, 256, 3, 1, 1, 1, 64),
    ("T.stem 3->64 k7 refl @256", 3, 0, 64, 7, 1, 3, 1, 256),
    ("T.down1 64->128 k3s2 @256", 64, 0, 128, 3, 2, 1, 0, 256),
    ("T.down2 128->256 k3s2 @128", 128, 0, 256, 3, 2, 1, 0, 128),
    ("T.head 64->3 k7 refl @256", 64, 0, 3, 7, 1, 3, 1, 256),
    ("R.res 32->32 k3 refl @256", 32, 0, 32, 3, 1, 1, 1, 256),
    ("R.up1 96->32 k3 @256", 64, 32, 32, 3, 1, 1, 0, 256),
    ("R.res 64->64 k3 refl @128", 64, 0, 64, 3, 1, 1, 1, 128),
    ("R.up2 128->64 k3 @128", 64, 64, 64, 3, 1, 1, 0, 128),
    ("D.l1 6->64 k4s2 @256", 3, 3, 64, 4, 2, 1, 0, 256),
    ("D.l2 64->128 k4s2 @128", 64, 0, 128, 4, 2, 1, 0, 128),
    ("D.l3 128->256 k4s2 @64", 128, 0, 256, 4, 2, 1, 0, 64),
    ("D.l4 256->512 k4s1 @32", 256, 0, 512, 4, 1, 1, 0, 32),
    # (padded width a multiple of four floats: the padded-domain reflect data gradient with 16-byte-aligned rows, for comparison)
    ("X.res 32->32 k3 refl @254", 32, 0, 32, 3, 1, 1, 1, 254),
    ("X.res 64->64 k3 refl @126", 64, 0, 64, 3, 1, 1, 1, 126),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--tune", type=int, nargs=2, action='append', default=[], help="nemar_tune key value")
    ap.add_argument("--only", type=str, default=None)
    ap.add_argument("--arena", action='store_true', help="register a scratch arena (split-16 kernels for the wide 3x3 layers); "
                                                         "forward is then timed without the fused activation, as the resblocks run it")
    a = ap.parse_args()
    lib = SideInputs(_lib.load())
    for k, v in a.tune:
        lib.tune(k, v)
    dev = torch.device("cuda:0")
    act = 0 if a.arena else 1
    arena = None
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    N = a.batch
    for (name, C0, C1, K, R, s, p, pm, H) in SHAPES:
        if a.only and a.only not in name:
            continue
        C = C0 + C1
        OH = (H + 2 * p - R) // s + 1
        x0 = torch.randn(N, C0, H, H, device=dev)
        x1 = torch.randn(N, C1, H, H, device=dev) if C1 else None
        w = torch.randn(K, C, R, R, device=dev) * 0.05
        b = torch.randn(K, device=dev)
        y = torch.empty(N, K, OH, OH, device=dev)
        gy = torch.randn(N, K, OH, OH, device=dev)
        gx0 = torch.empty(N, C0, H, H, device=dev) if not C1 else torch.empty(N, C0, H, H, device=dev)
        gx1 = torch.empty(N, C1, H, H, device=dev) if C1 else None
        gw = torch.zeros_like(w)
        if a.arena:
            need = lib.conv2d_scratch(N, H, H, K, C, R, R, s, p)
            if need:
                arena = torch.empty(need // 4 + 16, device=dev)
                lib.set_scratch(P(arena), need)
            else:
                lib.set_scratch(None, 0)
        wsb = max(lib.conv2d_fwd_workspace(N, H, H, K, C, R, R, s, p), lib.conv2d_bwd_data_workspace(N, C, H, H, K, R, R, s, p, pm))
        ws = torch.empty(wsb // 4 + 16, device=dev)
        ws2 = torch.empty(wsb // 4 + 16, device=dev)
        # weights are packed once per optimizer step in the real path: time the steady state (prepacked = 1)
        lib.conv2d_fwd(P(x0), C0, P(x1), C1, P(w), P(b), P(y), N, H, H, K, R, R, s, p, pm, act, 0.2, P(ws), wsb, 0, st())
        if not (pm == 1 and C1):
            lib.conv2d_bwd_data(P(gy), P(w), None, 0, 0.0, P(gx0), C0, P(gx1), C1, N, H, H, K, OH, OH, R, R, s, p, pm,
                                P(ws2), wsb, 0, st())
        flop = 2.0 * N * K * OH * OH * C * R * R
        t_f = timeit(lambda: lib.conv2d_fwd(P(x0), C0, P(x1), C1, P(w), P(b), P(y), N, H, H, K, R, R, s, p, pm, act, 0.2,
                                            P(ws), wsb, 1, st()), a.iters, 2)
        if pm == 1 and C1:
            t_d = float('nan')
        else:
            t_d = timeit(lambda: lib.conv2d_bwd_data(P(gy), P(w), None, 0, 0.0, P(gx0), C0, P(gx1), C1, N, H, H, K, OH,
                                                     OH, R, R, s, p, pm, P(ws2), wsb, 1, st()), a.iters, 2)
        wwb = lib.conv2d_bwd_weight_workspace(N, C, H, H, K, OH, OH, R, R, s, p)
        ws3 = torch.empty(wwb // 4 + 16, device=dev)
        t_w = timeit(lambda: lib.conv2d_bwd_weight(P(x0), C0, P(x1), C1, P(gy), P(gw), P(b), N, H, H, K, OH, OH, R, R,
                                                   s, p, pm, P(ws3), wwb, st()), a.iters, 2)
        print(json.dumps(dict(layer=name, gflop=flop / 1e9, fwd_us=t_f * 1e6, fwd_TF=flop / t_f / 1e12,
                              dgrad_us=t_d * 1e6, dgrad_TF=flop / t_d / 1e12, wgrad_us=t_w * 1e6,
                              wgrad_TF=flop / t_w / 1e12)))


if __name__ == "__main__":
    main()
