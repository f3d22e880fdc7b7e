"""Kernel-level micro-benchmarks (HIP events on the launch stream) for the HBM-bound ops.
Prints one JSON object per case: algorithmic bytes / kernel time -> GB/s.  GPU only."""
import argparse
import ctypes
import json
import sys
import os
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nemar_amd import _lib


def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--size", type=int, default=0, help="only the shape with this image size (PMC passes)")
    ap.add_argument("--sigma0", action="store_true", help="only the sigma = 0 regime (what the training step starts in)")
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    out = []
    for (N, C, H, W) in ((8, 3, 256, 256), (16, 3, 512, 512), (8, 3, 1024, 1024)):
        if a.size and H != a.size:
            continue
        torch.manual_seed(0)
        img = torch.rand(N, C, H, W, device=dev) * 2 - 1
        go = torch.randn(N, C, H, W, device=dev)
        res = torch.empty_like(img)
        gin = torch.empty_like(img)
        wsb = lib.grid_sample_bwd_workspace(N, C, H, W)
        gws = torch.zeros(wsb // 4 + 16, device=dev)
        for sigma in ((0.0,) if a.sigma0 else (0.0, 2.0 / W, 0.1, 'smooth3px')):
            if sigma == 'smooth3px':      # low-frequency field of ~3 pixels amplitude (what a trained registration net emits)
                off = torch.nn.functional.interpolate(torch.randn(N, 2, H // 32, W // 32, device=dev), size=(H, W), mode='bilinear',
                                                      align_corners=False) * (6.0 / W)
            else:
                off = torch.randn(N, 2, H, W, device=dev) * sigma
            gd = torch.empty_like(off)
            px = N * H * W
            t = timeit(lambda: lib.grid_sample_fwd(P(img), P(off), 1, P(res), N, C, H, W, H, W, st()), a.iters)
            out.append(dict(op="grid_sample_fwd", shape=[N, C, H, W], sigma=sigma, us=t * 1e6,
                            GBps=px * 4 * (2 * C + 2) / t / 1e9))
            for variant, name in ((0, "gather fused (default)"), (16, "gather fused 256 threads"), (8, "gather 2-pass"), (2, "global fp32 atomics (round 1)"), (1, "LDS-tile fp32 atomics")):
                lib.grid_sample_tune(variant)
                t = timeit(lambda: lib.grid_sample_bwd(P(img), P(off), 1, P(go), P(gin), 0, P(gd), 0, N, C, H, W, H, W, P(gws), wsb, st()), a.iters)
                out.append(dict(op="grid_sample_bwd+gin", variant=name, shape=[N, C, H, W], sigma=sigma, us=t * 1e6,
                                GBps=px * 4 * (3 * C + 4) / t / 1e9))
            lib.grid_sample_tune(0)
            t = timeit(lambda: lib.grid_sample_bwd(P(img), P(off), 1, P(go), None, 0, P(gd), 0, N, C, H, W, H, W, P(gws), wsb, st()), a.iters)
            out.append(dict(op="grid_sample_bwd", shape=[N, C, H, W], sigma=sigma, us=t * 1e6,
                            GBps=px * 4 * (2 * C + 4) / t / 1e9))
        off = torch.randn(N, 2, H, W, device=dev) * 0.01
        gd = torch.empty_like(off)
        loss = torch.zeros(1, device=dev)
        gs = torch.ones(1, device=dev)
        wsb = lib.smoothness_workspace(N, H, W)
        ws = torch.empty(wsb // 4 + 1, device=dev)
        for alpha, Ci in ((0.0, 0), (2.0, 3)):
            im = img if Ci else None
            t = timeit(lambda: lib.smoothness_fwd(P(off), P(im), Ci, alpha, 1.0, P(loss), 0, P(ws), wsb, N, H, W, st()), a.iters)
            out.append(dict(op="smoothness_fwd", shape=[N, 2, H, W], alpha=alpha, us=t * 1e6,
                            GBps=N * H * W * (8 + 4 * Ci) / t / 1e9))
            t = timeit(lambda: lib.smoothness_bwd(P(off), P(im), Ci, alpha, P(gs), 1.0, P(gd), 0, N, H, W, st()), a.iters)
            out.append(dict(op="smoothness_bwd", shape=[N, 2, H, W], alpha=alpha, us=t * 1e6,
                            GBps=N * H * W * (16 + 4 * Ci) / t / 1e9))
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
