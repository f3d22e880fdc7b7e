"""Shader clock under the dominant kernels ('prove the clock claim'): a one-wave probe kernel on a second stream
samples s_memtime (shader cycles) and s_memrealtime (100 MHz) every ~25 us while the main stream runs (a) nothing, (b) the fp16 x 3
residual-block convolution back to back (batch 16, as in the step), (c) its weight gradient, (d) a plain-fp32 HBM-bound kernel
(InstanceNorm).  Prints the clock per phase: median / min / max MHz over the samples that fall inside the loaded interval.
Build the probe first: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/clock_probe.hip -o tools/probes/_build/libclock_probe.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nemar_amd import _lib
from tests.side_inputs import SideInputs
lib = SideInputs(_lib.load()); dev = torch.device('cuda:0')
probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'probes', '_build', 'libclock_probe.so'))
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
st = lambda s: ctypes.c_void_p(s.cuda_stream)
N, C, K, H = 16, 256, 256, 64
x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.05
y = torch.empty(N, K, H, H, device=dev); gy = torch.randn(N, K, H, H, device=dev); gw = torch.zeros_like(w)
stats = torch.empty(N * C, 2, device=dev)
wsb = lib.conv2d_fwd_workspace(N, H, H, K, C, 3, 3, 1, 1); ws = torch.empty(wsb // 4 + 16, device=dev)
wwb = lib.conv2d_bwd_weight_workspace(N, C, H, H, K, H, H, 3, 3, 1, 1); ws3 = torch.empty(wwb // 4 + 16, device=dev)
need = lib.conv2d_scratch(N, H, H, K, C, 3, 3, 1, 1); arena = torch.empty(need // 4 + 16, device=dev); lib.set_scratch(P(arena), need)
fwd = lambda pre: lib.conv2d_fwd(P(x), C, None, 0, P(w), None, P(y), N, H, H, K, 3, 3, 1, 1, 1, 0, 0.2, P(ws), wsb, pre, st(main))
wgr = lambda pre: lib.conv2d_bwd_weight(P(x), C, None, 0, P(gy), P(gw), None, N, H, H, K, H, H, 3, 3, 1, 1, 1, P(ws3), wwb, st(main))
inn = lambda pre: lib.instnorm_fwd(P(x), None, P(y), P(stats), N * C, H * H, 1e-5, 1, 0.0, st(main))
SAMPLES, SPIN = 600, 16
for name, fn, reps in (('idle', None, 0), ('fp16x3 conv forward x 60', fwd, 60), ('fp16x3 weight gradient x 40', wgr, 40),
                       ('InstanceNorm forward x 300', inn, 300), ('idle again', None, 0)):
    if fn:
        fn(0)
        for _ in range(3): fn(1)
    torch.cuda.synchronize()
    out = torch.zeros(2 * SAMPLES, dtype=torch.int64, device=dev)
    with torch.cuda.stream(side):
        probe.clock_probe_launch(P(out), SAMPLES, SPIN, st(side))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(reps): fn(1)
    b.record(main)
    torch.cuda.synchronize()
    t = out.cpu().numpy().reshape(SAMPLES, 2)
    dc, dw = np.diff(t[:, 0]).astype(np.float64), np.diff(t[:, 1]).astype(np.float64)
    mhz = dc / dw * 100.0
    span_us = (t[-1, 1] - t[0, 1]) / 100.0
    busy_us = a.elapsed_time(b) * 1e3 if reps else 0.0
    k = int(len(mhz) * min(1.0, busy_us / span_us)) if reps else len(mhz)     # the samples taken while the main stream was busy
    sel = mhz[2:max(k - 2, 3)]
    print('%-30s probe span %7.0f us, main stream busy %7.0f us | sclk over the busy samples: median %6.0f MHz  min %6.0f  max %6.0f  (n=%d)'
          % (name, span_us, busy_us, np.median(sel), sel.min(), sel.max(), len(sel)))
    print('    trace (every 25th sample, MHz): ' + ' '.join('%.0f' % v for v in mhz[::25]))
