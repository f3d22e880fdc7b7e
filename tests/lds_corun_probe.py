"""TEST INFRASTRUCTURE (tests/test_lds_corun_gpu.py runs it in a fresh process; tools/diag_wgrad_beside.py is the same script under the name
the round-5 notes use).  Is the wide-layer weight gradient (split pass + wgrad_split16_kernel + slab sum: nemar_conv2d_bwd_weight_ex on the fp16 x 3 route) bitwise
repeatable on a SIDE stream while the compute stream runs a given kernel in a loop?  High-statistics form of tools/probes/side_queue_wgrad.cpp
(the second side-stream difference: ~4e-4 per call in the training step, so thousands of calls say nothing): every victim call
is compared with the reference ON THE DEVICE (no host round trip), tens of thousands of calls per co-runner.

    python tests/lds_corun_probe.py [victim calls per co-runner = 20000] [N = 4] [H = 64] [co-runners, comma separated]

Co-runners on the compute stream: none | torch.add (the autograd engine's gradient accumulation: at::native::vectorized_elementwise_kernel,
PyTorch's own build — packed-FP32 instructions allowed) | torch.mul | copy_ | the library's act_bwd (built without packed FP32) | the wide
data-gradient call.  Victims: with the gy planes handed over by a data-gradient call (src2_planes) and splitting gy themselves.
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nemar_amd import _lib  # noqa: E402

if os.environ.get('DIAG_LIB'):
    _lib.DEFAULT_PATH = os.path.abspath(os.environ['DIAG_LIB'])
L = _lib.load()
dev = torch.device('cuda', 0)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H = int(sys.argv[3]) if len(sys.argv) > 3 else 64
only = sys.argv[4].split(',') if len(sys.argv) > 4 else None
W, C, K = H, 256, 256
g0 = torch.Generator(device=dev)
g0.manual_seed(4242)
x = torch.rand(N, C, H, W, device=dev, generator=g0) * 2 - 1
gy = (torch.rand(N, K, H, W, device=dev, generator=g0) * 2 - 1) * 0.01
w = (torch.rand(K, C, 3, 3, device=dev, generator=g0) * 2 - 1) * 0.02
main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)


def p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def words(t):
    out = torch.zeros(N, dtype=torch.int32, device=dev)
    L.absmax_samples(p(t), N, t[0].numel(), p(out), ctypes.c_void_p(main.cuda_stream))
    return out


xmax, gmax = words(x), words(gy)
sb = L.conv2d_scratch(N, H, W, K, C, 3, 3, 1, 1)
assert sb, "not the wide route"
arena_m = torch.empty(sb // 4 + 64, device=dev)
arena_s = torch.empty(sb // 4 + 64, device=dev)
gpb = L.conv2d_gy_planes_bytes(N, C, H, W, K, 3, 3, 1, 1, 1)
planes = torch.empty(gpb // 4 + 64, device=dev)
dwb = L.conv2d_bwd_data_workspace(N, C, H, W, K, 3, 3, 1, 1, 1)
dws = torch.empty(dwb // 4 + 64, device=dev)
gx = torch.empty_like(x)
wwb = L.conv2d_bwd_weight_workspace(N, C, H, W, K, H, W, 3, 3, 1, 1)
wws = torch.empty(wwb // 4 + 64, device=dev)


def extras(arena, src_max=None, src2_max=None, gy_out=None, src2_planes=None):
    e = _lib.ConvExtras()
    e.scratch, e.scratch_bytes = arena.data_ptr(), arena.numel() * 4
    if src_max is not None:
        e.src_max_words, e.src_max_count = src_max.data_ptr(), src_max.numel()
    if src2_max is not None:
        e.src2_max_words, e.src2_max_count = src2_max.data_ptr(), src2_max.numel()
    if gy_out is not None:
        e.gy_planes_out, e.gy_planes_bytes = gy_out.data_ptr(), gy_out.numel() * 4
    if src2_planes is not None:
        e.src2_planes = src2_planes.data_ptr()
    return e


def dgrad(hit, out_planes):
    e = extras(arena_m, gmax, gy_out=out_planes)
    L.conv2d_bwd_data_ex(p(gy), p(w), None, 0, 0.0, p(gx), C, None, 0, N, H, W, K, H, W, 3, 3, 1, 1, 1, p(dws), dwb, hit,
                         ctypes.c_void_p(main.cuda_stream), ctypes.byref(e))


dgrad(0, planes)
assert L.last_gy_planes(), "the data-gradient call did not write the planes"
torch.cuda.synchronize()


def victim(out, handover):
    e = extras(arena_s, xmax, gmax, src2_planes=planes if handover else None)
    out.zero_()
    L.conv2d_bwd_weight_ex(p(x), C, None, 0, p(gy), p(out), None, N, H, W, K, H, W, 3, 3, 1, 1, 1, p(wws), wwb,
                           ctypes.c_void_p(side.cuda_stream), ctypes.byref(e))


NV = 8
VICTIM = os.environ.get('DIAG_VICTIM', 'wide_wgrad')
if VICTIM == 'wide_wgrad':
    gws = [torch.empty_like(w) for _ in range(NV)]
    refs = {}
    for ho in (True, False):
        with torch.cuda.stream(side):
            victim(gws[0], ho)
        torch.cuda.synchronize()
        assert L.last_route() == 2
        refs[ho] = gws[0].clone()
    print('references: hand-over vs own split equal: %s' % torch.equal(refs[True], refs[False]))
elif VICTIM == 'wide_wgrad4':
    # the discriminator's 256 -> 512 4x4 stride-1 layer: wgrad_split16_kernel<4, true>, the form that keeps LDS-DMA and claims its CU's whole LDS
    vN = 8
    vx = torch.rand(vN, 256, 32, 32, device=dev, generator=g0) * 2 - 1
    vg = (torch.rand(vN, 512, 31, 31, device=dev, generator=g0) * 2 - 1) * 0.01

    def vwords(t):
        out = torch.zeros(vN, dtype=torch.int32, device=dev)
        L.absmax_samples(p(t), vN, t[0].numel(), p(out), ctypes.c_void_p(main.cuda_stream))
        return out

    vxm, vgm = vwords(vx), vwords(vg)
    vsb = L.conv2d_scratch(vN, 32, 32, 512, 256, 4, 4, 1, 1)
    assert vsb, "not the wide route"
    varena = torch.empty(vsb // 4 + 64, device=dev)
    vwb = L.conv2d_bwd_weight_workspace(vN, 256, 32, 32, 512, 31, 31, 4, 4, 1, 1)
    vws = torch.empty(vwb // 4 + 64, device=dev)

    def victim(out, handover):
        e = extras(varena, vxm, vgm)
        out.zero_()
        L.conv2d_bwd_weight_ex(p(vx), 256, None, 0, p(vg), p(out), None, vN, 32, 32, 512, 31, 31, 4, 4, 1, 1, 0, p(vws), vwb,
                               ctypes.c_void_p(side.cuda_stream), ctypes.byref(e))

    gws = [torch.empty(512, 256, 4, 4, device=dev) for _ in range(NV)]
    with torch.cuda.stream(side):
        victim(gws[0], False)
    torch.cuda.synchronize()
    print('victim %s: route %d' % (VICTIM, L.last_route()))
    refs = {True: gws[0].clone(), False: gws[0].clone()}
elif VICTIM == 'wgrad2':
    # the exact-fp32 weight gradient (wgrad2_kernel, LDS-DMA staged, several workgroups per CU): the discriminator's 64 -> 128 4x4 stride-2 layer
    vN = 8
    vx = torch.rand(vN, 64, 128, 128, device=dev, generator=g0) * 2 - 1
    vg = (torch.rand(vN, 128, 64, 64, device=dev, generator=g0) * 2 - 1) * 0.01
    vwb = L.conv2d_bwd_weight_workspace(vN, 64, 128, 128, 128, 64, 64, 4, 4, 2, 1)
    vws = torch.empty(vwb // 4 + 64, device=dev)

    def victim(out, handover):
        out.zero_()
        L.conv2d_bwd_weight_ex(p(vx), 64, None, 0, p(vg), p(out), None, vN, 128, 128, 128, 64, 64, 4, 4, 2, 1, 0, p(vws), vwb,
                               ctypes.c_void_p(side.cuda_stream), None)

    gws = [torch.empty(128, 64, 4, 4, device=dev) for _ in range(NV)]
    with torch.cuda.stream(side):
        victim(gws[0], False)
    torch.cuda.synchronize()
    print('victim %s: route %d' % (VICTIM, L.last_route()))
    refs = {True: gws[0].clone(), False: gws[0].clone()}
elif VICTIM in ('wide_dgrad', 's16g_dgrad', 'fused_dgrad'):
    # data-gradient calls on the side stream.  wide_dgrad: igemm_split16_kernel with the reflect fold, behind its own split pass;
    # fused_dgrad: the round-6 form (planes from nemar_instnorm_bwd_planes, skip gradient + max words in the epilogue: igemm_split16_kernel<.., 2>);
    # s16g_dgrad: s16g_kernel's parity classes (the translation net's 64 -> 128 stride-2 layer)
    if VICTIM == 's16g_dgrad':
        vN, vC, vK, vH, vs, vmode = N, 64, 128, 256, 2, 0
        vw = (torch.rand(vK, vC, 3, 3, device=dev, generator=g0) * 2 - 1) * 0.05
        vg = (torch.rand(vN, vK, vH // 2, vH // 2, device=dev, generator=g0) * 2 - 1) * 0.01
        varena, vgm = None, None
    else:
        vN, vC, vK, vH, vs, vmode = N, C, K, H, 1, 1
        vw, vg, varena, vgm = w, gy, arena_s, gmax
    voh = vg.shape[2]
    vdb = L.conv2d_bwd_data_workspace(vN, vC, vH, vH, vK, 3, 3, vs, 1, vmode)
    vdw = torch.empty(vdb // 4 + 64, device=dev)
    fused = None
    if VICTIM == 'fused_dgrad':
        assert L.conv2d_bwd_data_fusable(vN, vC, vH, vH, vK, 3, 3, 1, 1, 1) == 1, "N too small for the fused epilogue (reduction split over workgroups)"
        stats_v = torch.empty(vN * vK, 2, device=dev)
        yv = torch.empty_like(vg)
        xin = torch.rand(vN, vK, vH, vH, device=dev, generator=g0) * 2 - 1
        L.instnorm_fwd(p(xin), None, p(yv), p(stats_v), vN * vK, vH * vH, 1e-5, 0, 0.0, ctypes.c_void_p(main.cuda_stream))
        dpl = torch.empty(2 * vN * (vK // 8) * (vH + 4) * (vH + 4) * 4, device=dev)
        gpl_v = torch.empty(gpb // 4 + 64, device=dev)
        bscale = torch.empty(vN, dtype=torch.int32, device=dev)
        L.instnorm_bwd_planes(p(xin), p(stats_v), p(vg), p(gmax), vN, vK, vH, vH, 0, 0.0, 0.0, 0, 0, 1, None, p(dpl), p(gpl_v), p(bscale), None,
                              ctypes.c_void_p(main.cuda_stream))
        skip = torch.rand(vN, vC, vH, vH, device=dev, generator=g0) * 0.01
        owords = torch.empty(vN * 2049, dtype=torch.int32, device=dev)
        fused = (dpl, bscale, skip, owords)
        torch.cuda.synchronize()

    def victim(out, handover, hit=1):
        e = None
        if fused is not None:
            e = extras(varena, fused[1])
            e.src_planes, e.addend, e.out_max_words = fused[0].data_ptr(), fused[2].data_ptr(), fused[3].data_ptr()
        elif varena is not None:
            e = extras(varena, vgm)
        L.conv2d_bwd_data_ex(p(vg), p(vw), None, 0, 0.0, p(out), vC, None, 0, vN, vH, vH, vK, voh, voh, 3, 3, vs, 1, vmode, p(vdw), vdb, hit,
                             ctypes.c_void_p(side.cuda_stream), ctypes.byref(e) if e is not None else None)

    gws = [torch.empty(vN, vC, vH, vH, device=dev) for _ in range(NV)]
    with torch.cuda.stream(side):
        victim(gws[0], False, 0)
    torch.cuda.synchronize()
    print('victim %s: route %d' % (VICTIM, L.last_route()))
    refs = {True: gws[0].clone(), False: gws[0].clone()}
else:
    # other LDS-DMA staged kernels as the victim: a forward call on the side stream (DIAG_VICTIM = wide_fwd: igemm_split16_kernel, the same
    # layer; s16g_fwd: s16g_kernel, 64 -> 128 3x3 stride 2 at 256 x 256, the translation net's first down-sampling layer; exact_fwd:
    # igemm_kernel of the exact-fp32 route, the registration net's 64 -> 64 layer at 16 x 16)
    if VICTIM == 'wide_fwd':
        vx, vw, vs, vp, vmode, vK = x, w, 1, 1, 1, K
    elif VICTIM == 'exact_fwd':
        vx = torch.rand(8, 64, 16, 16, device=dev, generator=g0) * 2 - 1
        vw = (torch.rand(64, 64, 3, 3, device=dev, generator=g0) * 2 - 1) * 0.05
        vs, vp, vmode, vK = 1, 1, 1, 64
    else:
        vx = torch.rand(N, 64, 256, 256, device=dev, generator=g0) * 2 - 1
        vw = (torch.rand(128, 64, 3, 3, device=dev, generator=g0) * 2 - 1) * 0.05
        vs, vp, vmode, vK = 2, 1, 0, 128
    vN, vC, vH, vW = vx.shape
    oh = (vH + 2 * vp - 3) // vs + 1
    fwb = L.conv2d_fwd_workspace(vN, vH, vW, vK, vC, 3, 3, vs, vp)
    fws_ = torch.empty(fwb // 4 + 64, device=dev)
    vxmax = None
    if VICTIM == 'wide_fwd':
        vxmax = xmax

    def victim(out, handover, hit=1):
        e = extras(arena_s, vxmax) if VICTIM == 'wide_fwd' else None
        L.conv2d_fwd_ex(p(vx), vC, None, 0, p(vw), None, p(out), vN, vH, vW, vK, 3, 3, vs, vp, vmode, 0, 0.0, p(fws_), fwb, hit,
                        ctypes.c_void_p(side.cuda_stream), ctypes.byref(e) if e is not None else None)

    gws = [torch.empty(vN, vK, oh, oh, device=dev) for _ in range(NV)]
    with torch.cuda.stream(side):
        victim(gws[0], False, 0)
    torch.cuda.synchronize()
    print('victim %s: route %d' % (VICTIM, L.last_route()))
    refs = {True: gws[0].clone(), False: gws[0].clone()}

a = torch.rand(N, C, H, W, device=dev)
b = torch.rand(N, C, H, W, device=dev)
c = torch.empty_like(a)
RELU = 1


def co_none():
    pass


def co_add():
    for _ in range(24):
        torch.add(a, b, out=c)


def co_add_alpha():
    for _ in range(24):
        torch.add(a, b, alpha=0.5, out=c)


def co_mul():
    for _ in range(24):
        torch.mul(a, b, out=c)


def co_copy():
    for _ in range(24):
        c.copy_(a)


def co_act():
    for _ in range(24):
        L.act_bwd(p(a), p(b), p(c), a.numel(), RELU, 0.0, ctypes.c_void_p(main.cuda_stream))


def co_dgrad():
    for _ in range(3):
        dgrad(1, None)


planes2 = torch.empty(gpb // 4 + 64, device=dev)


def co_dgrad_dual():     # the data-gradient call as the step issues it: its split pass also writes gy planes for a weight gradient (another buffer)
    for _ in range(3):
        dgrad(1, planes2)


def co_chain_dual():
    for _ in range(3):
        torch.add(a, b, out=c)
        L.instnorm_bwd_max(p(a), p(stats), p(b), p(c), N * C, H * W, RELU, 0.0, p(mwords), C, ctypes.c_void_p(main.cuda_stream))
        dgrad(1, planes2)


AGG = None
_agg_so = os.path.join(ROOT, 'tests', 'emu', '_build', 'liblds_aggressors.so')        # (a build directory that is neither in git nor in the GPU snapshot's ignore list)
if only and any(n.startswith('agg_') for n in only) and not os.path.exists(_agg_so):
    import subprocess
    os.makedirs(os.path.dirname(_agg_so), exist_ok=True)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(ROOT, 'tools', 'probes', 'lds_aggressors.hip'), '-o', _agg_so],
                   check=True)
if os.path.exists(_agg_so):
    AGG = ctypes.CDLL(_agg_so)
    AGG.launch_aggressor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]


def agg(kind):
    def co():
        for _ in range(6):      # 2048 workgroups of 256 threads (the split kernels' grid class), ~20 us each
            AGG.launch_aggressor(kind, 2048, 40, p(a), p(c), a.numel() // 4, ctypes.c_void_p(main.cuda_stream))
    return co


stats = torch.empty(N * C, 2, device=dev)
yin = torch.empty_like(a)
L.instnorm_fwd(p(a), None, p(yin), p(stats), N * C, H * W, 1e-5, RELU, 0.0, ctypes.c_void_p(main.cuda_stream))
mwords = torch.zeros(N * (1 + 2048), dtype=torch.int32, device=dev)       # NEMAR_MAX_WORDS(N): results | partials


def co_in_bwd():
    for _ in range(12):
        L.instnorm_bwd(p(a), p(stats), p(b), p(c), N * C, H * W, RELU, 0.0, ctypes.c_void_p(main.cuda_stream))


def co_in_bwd_max():
    for _ in range(12):
        L.instnorm_bwd_max(p(a), p(stats), p(b), p(c), N * C, H * W, RELU, 0.0, p(mwords), C, ctypes.c_void_p(main.cuda_stream))


def co_chain():          # what the compute stream runs right after the fork in the step: the skip-connection add, InstanceNorm backward, the next data gradient
    for _ in range(3):
        torch.add(a, b, out=c)
        L.instnorm_bwd_max(p(a), p(stats), p(b), p(c), N * C, H * W, RELU, 0.0, p(mwords), C, ctypes.c_void_p(main.cuda_stream))
        dgrad(1, None)


CO = dict(agg_alloc=agg(0), agg_lds=agg(1), agg_glob=agg(2), agg_lds1k=agg(3), dgrad_dual=co_dgrad_dual, chain_dual=co_chain_dual, in_bwd=co_in_bwd, in_bwd_max=co_in_bwd_max, chain=co_chain, none=co_none, add=co_add, add_alpha=co_add_alpha, mul=co_mul, copy=co_copy, act_bwd=co_act, dgrad=co_dgrad)
for name, co in CO.items():
    if only and name not in only:
        continue
    for ho in ((False,) if (os.environ.get('DIAG_OWN_ONLY') or VICTIM != 'wide_wgrad') else (True, False)):
        cnt = torch.zeros((), dtype=torch.int64, device=dev)
        snap = refs[ho].clone()                                        # the LAST differing result (one extra elementwise kernel per call)
        done = 0
        t0 = time.time()
        torch.cuda.synchronize()
        while done < calls:
            co()
            with torch.cuda.stream(side):
                for v in range(NV):
                    victim(gws[v], ho)
                    bad = (gws[v] != refs[ho]).any()
                    torch.where(bad, gws[v], snap, out=snap)
                    cnt += bad
            done += NV
            if done % (NV * 64) == 0:
                torch.cuda.synchronize()          # bound the launch queues
        torch.cuda.synchronize()
        print('co-runner %-10s victim %-9s: %d of %d calls differ  (%.1f s)' % (name, 'hand-over' if ho else 'own split', int(cnt), done, time.time() - t0),
              flush=True)
        if int(cnt) and VICTIM != 'wide_wgrad':
            d = snap != refs[ho]
            print('      last event: %d elements differ, dim 0 %s, dim 1 %s' % (int(d.sum()), d.any(3).any(2).any(1).nonzero().flatten().tolist()[:40],
                                                                              d.any(3).any(2).any(0).nonzero().flatten().tolist()[:40]), flush=True)
        elif int(cnt):
            d = (snap != refs[ho]).view(K, C, 9)
            ks, cs = d.any(2).any(1).nonzero().flatten().tolist(), d.any(2).any(0).nonzero().flatten().tolist()
            rel = float((snap - refs[ho]).abs().max() / refs[ho].abs().max())
            print('      last event: %d elements, k %d..%d (%d), c %d..%d (%d), taps %s, largest difference %.2e of the tensor maximum'
                  % (int(d.sum()), ks[0], ks[-1], len(ks), cs[0], cs[-1], len(cs), d.any(1).any(0).nonzero().flatten().tolist(), rel), flush=True)
