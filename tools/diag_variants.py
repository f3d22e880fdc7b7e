"""Variant builds of the C-ABI library for the lost-store investigation: a patched copy of ONE kernel source is
compiled and linked with the product's other objects into tools/probes/_build/libnemar_hip_<variant>.so (git-ignored; travels to
the GPU box).  tools/diag_lost_stores.py loads a variant through DIAG_LIB.  Diagnostic tooling only.

    python tools/diag_variants.py            # builds every variant
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nemar_amd.csrc import build as B  # noqa: E402

OUT = os.path.join(ROOT, "tools", "probes", "_build")

STORE = "if (accum_ggrid) { q[0] += ggx; q[oplane] += ggy; } else { q[0] = ggx; q[oplane] = ggy; }"


def _victim(repl):
    def patch(src):
        assert src.count(STORE) == 1
        return src.replace(STORE, repl)
    return patch


def _k7env(src):
    a = "    hipLaunchKernelGGL(k7_sample_max_kernel, dim3(SMAX_CHUNKS, N), dim3(256), 0, st, p.small.p, (long long)p.small.C * H * W, smax);"
    assert src.count(a) == 1
    src = src.replace(a, "    if (!getenv(\"K7_SKIP_SMAX\"))\n" + a)
    b = "    if (!g.aligned) K7_WGC(false, false)\n    else if (edge) K7_WGC(true, true)\n    else K7_WGC(true, false)\n"
    assert src.count(b) == 1
    src = src.replace(b, "    if (!getenv(\"K7_SKIP_MAIN\")) {\n" + b + "    }\n")
    c = "    nemar_sum_partials(part, stride, slabs, gw, J, true, st);\n    if (p.bias_off >= 0) nemar_sum_partials(part + J, stride, slabs, gb, K, true, st);\n"
    assert src.count(c) == 1
    src = src.replace(c, "    if (!getenv(\"K7_SKIP_SUMS\")) {\n" + c + "    }\n")
    return "#include <cstdlib>\n" + src


def _k7shfl(src):
    a = "#ifdef NEMAR_HOST_EMULATION\n#pragma unroll\n    for (int o = 32; o > 0; o >>= 1) x = max(x, (unsigned)__shfl_xor((int)x, o, 64));\n    return x;\n#else"
    assert src.count(a) == 1
    return src.replace(a, "#if 1\n#pragma unroll\n    for (int o = 32; o > 0; o >>= 1) x = max(x, (unsigned)__shfl_xor((int)x, o, 64));\n    return x;\n#else")


def _k7lb1(src):
    a = "__global__ __launch_bounds__(256, 2) void k7_wgrad_kernel(K7WgParams p) {"
    assert src.count(a) == 1
    return src.replace(a, "__global__ __launch_bounds__(256) void k7_wgrad_kernel(K7WgParams p) {")


def _k7nomfma(src):
    import re
    a = src.index("void k7_wgrad_kernel(K7WgParams p) {")
    b = src.index("constexpr int WG_KW = 4;")
    body = re.sub(r"acc\[i\] = __builtin_amdgcn_mfma_f32_32x32x16_f16\(([^;]*), acc\[i\], 0, 0, 0\);",
                  r"acc[i][0] += __builtin_bit_cast(float, (\1)[0]);", src[a:b])
    body = body.replace("(__builtin_bit_cast(f16x8, al[s]), __builtin_bit_cast(f16x8, bh))[0]", "al[s][0] ^ bh[0]")
    return src[:a] + body + src[b:]


def _wgdeep(src):
    # the wide weight gradient waits for its copies one stage EARLIER (one stage in flight instead of two at a step's barrier)
    a = "            WG_VMCNT(2 * NCP)                          // this wave's copies of stage T + 2 have landed (T + 3, T + 4 in flight)"
    assert src.count(a) == 1
    return src.replace(a, "            WG_VMCNT(NCP)")


def _wgdrain(src):
    # ... or drains them completely (no assumption about the order in which LDS copies complete)
    a = "            WG_VMCNT(2 * NCP)                          // this wave's copies of stage T + 2 have landed (T + 3, T + 4 in flight)"
    assert src.count(a) == 1
    return src.replace(a, "            WG_VMCNT(0)")


def _wgprolog(src):
    # every copy of the four prologue stages has landed before the first barrier
    a = "    WG_VMCNT(2 * NCP)                                  // stages 0 and 1 have landed (2 and 3 may be in flight)"
    assert src.count(a) == 1
    return src.replace(a, "    WG_VMCNT(0)")


def _wgsync(src):
    # __syncthreads() (fence + full drain) instead of the counted wait + raw barrier at the end of a step
    a = "            WG_VMCNT(2 * NCP)                          // this wave's copies of stage T + 2 have landed (T + 3, T + 4 in flight)\n            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): every fragment of step T + 1 is in registers\n            __builtin_amdgcn_s_barrier();              // stage T + 2 complete for all waves; slot of stage T + 1 is free\n"
    assert src.count(a) == 1, src.count(a)
    return src.replace(a, "            __syncthreads();\n")


def _wgnoxcd(src):
    a = "    if (p.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);      // the tiles of one pixel slab stay on one XCD (same sources)"
    assert src.count(a) == 1
    return src.replace(a, "")


def _wglds(nbytes):
    # the kernel's workgroup claims `nbytes` of LDS (dynamic), so that no other kernel's LDS-using workgroup fits beside it on the CU
    def patch(src):
        a = "    __shared__ __attribute__((aligned(16))) u32x4 smem[RING * STAGE16];\n    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;\n    const int tiles = p.KBLK * p.CBLK;"
        assert src.count(a) == 1, src.count(a)
        src = src.replace(a, a.replace("__shared__ __attribute__((aligned(16))) u32x4 smem[RING * STAGE16];", "extern __shared__ __attribute__((aligned(16))) u32x4 smem[];"))
        b = "    if (KS == 3 && oneg) hipLaunchKernelGGL((wgrad_split16_kernel<3, true>), dim3(grid), dim3(256), 0, st, p);"
        assert src.count(b) == 1
        return src.replace(b, "    if (KS == 3 && oneg) {\n        static bool attr_ = false;\n        if (!attr_) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_split16_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, %d); attr_ = true; }\n        hipLaunchKernelGGL((wgrad_split16_kernel<3, true>), dim3(grid), dim3(256), %d, st, p);\n    }" % (nbytes, nbytes))
    return patch


def _wgrot(src):
    # WHICH wave copies what: wave 3 copies the G pieces, waves 0 .. 2 the X pieces (content and LDS position unchanged) — are the failing X
    # fragments tied to the X pieces or to the waves that copy them?
    a = "        const int i = NCP * wid + q;\n        if (i < 4 * GC) {"
    assert src.count(a) == 1, src.count(a)
    src = src.replace(a, "        const int i = (NCP * wid + q + NCP) % NCOL;\n        if (i < 4 * GC) {")
    b = "        u32x4* const d_ = smem + ((stage_) & (RING - 1)) * STAGE16 + wid * (NCP * 64);"
    assert src.count(b) == 1
    src = src.replace(b, "        u32x4* const d_ = smem + ((stage_) & (RING - 1)) * STAGE16 + ((wid + 1) & 3) * (NCP * 64);")
    c = "                u32x4* const d_ = smem + (T & (RING - 1)) * STAGE16 + wid * (NCP * 64);"
    assert src.count(c) == 1
    return src.replace(c, "                u32x4* const d_ = smem + (T & (RING - 1)) * STAGE16 + ((wid + 1) & 3) * (NCP * 64);")


def _wgnoclaim(src):
    a = "(g_lds_claim & 1) != 0"
    assert src.count(a) == 1
    return src.replace(a, "false")


def _chain(*fs):
    def patch(src):
        for f in fs:
            src = f(src)
        return src
    return patch


def _wgcheck(src):
    # every step re-reads the NEXT step's X fragments from LDS after the wait + barrier protocol says they are final and counts differences
    a = "            WG_VMCNT(2 * NCP)                          // this wave's copies of stage T + 2 have landed (T + 3, T + 4 in flight)\n            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): every fragment of step T + 1 is in registers\n"
    assert src.count(a) == 1, src.count(a)
    chk = a + """            if (T + 1 < nsteps) {
                bool bad_ = false;
                _Pragma("unroll") for (int i = 0; i < 2 * KS; ++i) {
                    const u32x4 again_ = S_[b_off + i * 128];
                    bad_ = bad_ || again_[0] != bf[nxt][i >> 1][i & 1][0] || again_[1] != bf[nxt][i >> 1][i & 1][1] || again_[2] != bf[nxt][i >> 1][i & 1][2] || again_[3] != bf[nxt][i >> 1][i & 1][3];
                }
                _Pragma("unroll") for (int i = 0; i < 2 * GC; ++i) {
                    const u32x4 again_ = S_[a_off + i * 128];
                    bad_ = bad_ || again_[0] != af[nxt][i >> 1][i & 1][0] || again_[3] != af[nxt][i >> 1][i & 1][3];
                }
                if (bad_) { atomicAdd(&g_wg_dbg[0], 1u); g_wg_dbg[1] = (unsigned)T; g_wg_dbg[2] = (unsigned)wid; g_wg_dbg[3] = (unsigned)nsteps; }
            }
"""
    src = src.replace(a, chk)
    src = src.replace("struct WgParams {", "__device__ unsigned g_wg_dbg[8];\nstruct WgParams {", 1)
    src += """
extern "C" __attribute__((visibility("default"))) int nemar_wg_dbg(unsigned* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_dbg), 32);
}
"""
    return src


VARIANTS = {
    # the victim (grid_sample's grid-gradient kernel, warp.hip): what about ITS stores matters?
    "fence": ("warp.hip", _victim("if (accum_ggrid) { q[0] += ggx; q[oplane] += ggy; } else { q[0] = ggx; q[oplane] = ggy; "
                                  "asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); }")),
    "swap": ("warp.hip", _victim("if (accum_ggrid) { q[0] += ggx; q[oplane] += ggy; } else { q[oplane] = ggy; q[0] = ggx; }")),
    "nt": ("warp.hip", _victim("if (accum_ggrid) { q[0] += ggx; q[oplane] += ggy; } else { __builtin_nontemporal_store(ggx, q); "
                               "__builtin_nontemporal_store(ggy, q + oplane); }")),
    # the trigger (the stem's weight-gradient call on the side queue, conv_k7.hip): which of its launches?  K7_SKIP_SMAX / _MAIN / _SUMS
    "k7env": ("conv_k7.hip", _k7env),
    # ... and what about the kernel: the DPP wave maximum (row_bcast) replaced by shuffles; one workgroup per CU instead of two
    "k7shfl": ("conv_k7.hip", _k7shfl),
    "k7lb1": ("conv_k7.hip", _k7lb1),
    "wgdeep": ("conv_split16_wgrad.hip", _wgdeep),
    "wgdrain": ("conv_split16_wgrad.hip", _wgdrain),
    "wgcheck": ("conv_split16_wgrad.hip", _wgcheck),
    "wgprolog": ("conv_split16_wgrad.hip", _wgprolog),
    "wgnoclaim": ("conv_split16_wgrad.hip", _wgnoclaim),
    "wgrot": ("conv_split16_wgrad.hip", _chain(_wgnoclaim, _wgrot)),
    "wglds64": ("conv_split16_wgrad.hip", _wglds(65536)),
    "wglds148": ("conv_split16_wgrad.hip", _wglds(148 * 1024)),
    "wgsync": ("conv_split16_wgrad.hip", _wgsync),
    "wgnoxcd": ("conv_split16_wgrad.hip", _wgnoxcd),
}


def build(name):
    fname, patch = VARIANTS[name]
    os.makedirs(OUT, exist_ok=True)
    B.build(verbose=False)                                      # the product objects are current
    vdir = os.path.join(OUT, "src_" + name)
    os.makedirs(vdir, exist_ok=True)
    for h in os.listdir(B.HERE):                                # headers next to the patched source
        if h.endswith(".h"):
            with open(os.path.join(B.HERE, h)) as f, open(os.path.join(vdir, h), "w") as o:
                o.write(f.read())
    with open(os.path.join(B.HERE, fname)) as f:
        text = patch(f.read())
    src = os.path.join(vdir, fname)
    with open(src, "w") as o:
        o.write(text)
    obj = os.path.join(vdir, fname.replace(".hip", ".o"))
    extra = ["-munsafe-fp-atomics"] if fname in B.UNSAFE_FP_ATOMICS else []
    subprocess.run([B._hipcc(), *B.HIPCC_FLAGS, *extra, "-c", src, "-o", obj], check=True)
    objs = [obj if s == fname else os.path.join(B.OBJ_DIR, s.replace(".hip", ".o")) for s in B.sources()]
    lib = os.path.join(OUT, "libnemar_hip_%s.so" % name)
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
    return lib


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(VARIANTS)):
        print(build(n))
