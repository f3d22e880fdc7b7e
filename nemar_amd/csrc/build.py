"""Build the gfx950 C-ABI library with hipcc — twice, from the same sources:

  nemar_amd/lib/libnemar_hip.so      the product (include/nemar_hip.h): measurement switches are compile-time constants, the kernels
                                     only a non-default switch reaches are not compiled, no nemar_tune* entry point exists;
  nemar_amd/lib/libnemar_hip_ab.so   the measurement build (-DNEMAR_AB, include/nemar_hip_ab.h adds nemar_tune / nemar_tune_ptr /
                                     nemar_grid_sample_tune) for tools/ and the A/B tests.  Sources that do not mention the macro
                                     share their object file with the product.

In-tree build (the .so files travel to the GPU box with the repo snapshot).  hipcc cross-compiles for
gfx950 without a GPU.  Usage:  python -m nemar_amd.csrc.build [--force] [--jobs N]
"""
import argparse
import concurrent.futures as cf
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB_DIR = os.path.join(PKG, "lib")
OBJ_DIR = os.path.join(HERE, "build")
LIB_PATH = os.path.join(LIB_DIR, "libnemar_hip.so")
LIB_PATH_AB = os.path.join(LIB_DIR, "libnemar_hip_ab.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
    "-ffp-contract=off",        # keep a*b+c un-fused unless the source says fmaf (parity with the oracle)
    "-Wall", "-Wno-unused-function",
    # No packed-FP32 VOP3P instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) in any kernel of the library.  On the MI355X a
    # v_pk_*_f32 whose op_sel swaps the halves of src1 (what the SLP vectoriser makes of two crossed scalar chains) returns wrong
    # results in lanes 48..63 while waves of ANOTHER kernel on the same SIMD — a second HIP stream — issue 16-bit MFMAs next to VALU
    # work: the round-4 "lost store" anomaly of the side-stream weight gradients (tools/probes/pk_f32_corun.hip is the
    # stand-alone repro).  The feature switch is per compilation, so it also covers what a later compiler version would vectorise.
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
]
# (the host half of a hipcc compilation sees the same -target-feature and says so on stderr: not a diagnostic, filtered below)
_HOST_NOISE = "is not a recognized feature for this target (ignoring feature)"


# fp32 atomicAdd -> global_atomic_add_f32 instead of a CAS loop: only the sources that HAVE floating-point atomics (the non-default
# nemar_tune(14, 0) accumulation, the legacy grid_sample / resize gradients of shapes the gather kernels do not take).  The default
# path of every operator is atomic-free and bitwise reproducible.
UNSAFE_FP_ATOMICS = {"conv.hip", "conv_narrow.hip", "conv_wgrad.hip", "pointwise.hip", "warp.hip"}


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the gfx950 extension")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


_AB_WORDS = re.compile(r"\b(NEMAR_(AB|SWITCH|AB_ONLY)|g_lds_claim)\b")      # (g_lds_claim: a switch variable DECLARED in common.h — its users change with -DNEMAR_AB too)


def ab_sensitive(src):
    """Does -DNEMAR_AB change this translation unit?  (the source or a local header it includes — other than common.h, which only
    DEFINES the macros — uses them)"""
    seen, todo = set(), [src]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(os.path.join(HERE, f)):
            continue
        seen.add(f)
        text = open(os.path.join(HERE, f)).read()
        if f != "common.h" and _AB_WORDS.search(text):
            return True
        todo += re.findall(r'#include\s+"([^"]+)"', text)
    return False


def _compile(src, force, ab=False):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", "_ab.o" if ab else ".o"))
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".h")]
    if force or _stale(obj, deps):
        extra = ["-munsafe-fp-atomics"] if src in UNSAFE_FP_ATOMICS else []
        if ab:
            extra.append("-DNEMAR_AB")
        cmd = [_hipcc(), *HIPCC_FLAGS, *extra, "-c", os.path.join(HERE, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        err = "".join(ln for ln in r.stderr.splitlines(True) if _HOST_NOISE not in ln)
        if err.strip():
            sys.stderr.write(err)
    return obj


def build(force=False, jobs=None, verbose=True):
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sources()
    units = [(s, False) for s in srcs] + [(s, True) for s in srcs if ab_sensitive(s)]
    jobs = jobs or min(len(units), os.cpu_count() or 4)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = dict(zip(units, ex.map(lambda u: _compile(u[0], force, u[1]), units)))
    for lib, ab in ((LIB_PATH, False), (LIB_PATH_AB, True)):
        mine = [objs.get((s, ab), objs[(s, False)]) for s in srcs]
        if force or _stale(lib, mine):
            cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *mine, "-o", lib]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built %s and %s (%d sources, %d of them also with -DNEMAR_AB)" % (LIB_PATH, LIB_PATH_AB, len(srcs), len(units) - len(srcs)))
    return LIB_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs)
