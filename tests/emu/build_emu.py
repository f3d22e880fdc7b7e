"""TEST INFRASTRUCTURE ONLY: compile the unmodified nemar_amd/csrc/*.hip kernels for the HOST with the SIMT
emulator header shadowing <hip/hip_runtime.h> (tests/emu/include), producing tests/emu/_build/libnemar_emu.so.
Used by the CPU test tier to check kernel logic against the oracle without a GPU; never by the product."""
import os
import shutil
import subprocess
import concurrent.futures as cf

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "nemar_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libnemar_emu.so")


def _cxx():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("host clang++ (ext_vector_type support) not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    hips = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(HERE, "include", "hip", "hip_runtime.h")]
    flags = ["-std=c++17", "-O1", "-g", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
             "-I", os.path.join(HERE, "include"), "-Wno-unused-function", "-Wno-unknown-attributes",
             "-DNEMAR_AB"]          # the measurement build: the emulated kernel tests drive routes through nemar_tune

    def one(src):
        obj = os.path.join(OUT_DIR, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + hdrs):
            lang = ["-x", "c++"] if src.endswith(".hip") else []
            r = subprocess.run([_cxx(), *flags, *lang, "-c", src, "-o", obj], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("emu compile failed on %s:\n%s" % (src, r.stderr))
        return obj

    srcs = hips + [os.path.join(HERE, "emu_runtime.cpp")]
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(one, srcs))
    if force or _stale(LIB, objs):
        r = subprocess.run([_cxx(), "-shared", "-fPIC", *objs, "-o", LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu link failed:\n%s" % r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
