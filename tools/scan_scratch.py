"""Which kernels of the library use scratch (private segment) memory, i.e. spill registers or index private arrays dynamically?

Compiles every nemar_amd/csrc/*.hip to gfx950 assembly with the product flags and lists the kernels whose
`.private_segment_fixed_size` is non-zero (name, scratch bytes per lane, spilled VGPRs, VGPRs, static LDS).  No GPU needed.
Round 5: a scratch-using kernel on a second HIP stream is what made a concurrently running kernel lose store sectors
 — the product library must list NOTHING here (tests/test_abi.py::test_no_kernel_uses_scratch runs this scan).

    python tools/scan_scratch.py [--all] [file.hip ...]
"""
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nemar_amd.csrc import build as B  # noqa: E402


def _asm(src, out_dir, defines=()):
    out = os.path.join(out_dir, os.path.basename(src).replace(".hip", ".s"))
    extra = ["-munsafe-fp-atomics"] if os.path.basename(src) in B.UNSAFE_FP_ATOMICS else []
    cmd = [B._hipcc(), *B.HIPCC_FLAGS, *extra, *defines, "--cuda-device-only", "-S", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
    return out


def _demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        out = r.stdout.split("\n")
        if len(out) >= len(names):
            return out[:len(names)]
    except OSError:
        pass
    return names


def kernels(asm_path):
    """-> [(mangled name, scratch bytes, spilled vgprs, vgprs, agprs, lds bytes)] from the amdhsa metadata of one .s file"""
    txt = open(asm_path).read()
    res = []
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))      # noqa: E731
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        agpr = int(re.match(r"\s*(\d+)", blk).group(1))
        res.append((name, g("private_segment_fixed_size"), g("vgpr_spill_count"), g("vgpr_count"), agpr,
                    g("group_segment_fixed_size")))
    return res


def scan(files=None, defines=(), show_all=False):
    files = files or [os.path.join(B.HERE, s) for s in B.sources()]
    rows = []
    with tempfile.TemporaryDirectory() as td, cf.ThreadPoolExecutor(max_workers=min(len(files), os.cpu_count() or 4)) as ex:
        for src, asm in zip(files, ex.map(lambda s: _asm(s, td, defines), files)):
            for k in kernels(asm):
                if show_all or k[1] > 0:
                    rows.append((os.path.basename(src),) + k)
    names = _demangle([r[1] for r in rows])
    return [(r[0], n) + r[2:] for r, n in zip(rows, names)]


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = scan([os.path.abspath(a) for a in args] or None, show_all="--all" in sys.argv)
    print("%-24s %7s %6s %5s %5s %7s  %s" % ("file", "scratch", "spills", "vgpr", "agpr", "lds", "kernel"))
    for f, name, priv, spill, vg, ag, lds in rows:
        print("%-24s %7d %6d %5d %5d %7d  %s" % (f, priv, spill, vg, ag, lds, name[:150]))
    print("%d kernel(s) with scratch" % sum(1 for r in rows if r[2] > 0))
