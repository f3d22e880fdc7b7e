"""The C-ABI library builds for gfx950 without a GPU, loads, and exports exactly the entry points that
include/nemar_hip.h declares — and the ctypes table in nemar_amd/_lib.py covers each of them with the right arity.
No compute calls (there is no GPU in the CPU tier)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_decls():
    src = open(os.path.join(ROOT, "include", "nemar_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\s*\*)\s*(nemar_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        decls[m.group(1)] = n
    return decls


@pytest.fixture(scope="module")
def libpath():
    from nemar_amd.csrc import build
    return build.build(verbose=False)


def test_header_declares_something():
    d = header_decls()
    assert len(d) >= 25 and "nemar_grid_sample_fwd" in d and "nemar_conv2d_bwd_weight" in d


def test_library_exports_every_declared_symbol(libpath):
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    declared = set(header_decls())
    assert declared <= exported, sorted(declared - exported)
    extra = {s for s in exported if s.startswith("nemar_")} - declared
    assert not extra, "exported but undeclared: %s" % sorted(extra)


def test_ctypes_table_matches_header(libpath):
    from nemar_amd import _lib
    decls = header_decls()
    assert set(_lib.SIGNATURES) == set(decls), set(_lib.SIGNATURES) ^ set(decls)
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        assert len(argtypes) == decls[name], (name, len(argtypes), decls[name])
    lib = _lib.load()                       # dlopen + symbol binding works without a GPU
    assert lib.nemar_version() >= 100
    assert lib.last_error() == ""


def test_missing_library_fails_loudly(tmp_path):
    from nemar_amd import _lib
    with pytest.raises(_lib.NemarHipError):
        _lib.Library(str(tmp_path / "libnemar_hip.so"))


def test_gpu_object_targets_gfx950(libpath):
    """The fat binary inside the .so carries a gfx950 code object (no other architectures, no host fallback)."""
    data = open(libpath, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in data
