"""Build the oracle's C restatement (oracle/c/nemar_ref.c) into oracle/_build/libnemar_ref.so with gcc.
Test infrastructure: called by __graft_entry__.build() and by tests/test_oracle_c.py; never used by the product."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "nemar_ref.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libnemar_ref.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        r = subprocess.run(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off", SRC, "-lm", "-o", LIB],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed on oracle/c/nemar_ref.c:\n" + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
