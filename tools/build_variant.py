"""Build a VARIANT of the product library for a same-box A/B run (tools/gpu_lib_ab.sh):
    python tools/build_variant.py <name> -DFLAG [-DFLAG ...]   ->  nemar_amd/lib/libnemar_hip_<name>.so
Same sources and flags as nemar_amd/csrc/build.py plus the given defines; objects in nemar_amd/csrc/build/variant_<name>/."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nemar_amd.csrc import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
B.HIPCC_FLAGS = B.HIPCC_FLAGS + flags
B.OBJ_DIR = os.path.join(B.HERE, "build", "variant_" + name)
B.LIB_PATH = os.path.join(B.LIB_DIR, "libnemar_hip_%s.so" % name)
B.LIB_PATH_AB = os.path.join(B.LIB_DIR, "libnemar_hip_%s_ab.so" % name)
print(B.build(force=True))
