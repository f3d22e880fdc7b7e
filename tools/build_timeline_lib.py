"""Builds nemar_amd/lib/libnemar_hip_tl.so = the measurement library (-DNEMAR_AB) with -DNEMAR_TIMELINE (s_memtime stamps in the wave-specialised igemm,
tools/timeline_ws2.py).  Not loaded by the product; NEMAR_TL_LIB points tools at it."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemar_amd.csrc import build as B
objs = []
os.makedirs('/tmp/nemar_tl', exist_ok=True)
for src in B.sources():
    obj = os.path.join('/tmp/nemar_tl', src.replace('.hip', '.o'))
    subprocess.check_call([B._hipcc(), *B.HIPCC_FLAGS, '-DNEMAR_TIMELINE', '-DNEMAR_AB', '-c', os.path.join(B.HERE, src), '-o', obj])
    objs.append(obj)
out = os.path.join(B.LIB_DIR, 'libnemar_hip_tl.so')
subprocess.check_call([B._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', out])
print(out)
