"""`-m gpu`: the wide-layer weight gradient (nemar_conv2d_bwd_weight_ex on the fp16 x 3 route: split passes, wgrad_split16_kernel, slab sum)
is bitwise repeatable on a side stream while the compute stream runs LDS-active kernels — the data-gradient call whose split pass also
writes gy planes (split_dual_kernel: the co-runner that made ~0.3 % of such calls differ in one wave tile before wgrad_split16_kernel
staged through registers) and a plain data-gradient call.  tests/lds_corun_probe.py is the probe (every call is
compared on the device); it runs in a fresh process on the PRODUCT library."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
CALLS = 24000


def test_wide_weight_gradient_is_repeatable_beside_lds_active_kernels():
    env = dict(os.environ, NEMAR_AB_LIBRARY="0")
    env.pop("NEMAR_TUNE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "lds_corun_probe.py"), str(CALLS), "4", "64", "dgrad_dual,dgrad,in_bwd_max"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = re.findall(r"co-runner (\S+)\s+victim (hand-over|own split): (\d+) of (\d+) calls differ", r.stdout)
    assert len(rows) == 6, r.stdout
    for co, victim, bad, n in rows:
        assert int(n) >= CALLS and int(bad) == 0, (co, victim, bad, n)
