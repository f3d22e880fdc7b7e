"""tools/diag_wgrad_beside.py = tests/lds_corun_probe.py (the stand-alone repro of the LDS-DMA misread; the test tier owns the
script, the round-5 notes and tools/gpu_r5*.sh call it by this name)."""
import os
import runpy
import sys

sys.argv[0] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "lds_corun_probe.py")
runpy.run_path(sys.argv[0], run_name="__main__")
