"""bench.py on a VARIANT build of the library (A/B of compiler flags / kernel variants on one box):  DIAG_LIB=path python tools/bench_with_lib.py [bench args]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nemar_amd import _lib  # noqa: E402
if os.environ.get('DIAG_LIB'):
    _lib.DEFAULT_PATH = os.path.abspath(os.environ['DIAG_LIB'])
import bench  # noqa: E402
bench.main()
